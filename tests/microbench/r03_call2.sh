#!/bin/bash
# round 3, GPU call 2: two-token-tile weight-streaming kernel (17..32 rows), tile kernel for the vocabulary projection only,
# persistent patch-lockstep encoder GEMM (+ L2 hit / miss counters of both forms), 32-stream bench
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c2; mkdir -p $O
echo "== pytest batched (micro / tiny shapes)"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py -m gpu -q -x -p no:cacheprovider -k "many_streams or wide_batch or micro_batches or batch_equals or encoder_output or fp8_decoder or long_prompts or greedy_mode" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== pytest large"; timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -k "twelve or four_stream or block_decode or big_batch or greedy_equals" > $O/pytest_large.log 2>&1; echo rc $?; tail -3 $O/pytest_large.log
echo "== sweep"; timeout 600 python tests/microbench/r03_sweep.py --out $O/sweep.json > $O/sweep.log 2>&1; echo rc $?; grep -E "^rows|^encoder" $O/sweep.log
echo "== L2 hit/miss of the encoder GEMMs (32 clips)"
cd /tmp && timeout 400 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --kernel-trace -d $O/pmc_l2 -o enc -- python $R/tests/microbench/r03_sweep.py --enc --enc-only-batch --out $O/sweep_pmc.json > $O/pmc_l2.log 2>&1; echo rc $?
cd $R; python - <<PY
import csv, glob, collections
fs = glob.glob("$O/pmc_l2/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for f in fs:
    for r in csv.DictReader(open(f)):
        k = (r["Kernel_Name"][:60], r.get("Grid_Size", r.get("Grid_Size_X", "")))
        a = agg[k]; a[0] += 1
        if r["Counter_Name"] == "TCC_HIT_sum": a[1] += float(r["Counter_Value"])
        if r["Counter_Name"] == "TCC_MISS_sum": a[2] += float(r["Counter_Value"])
for k, a in sorted(agg.items(), key=lambda kv: -kv[1][1])[:24]:
    tot = a[1] + a[2]
    print(f"{k[0]:60s} grid {k[1]:>9s} n {a[0]:6d} hit {a[1]:.3e} miss {a[2]:.3e} hit-rate {a[1] / tot if tot else 0:.3f}")
PY
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$tag", d["roofline"]["ms_per_launch"], "ms/iter", d["value"], "tok/s; vanilla ms/step", va.get("ms_per_token_step"), "medusa/vanilla", va.get("medusa_over_vanilla"), "prefill TF", d["roofline"]["prefill"]["achieved"])
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-800:])
PY
}
run new A=1
