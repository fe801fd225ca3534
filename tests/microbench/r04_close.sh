#!/bin/bash
# round 4, last call: the dense-row QKV epilogue as a type (no run-time branch in the one-stream launch): parity subset, HBM traffic counters on
# the final decode sources, kernel traces at 1 / 32 streams, the default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04close; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "merged_step or batch or streams or carry or micro_batches or bit_exact" > $O/pytest.log 2>&1; echo pytest rc $?; tail -3 $O/pytest.log
cd /tmp
echo "== pmc fetch b1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r04_pmc_fetch_size_bench_b1.md $O/r04_pmc_traffic.json | tail -1
echo "== kernel trace b1"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r04_kernel_trace_bench_b1.md | tail -1
echo "== kernel trace b32"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r04_kernel_trace_bench_b32.md | tail -1
cd $R
cp $O/r04_pmc_traffic.json $R/profiles/r04_pmc_traffic.json
echo "== bench default"
( time timeout 900 python bench.py > $O/r04_bench_default.json 2> $O/bench.err ) 2>&1 | grep real
python - <<PY
import json
try:
    d = json.loads(open("$O/r04_bench_default.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("b1", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "frac_executed", r.get("frac_executed"), "traffic", r.get("traffic"), "prefill", r["prefill"]["achieved"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "vanilla frac", r.get("vanilla_step"))
    for c in d["configs"]: print("  ", c["config"][:60], c["tokens_per_sec"], c["ms_per_iteration"], c["medusa_over_vanilla"], c["roofline_frac_hbm"], c["prefill_tflops"], c["parity_checked"], c.get("parity_tokens_compared"), c.get("merged_steps_per_iteration"))
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_checked"])
    print("  sens", {k: (v["ms_per_iteration"], v.get("frac_hbm_executed")) for k, v in d["acceptance_sensitivity"].items()})
except Exception as e:
    print("bench failed", e, open("$O/bench.err").read()[-800:])
PY
