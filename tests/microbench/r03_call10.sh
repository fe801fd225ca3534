#!/bin/bash
# round 3, GPU call 10: encoder epilogues by instruction count (branch-free GELU, wave-level q/k/v stores, V tiles token-major),
# fragment-layout LayerNorm, flash attention with 2^x softmax, MFMA results in VGPRs (build flag) — parity tests that cover them, then timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c10; mkdir -p $O
echo "== pytest (encoder, fp8 encoder, logits, tokens: tiny shapes)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "encoder or logits or bit_exact or end_to_end or fp8" > $O/pytest_parity.log 2>&1; echo rc $?; tail -5 $O/pytest_parity.log
echo "== pytest (large-v2: big-batch encoder, prompt pass, batch consistency, fp8)"
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "big_batch or prompt_pass or greedy_equals or twelve_streams or fp8_mfma or linear_decode" > $O/pytest_large.log 2>&1; echo rc $?; tail -5 $O/pytest_large.log
echo "== encoder (32 clips, 1 clip)"
timeout 600 python tests/microbench/r03_sweep.py --enc --enc-now --out $O/enc.json > $O/enc.log 2>&1; echo rc $?; grep "^encoder" $O/enc.log
echo "== bench 32 streams / 1 stream"
timeout 600 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32.json 2> $O/b32.err; echo rc $?
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b1.json 2> $O/b1.err; echo rc $?
python - <<PY
import json
for n in ("b32", "b1"):
    try:
        d = json.loads(open("$O/%s.json" % n).read().strip().splitlines()[-1])
        r = d["roofline"]; v = d.get("vanilla_anchor", {})
        print(n, d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "traffic", r.get("traffic"), "prefill TF", r["prefill"]["achieved"], "vanilla ms", v.get("ms_per_token_step"), "ratio", v.get("medusa_over_vanilla"), "enc ms", d.get("ms_encode_per_step"))
    except Exception as e:
        print(n, "failed", e, open("$O/%s.err" % n).read()[-600:])
PY
