#!/bin/bash
# round 3, GPU call 20: fp8 encoder GEMMs with the strip-major tile order and the wave-level epilogues (dequantise in place, then the wrapped functor's)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03c20; mkdir -p $O
timeout 110 python -m pytest tests/test_gpu_large.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -s -k "fp8_mfma_encoder" > $O/pytest.log 2>&1; echo rc $?; grep -i "max|d|\|passed\|failed\|error" $O/pytest.log | tail -8
timeout 70 python bench.py --fp8-weights --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/f8.json 2> $O/f8.err; echo rc $?
python - <<PY
import json
try:
    d = json.loads(open("$O/f8.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print("fp8 b32", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "prefill", r["prefill"]["achieved"], "enc ms", d.get("ms_encode_per_step"))
except Exception as e: print("failed", e)
PY
