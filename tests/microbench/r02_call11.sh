#!/bin/bash
# round 2, GPU call 11: fp8 MFMA encoder path — parity tests (micro / tiny / large-v2) and prefill A/B at 32 clips
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c11; mkdir -p $O
show() { python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1])
    print("$1", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", "prefill TF", d["roofline"]["prefill"]["achieved"], "ms_enc/step", d["ms_encode_per_step"], "tok/iter", d["tokens_per_iter"])
except Exception as e: print("$1", "failed", e, open("$O/$1.err").read()[-1500:])
PY
}
echo "== pytest fp8 small"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -s -k "fp8" > $O/pytest.log 2>&1; echo rc $?; grep -E "fp8 encoder|passed|failed|Error|assert" $O/pytest.log | tail -20
echo "== pytest fp8 large"; timeout 1200 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -s -k "fp8" > $O/pytest_large.log 2>&1; echo rc $?; grep -E "large fp8|passed|failed|Error|assert|parity" $O/pytest_large.log | tail -20
B="--steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla"
echo "== b32 fp8"; timeout 600 python bench.py --batch 32 --fp8-weights $B > $O/b32_fp8.json 2> $O/b32_fp8.err; echo rc $?; show b32_fp8
echo "== b32 bf16"; timeout 600 python bench.py --batch 32 $B > $O/b32_bf16.json 2> $O/b32_bf16.err; echo rc $?; show b32_bf16
echo "== b1 fp8"; timeout 600 python bench.py --fp8-weights $B > $O/b1_fp8.json 2> $O/b1_fp8.err; echo rc $?; show b1_fp8
