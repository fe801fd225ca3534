#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c8; mkdir -p $O
python whisper-medusa_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
show() { python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], "prefill TF", d["roofline"]["prefill"]["achieved"], "ms_enc", d["ms_encode_per_step"])
except Exception as e: print("$1", "failed", e, open("$O/$1.err").read()[-800:])
PY
}
B="--batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla"
echo "== b32 256-tile"; timeout 600 python bench.py $B > $O/b32_256.json 2> $O/b32_256.err; echo rc $?; show b32_256
echo "== b32 128-tile"; WM_ENC_GEMM_256=0 timeout 600 python bench.py $B > $O/b32_128.json 2> $O/b32_128.err; echo rc $?; show b32_128
echo "== pytest large subset"; timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -s -k "encoder_big or twelve" > $O/pytest.log 2>&1; echo rc $?; tail -8 $O/pytest.log
