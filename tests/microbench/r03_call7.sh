#!/bin/bash
# round 3, GPU call 7: phase-grouped flash attention (encoder), micro-batches and cross-attention split at 32 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c7; mkdir -p $O
echo "== pytest encoder"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "encoder_output or cross_kv or fp8_mfma or silence" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== pytest large encoder"; timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -k "big_batch or end_to_end" > $O/pytest_large.log 2>&1; echo rc $?; tail -4 $O/pytest_large.log
run() { tag=$1; b=$2; shift; shift; env "$@" timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs $EXTRA > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$tag", d["roofline"]["ms_per_launch"], "ms/iter", d["value"], "tok/s; tok/iter", d["tokens_per_iter"], "vanilla ms/step", va.get("ms_per_token_step"), "medusa/vanilla", va.get("medusa_over_vanilla"), "prefill TF", d["roofline"]["prefill"]["achieved"], "enc ms", d["ms_encode_per_step"])
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-800:])
PY
}
run b32 32 A=1
run b32_x1280 32 WM_XATTN_TARGET_BLOCKS=1280
EXTRA="--micro-batches 2" run b32_mb2 32 A=1
EXTRA="--micro-batches 2" run b64_mb2 64 A=1
run b1 1 A=1
