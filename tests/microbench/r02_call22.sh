#!/bin/bash
# round 2, GPU call 22: encoder GEMMs with all LDS fragment reads of a k-step issued ahead of its MFMAs
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c22; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla "$@" > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); print("$tag", "ms_enc/step", d["ms_encode_per_step"], "prefill TF", d["roofline"]["prefill"]["achieved"], "tok/s", d["value"])
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-600:])
PY
}
run b1
run b32 --batch 32
run b32_fp8 --batch 32 --fp8-weights
run b4 --batch 4
echo "== encoder parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "encoder_output or cross_kv or logmel or fp8_mfma" > $O/pytest.log 2>&1; echo rc $?; tail -2 $O/pytest.log
