#!/bin/bash
# round 2, GPU call 18: encoder at one clip — flash attention with 64 queries per block, lazy rescale, 64-row GEMM tiles below a higher block count
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c18; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla $EXTRA > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); print("$tag", "ms_enc/step", d["ms_encode_per_step"], "prefill TF", d["roofline"]["prefill"]["achieved"], "tok/s", d["value"])
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-600:])
PY
}
EXTRA=""
run base A=1
run qt1 WM_FLASH_QT1_BELOW=512
run qt1_bm600 WM_FLASH_QT1_BELOW=512 WM_ENC_BM64_BELOW=600
run qt1_bm600_ks1 WM_FLASH_QT1_BELOW=512 WM_ENC_BM64_BELOW=600 WM_ENC_GEMM_KSPLIT=1
run qt1_bm1000 WM_FLASH_QT1_BELOW=512 WM_ENC_BM64_BELOW=1000
EXTRA="--batch 2"
run b2_base A=1
run b2_qt1_bm1000 WM_FLASH_QT1_BELOW=512 WM_ENC_BM64_BELOW=1000
EXTRA="--batch 32"
run b32 A=1
echo "== encoder parity"; timeout 600 env WM_FLASH_QT1_BELOW=512 WM_ENC_BM64_BELOW=1000 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "encoder_output or cross_kv or logmel" > $O/pytest.log 2>&1; echo rc $?; tail -2 $O/pytest.log
