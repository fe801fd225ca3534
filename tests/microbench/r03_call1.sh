#!/bin/bash
# round 3, GPU call 1: LDS-ring token-tile GEMM (decode, >= 48 rows) and pipelined 256 x 256 encoder GEMM: parity tests of the batched
# paths, kernel sweeps, 32-stream bench A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c1; mkdir -p $O
echo "== pytest batched (micro / tiny shapes)"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "many_streams or wide_batch or micro_batches or batch_equals or encoder_output or fp8_mfma" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== pytest large"; timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -k "twelve or four_stream or block_decode or big_batch or end_to_end" > $O/pytest_large.log 2>&1; echo rc $?; tail -3 $O/pytest_large.log
echo "== sweep"; timeout 600 python tests/microbench/r03_sweep.py --out $O/sweep.json > $O/sweep.log 2>&1; echo rc $?; grep -E "^rows|^encoder" $O/sweep.log
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
run old WM_TILE_GEMM_MIN_MT=0 WM_ENC_GEMM_256P=0
