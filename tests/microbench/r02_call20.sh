#!/bin/bash
# round 2, GPU call 20: LayerNorm fused into the token-tile GEMM (R > 16 rows): bit-identity tests + 32-stream A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c20; mkdir -p $O
echo "== pytest batched"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tree.py -m gpu -q -x -p no:cacheprovider -k "many_streams or wide_batch or micro_batches or batch_equals or tree_decode_tokens or decode_tokens_bit_exact or fp8_decoder" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== pytest large"; timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -k "twelve or four_stream or block_decode" > $O/pytest_large.log 2>&1; echo rc $?; tail -3 $O/pytest_large.log
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$tag", d["roofline"]["ms_per_launch"], "ms/iter", d["value"], "tok/s; vanilla ms/step", va.get("ms_per_token_step"), "medusa/vanilla", va.get("medusa_over_vanilla"))
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-600:])
PY
}
run fused4 A=1
run fused2 WM_ROWS_LN_RT_MAX=2
run fused1 WM_ROWS_LN_RT_MAX=1
run unfused WM_ROWS_LN_FUSED=0
