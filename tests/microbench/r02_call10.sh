#!/bin/bash
# round 2, GPU call 10: LDS-tiled token-tile GEMM (R >= 64 rows): bit-identity tests + 32-stream A/B
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c10; mkdir -p $O
show() { python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$1", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], "tok/iter; medusa/vanilla", va.get("medusa_over_vanilla"), "vanilla ms/step", va.get("ms_per_token_step"))
except Exception as e: print("$1", "failed", e, open("$O/$1.err").read()[-800:])
PY
}
echo "== pytest batched subset"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tree.py -m gpu -q -x -p no:cacheprovider -k "many_streams or wide_batch or micro_batches or batch_equals or tree_decode_tokens" > $O/pytest.log 2>&1; echo rc $?; tail -4 $O/pytest.log
echo "== pytest large twelve"; timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -k "twelve or four_stream" > $O/pytest_large.log 2>&1; echo rc $?; tail -4 $O/pytest_large.log
B="--batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs"
echo "== b32 tiled"; timeout 600 python bench.py $B > $O/b32_tiled.json 2> $O/b32_tiled.err; echo rc $?; show b32_tiled
echo "== b32 rows_gemm"; WM_ROWS_TILED_MIN_MT=0 timeout 600 python bench.py $B > $O/b32_rows.json 2> $O/b32_rows.err; echo rc $?; show b32_rows
echo "== b32 block tiled"; timeout 600 python bench.py $B --heads block > $O/b32_block.json 2> $O/b32_block.err; echo rc $?; show b32_block
