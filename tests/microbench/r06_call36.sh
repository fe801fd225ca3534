#!/bin/bash
# round 6, GPU call 36: row-tile count per wave of k_rows_gemm under the fp16 operand — the block-count threshold (WM_ROWS_GEMM_MIN_BLOCKS, swept at 400 under hi / lo)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c36; mkdir -p $O
for rep in 1 2; do
for mb in 400 200 300 600 800; do
  WM_ROWS_GEMM_MIN_BLOCKS=$mb timeout 400 python bench.py --batch 32 --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b32_$mb.json 2> $O/b32_$mb.err
  python - <<PY
import json
d = json.loads(open("$O/b32_$mb.json").read().strip().splitlines()[-1])
print("min_blocks=$mb", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter")
PY
done
done 2>&1 | tee $O/sweep.log
