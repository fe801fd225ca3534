#!/bin/bash
# round 5, GPU call 10: row sub-blocks in the batched LayerNorm launches (k_ln_tiles, WM_LN_SUB = 1 / 2 / 4 / 8 blocks per 16-row token tile):
# bit-exactness tests of the batched paths on the default (4), then the 32-stream bench per setting (Medusa and the vanilla anchor)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c10; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "merged_step or batch or streams or micro_batches or wide" > $O/pytest.log 2>&1; echo pytest rc $?; tail -3 $O/pytest.log
for sub in 1 4 2 8; do
  WM_LN_SUB=$sub timeout 200 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_sub$sub.json 2> $O/b32_sub$sub.err; echo sub $sub rc $?
done
python - <<PY
import json
for sub in (1, 2, 4, 8):
    try:
        d = json.loads(open("$O/b32_sub%d.json" % sub).read().strip().splitlines()[-1]); r = d["roofline"]
        v = d["vanilla_anchor"]
        print("sub", sub, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "decode tok/s", d["decode_tokens_per_sec_per_gpu"], "ratio", v["medusa_over_vanilla"], "vanilla", v.get("tokens_per_sec"), v.get("ms_per_step"), "tok/it", d["tokens_per_iter"], "hist", d.get("accept_hist"))
    except Exception as e: print(sub, "failed", e)
PY
