#!/bin/bash
# round 2, GPU call 25: 2-8 streams as one batched context vs one context per stream / pair (concurrent micro-batches)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c25; mkdir -p $O
run() { tag=$1; shift; timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla "$@" > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); print("$tag", d["value"], "tok/s", d["ms_per_step"], "ms/step", "decode", d["ms_decode_per_step"], "enc", d["ms_encode_per_step"])
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-500:])
PY
}
run b2_mb1 --batch 2
run b2_mb2 --batch 2 --micro-batches 2
run b3_mb1 --batch 3
run b3_mb3 --batch 3 --micro-batches 3
run b4_mb1 --batch 4
run b4_mb2 --batch 4 --micro-batches 2
run b4_mb4 --batch 4 --micro-batches 4
run b8_mb1 --batch 8
run b8_mb2 --batch 8 --micro-batches 2
run b8_mb4 --batch 8 --micro-batches 4
