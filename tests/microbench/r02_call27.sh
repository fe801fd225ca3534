#!/bin/bash
# round 2, GPU call 27: launch-plan knobs of the single-stream decode step on the final kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c27; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); print("$tag", d["roofline"]["ms_per_launch"], "ms/iter", d["value"], "tok/s")
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-300:])
PY
}
run base A=1
run rt2_off WM_PLAN_RT2=0
run cap5 WM_PLAN_WAVE_CAP=5
run target512 WM_PLAN_TARGET_WAVES=512
run target2048 WM_PLAN_TARGET_WAVES=2048
run xattn384 WM_XATTN_TARGET_BLOCKS=384
run xattn1536 WM_XATTN_TARGET_BLOCKS=1536
run xattn_nt0 WM_XATTN_NT=0
run base2 A=1
