#!/bin/bash
# round 2, GPU call 19: cap the size of a prefetch job (long jobs outlive the launch's own blocks)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c19; mkdir -p $O
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$tag", d["roofline"]["ms_per_launch"], "ms/iter", d["value"], "tok/s; vanilla ms/step", va.get("ms_per_token_step"))
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-600:])
PY
}
run base A=1
run kb96 WM_PF_MAX_KB=96
run kb64 WM_PF_MAX_KB=64
run kb48 WM_PF_MAX_KB=48
run kb32 WM_PF_MAX_KB=32
run base2 A=1
