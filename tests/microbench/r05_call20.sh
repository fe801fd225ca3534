#!/bin/bash
# round 5, GPU call 20: the fused cross-q + cross-attention launch (WM_FUSE_CQ=1, round 2: 14.6 us against 14.3 us for the two launches it replaces)
# re-measured on the round-5 kernels, one stream, interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c20; mkdir -p $O
for rep in 1 2 3; do
for arm in split fused; do
  unset WM_FUSE_CQ
  if [ $arm = fused ]; then export WM_FUSE_CQ=1; fi
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b1_$arm$rep.json 2> $O/b1_$arm$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b1_$arm$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print("b1", "$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "hist", d["accept_hist"][:4])
except Exception as e: print("$arm", "failed", e, open("$O/b1_$arm$rep.err").read()[-300:])
PY
done; done
