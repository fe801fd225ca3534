#!/bin/bash
# round 6, GPU call 38: how many sibling rows?  WM_SIBLINGS = 5 / 4 / 3 / 2 at one stream (bench line, two rounds)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c38; mkdir -p $O
for rep in 1 2; do
for sib in 5 4 3 2; do
  WM_SIBLINGS=$sib timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b1_$sib.json 2> $O/b1_$sib.err
  python - <<PY
import json
d = json.loads(open("$O/b1_$sib.json").read().strip().splitlines()[-1])
print("WM_SIBLINGS=$sib", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter hits", d.get("sibling_hits"), "of", d["accept_hist"][0])
PY
done
done 2>&1 | tee $O/sweep.log
