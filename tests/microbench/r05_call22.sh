#!/bin/bash
# round 5, GPU call 22: the 32-stream legs of the closing library with 8 steps each, three runs (the default bench's legs run 4 steps: 7.65 and 7.84 ms per
# iteration in the two closing runs) — Linear, Block, fp8
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c22; mkdir -p $O
for rep in 1 2 3; do
for cfg in linear block fp8; do
  extra=""
  if [ $cfg = block ]; then extra="--heads block"; fi
  if [ $cfg = fp8 ]; then extra="--fp8-weights"; fi
  timeout 300 python bench.py --batch 32 --steps 8 --warmup 1 --no-cpu-baseline --no-extra-configs $extra > $O/b32_$cfg$rep.json 2> $O/b32_$cfg$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b32_$cfg$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("$cfg", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "vanilla", v["ms_per_token_step"], "ratio", v["medusa_over_vanilla"], "prefill", r["prefill"]["achieved"])
except Exception as e: print("$cfg", "failed", e)
PY
done; done
