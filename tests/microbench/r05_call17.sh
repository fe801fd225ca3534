#!/bin/bash
# round 5, GPU call 17: one-stream A/B of the closing library against the previous build (libwm_base.so) on a second box (call 16 ran on a box whose
# single-stream numbers were 5-13 % off every other box of the round: power-management state under light load), plus rocm-smi clocks for the record
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c17; mkdir -p $O
L=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
rocm-smi --showclocks --showperflevel --showpower 2>/dev/null | grep -v "^$" | head -20
for rep in 1 2 3; do
for arm in base new; do
  unset WM_LIB
  if [ $arm = base ]; then export WM_LIB=$L/libwm_base.so; fi
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_$arm$rep.json 2> $O/b1_$arm$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b1_$arm$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("b1", "$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "vanilla", v["ms_per_token_step"], "prefill", r["prefill"]["achieved"])
except Exception as e: print("$arm", "failed", e)
PY
done; done
