#!/bin/bash
# round 5, GPU calls 13 and 14: three arms in one call, interleaved three times, one stream: the committed library (libwm_base.so = HEAD sources),
# the new library with the small-operand prefetch off (WM_PREFETCH_SMALL=0: same code, 24 more kernarg bytes), and on.  Then 32 streams base vs new.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c13; mkdir -p $O
L=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
for rep in 1 2 3; do
for arm in base off on; do
  unset WM_LIB WM_PREFETCH_SMALL
  if [ $arm = base ]; then export WM_LIB=$L/libwm_base.so; fi
  if [ $arm = off ]; then export WM_PREFETCH_SMALL=0; fi
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b1_$arm$rep.json 2> $O/b1_$arm$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b1_$arm$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print("$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"])
except Exception as e: print("$arm", "failed", e)
PY
done; done
for rep in 1 2; do
for arm in base new; do
  unset WM_LIB WM_PREFETCH_SMALL
  if [ $arm = base ]; then export WM_LIB=$L/libwm_base.so; fi
  timeout 200 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm$rep.json 2> $O/b32_$arm$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b32_$arm$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("b32 $arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "ratio", v["medusa_over_vanilla"], {k: v[k] for k in v if "tok" in k or "ms" in k})
except Exception as e: print("$arm", "failed", e)
PY
done; done
