#!/bin/bash
# round 6, GPU call 1: the two upper bounds of the 32-stream step, numerically meaningless arms —
# no bf16 lo plane (NOLO), no LayerNorm launches (NOLN), both — against the product library: per-GEMM times and the 32-stream bench leg.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c1; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
export WM_ABI_ANY=1
for arm in base nolo noln nolonoln; do
  if [ $arm = base ]; then unset WM_LIB; else export WM_LIB=$P/libwm_$arm.so; fi
  timeout 300 python tests/microbench/r06_gemm_time.py 2> $O/gt_$arm.err | tee -a $O/gemm_time.log
done
for rep in 1 2; do
for arm in base nolo noln nolonoln; do
  if [ $arm = base ]; then unset WM_LIB; else export WM_LIB=$P/libwm_$arm.so; fi
  timeout 300 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm$rep.json 2> $O/b32_$arm$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b32_$arm$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "vanilla", v["ms_per_token_step"], "ratio", v["medusa_over_vanilla"], flush=True)
except Exception as e: print("$arm", "failed", e)
PY
done; done 2>&1 | tee $O/bench.log
