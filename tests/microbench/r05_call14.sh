#!/bin/bash
# round 5, GPU call 14: loader arguments of k_skinny_gemm as preloaded scalar kernel arguments (no scalar-memory round trip inside the request batch)
# against the committed library (libwm_base.so), one stream, interleaved; bit-exactness tests on the new library
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c14; mkdir -p $O
L=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "logits or bit_exact or carry or batch or merged or prompt" 2>&1 | tail -3
for rep in 1 2 3 4; do
for arm in base new; do
  unset WM_LIB
  if [ $arm = base ]; then export WM_LIB=$L/libwm_base.so; fi
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_$arm$rep.json 2> $O/b1_$arm$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b1_$arm$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "vanilla", {k: v[k] for k in v if "ms" in k}, "ratio", v["medusa_over_vanilla"])
except Exception as e: print("$arm", "failed", e)
PY
done; done
