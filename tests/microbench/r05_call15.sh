#!/bin/bash
# round 5, GPU calls 15, 16, 19: preloaded scalar arguments in the batched kernels too (k_ln_tiles, k_rows_gemm, k_skinny2_gemm) — new library against the
# committed one (libwm_base.so), 32 streams and one stream, interleaved; bit-exactness tests of the batched paths
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c15; mkdir -p $O
L=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "merged_step or batch or streams or micro_batches or wide or bit_exact or carry or logits" 2>&1 | tail -3
for rep in 1 2 3; do
for arm in base new; do
  unset WM_LIB
  if [ $arm = base ]; then export WM_LIB=$L/libwm_base.so; fi
  timeout 200 python bench.py --batch 32 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm$rep.json 2> $O/b32_$arm$rep.err
  timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_$arm$rep.json 2> $O/b1_$arm$rep.err
  python - <<PY
import json
for tag in ("b32", "b1"):
    try:
        d = json.loads(open("$O/%s_$arm$rep.json" % tag).read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
        print(tag, "$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "vanilla", {k: v[k] for k in v if "ms" in k}, "ratio", v["medusa_over_vanilla"])
    except Exception as e: print(tag, "$arm", "failed", e)
PY
done; done
