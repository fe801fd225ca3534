#!/bin/bash
# round 6, GPU call 3: the LayerNorm fold after its first profile — statistics partials summed before the K loop's MFMAs (QKV instance back to four
# waves per SIMD), the next GEMM's weights carried by the batched launch that produces the residual rows — against WM_LN_FOLD=0, and WM_PREFETCH=0
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c3; mkdir -p $O
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -x -q -p no:cacheprovider -k "not fp32_table_live and not pinned_fp32" > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
tail -6 $O/pytest_gpu.log | cut -c1-600
grep -h "rel to scale" $O/pytest_gpu.log | cut -c1-400
for fold in 1 0; do
  export WM_LN_FOLD=$fold
  timeout 300 python tests/microbench/r06_gemm_time.py 2> $O/gt_$fold.err | sed "s/^/[fold=$fold] /" | tee -a $O/gemm_time.log
done
for rep in 1 2; do
for arm in "1 1" "0 1" "1 0"; do
  set -- $arm; export WM_LN_FOLD=$1 WM_PREFETCH=$2
  timeout 300 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$1$2$rep.json 2> $O/b32_$1$2$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b32_$1$2$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("fold=$1 prefetch=$2 b32", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "vanilla", v["ms_per_token_step"], "ratio", v["medusa_over_vanilla"], "tok/iter", d["tokens_per_iter"], "steps/iter", r["passes_per_iteration"], flush=True)
except Exception as e: print("fold=$1 prefetch=$2", "failed", e)
PY
done; done 2>&1 | tee $O/bench.log
