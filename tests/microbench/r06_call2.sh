#!/bin/bash
# round 6, GPU call 2: LayerNorm folded into the LayerNorm-fed GEMMs (WM_LN_FOLD, default on): whole GPU suite on it, then the A/B
# against WM_LN_FOLD=0 (round 5's launches, same library): per-GEMM times, one-stream bench, 32-stream bench.
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c2; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
tail -15 $O/pytest_gpu.log | cut -c1-400
for fold in 1 0; do
  export WM_LN_FOLD=$fold
  timeout 300 python tests/microbench/r06_gemm_time.py 2> $O/gt_$fold.err | sed "s/^/[fold=$fold] /" | tee -a $O/gemm_time.log
done
for rep in 1 2; do
for fold in 1 0; do
  export WM_LN_FOLD=$fold
  timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_$fold$rep.json 2> $O/b1_$fold$rep.err
  timeout 300 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$fold$rep.json 2> $O/b32_$fold$rep.err
  python - <<PY
import json
for tag in ("b1", "b32"):
    try:
        d = json.loads(open("$O/%s_$fold$rep.json" % tag).read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
        print("fold=$fold", tag, d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "vanilla", v["ms_per_token_step"], "ratio", v["medusa_over_vanilla"], "tok/iter", d["tokens_per_iter"], flush=True)
    except Exception as e: print("fold=$fold", tag, "failed", e)
PY
done; done 2>&1 | tee $O/bench.log
