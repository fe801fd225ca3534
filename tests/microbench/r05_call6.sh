#!/bin/bash
# round 5, call 6: LayerNorm in the batched GEMM's own prologue (k_rows_norm_gemm, WM_ROWS_NORM=1) against LayerNorm launch + k_rows_gemm:
# parity (micro / tiny batched tests, the large-v2 batched tests), per-GEMM times, the 32-stream bench in both arms (bf16, Block, fp8)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c6; mkdir -p $O
export WM_ROWS_NORM=1
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "batch or streams or merged or wide or micro_batches or fp8_decoder" > $O/pytest_micro.log 2>&1; echo micro rc $?; tail -3 $O/pytest_micro.log
timeout 600 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "twelve_streams or four_stream or thirty_two or block_decode_loop or greedy_equals" > $O/pytest_large.log 2>&1; echo large rc $?; tail -3 $O/pytest_large.log
unset WM_ROWS_NORM
echo "== gemm times"; timeout 300 python tests/microbench/r05_norm_gemm_time.py 2>&1 | tee $O/gemm_time.log
echo "== bench arms"
for cfgname in linear block fp8; do
  for arm in 0 1; do
    export WM_ROWS_NORM=$arm
    case $cfgname in linear) X="";; block) X="--heads block";; fp8) X="--fp8-weights";; esac
    timeout 240 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs $X > $O/b32_${cfgname}_$arm.json 2> $O/b32_${cfgname}_$arm.err; echo $cfgname $arm rc $?
  done
done
unset WM_ROWS_NORM
python - <<PY
import json
for c in ("linear", "block", "fp8"):
    for arm in ("0", "1"):
        try:
            d = json.loads(open("$O/b32_%s_%s.json" % (c, arm)).read().strip().splitlines()[-1]); r = d["roofline"]
            print(c, "WM_ROWS_NORM=" + arm, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "decode tok/s", d["decode_tokens_per_sec_per_gpu"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "vanilla ms", d["vanilla_anchor"]["ms_per_token_step"], "hist", d.get("accept_hist")[:4])
        except Exception as e: print(c, arm, "failed", e)
PY
