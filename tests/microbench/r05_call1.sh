#!/bin/bash
# round 5, call 1: oracle thread pool on the GPU box; first run of k_rows_lds (LDS-shared token tiles) and the LayerNorm tail: parity on the
# micro shapes, per-GEMM times against k_rows_gemm, bench at 32 streams in three arms, then the large-v2 batched parity tests
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c1; mkdir -p $O
echo "== oracle threads"; timeout 240 python tests/microbench/oracle_threads.py 64 16 8 2>&1 | tee $O/oracle_threads.log
echo "== parity, micro shapes (new kernel + tail on by default)"
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "batch or streams or merged or wide or micro_batches or fp8_decoder" --durations=8 > $O/pytest_micro.log 2>&1; echo rc $?; tail -15 $O/pytest_micro.log
echo "== gemm times"; timeout 300 python tests/microbench/r05_gemm_time.py 2>&1 | tee $O/gemm_time.log
echo "== bench arms"
for arm in old lds ldstail; do
  case $arm in old) export WM_ROWS_LDS=0 WM_LN_TAIL=0;; lds) export WM_ROWS_LDS=1 WM_LN_TAIL=0;; ldstail) export WM_ROWS_LDS=1 WM_LN_TAIL=1;; esac
  timeout 240 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm rc $?
done
unset WM_ROWS_LDS WM_LN_TAIL
python - <<PY
import json
for arm in ("old", "lds", "ldstail"):
    try:
        d = json.loads(open("$O/b32_%s.json" % arm).read().strip().splitlines()[-1]); r = d["roofline"]
        print(arm, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "decode tok/s", d["decode_tokens_per_sec_per_gpu"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "hist", d.get("accept_hist"))
    except Exception as e: print(arm, "failed", e)
PY
echo "== parity, large-v2 batched"
timeout 600 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "twelve_streams or four_stream or thirty_two or block_decode_loop" --durations=8 > $O/pytest_large.log 2>&1; echo rc $?; tail -15 $O/pytest_large.log
