#!/bin/bash
# round 6, GPU call 15: (a) encoder-side GPU tests on the new defaults (GEMM K-loop schedule 3, flash fragments one half step ahead); (b) four token tiles per wave in
# k_rows_gemm (libwm_tt4.so, WM_ROWS_TT=4 against 2 in one library): per-GEMM times at 352 / 176 rows and the 32-stream bench leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c15; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "encoder or logmel or end_to_end or fp8 or big_batch" > $O/pytest_enc.log 2>&1 ) 2>&1 | grep real
tail -4 $O/pytest_enc.log | cut -c1-300
timeout 200 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | tee -a $O/enc_time.log
timeout 200 python tests/microbench/r06_enc_time.py --fp8 2>&1 | grep "^lib=" | tee -a $O/enc_time.log
for tt in 2 4; do
  echo "== WM_ROWS_TT=$tt"
  WM_ROWS_TT=$tt WM_LIB_F16=$P/libwm_tt4.so timeout 300 python tests/microbench/r06_gemm_time.py 2>&1 | grep "rows=" | tee -a $O/gemm_time_tt$tt.log
done
for tt in 2 4; do
  echo "== bench --batch 32 WM_ROWS_TT=$tt"
  WM_ROWS_TT=$tt WM_LIB_F16=$P/libwm_tt4.so timeout 400 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/bench_b32_tt$tt.json 2> $O/bench_b32_tt$tt.err
  python - <<PY
import json
d = json.loads(open("$O/bench_b32_tt$tt.json").read().strip().splitlines()[-1])
print("tt$tt", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["vanilla_anchor"], d["accept_hist"])
PY
done
