#!/bin/bash
# round 4, GPU call 3: (a) per-kernel trace of the fp8 encoder pass at 32 clips (what the 16x16x128 GEMMs cost now); (b) register-budget /
# load-group variants of k_rows_gemm at 352 rows
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c3; mkdir -p $O
PKG=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
export TMPDIR=/tmp
cd /tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_fp8 -o fp8 -- python $GRAFT_REPO_ROOT/bench.py --fp8-weights --batch 32 --steps 1 --warmup 1 --max-new 16 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/prof_fp8.json 2> $O/prof_fp8.err; echo prof rc $?
cd "$GRAFT_REPO_ROOT"
DB=$(find /tmp/kt_fp8 -name "*.db" | head -1); python tests/prof_summary.py $DB $O/r04_kernel_trace_fp8_b32.md | head -24
for v in base occ5 g2; do
  if [ $v = base ]; then unset WM_LIB; else export WM_LIB=$PKG/libwm_$v.so; fi
  timeout 120 python tests/microbench/r04_gemm_time.py $v 2>&1 | grep "^\[" | tee -a $O/gemm_time.txt
done
