#!/bin/bash
# round 6, GPU call 16: flash attention, K fragments ahead / V fragments at the half step's start (fw2) and the score MFMAs of the next group ahead of the softmax (fw3)
# against the product library (all fragments ahead), 32 clips, arms interleaved three times; kernel trace of fw3
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06c16; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
for rep in 1 2 3; do
  for v in f16 fw2 fw3; do
    WM_LIB_F16=$P/libwm_$v.so timeout 300 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | tee -a $O/enc_time.log
  done
done
cd /tmp
for v in fw3; do
WM_LIB_F16=$P/libwm_$v.so timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$v -o kt -- python $GRAFT_REPO_ROOT/tests/microbench/r06_enc_time.py --reps 3 > $O/kt_$v.log 2>&1
DB=$(find /tmp/kt_$v -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tests/prof_summary.py $DB $O/kernel_trace_enc_$v.md | tail -1
grep "k_flash_enc\|k_gemm_256p\|k_enc_ln" $O/kernel_trace_enc_$v.md | cut -c1-150
done
