#!/bin/bash
# round 6, GPU call 13: K-loop schedule variants of the pipelined 256 x 256 encoder GEMM (WM_GEMM_SCHED 0..4, csrc/wm_encoder.hip), 32 clips, arms interleaved twice
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c13; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
for rep in 1 2; do
  for v in f16 gs1 gs2 gs3 gs4; do
    WM_LIB_F16=$P/libwm_$v.so WM_ABI_ANY=1 WM_LIB=$P/libwm_$v.so timeout 300 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | tee -a $O/enc_time.log
  done
done
