#!/bin/bash
# round 6, GPU call 32: result-preserving encoder knobs on the final kernels (32 clips): ring depth 3 / 5 under K-loop schedule 3, flash softmax groups of 4
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c32; mkdir -p $O
for rep in 1 2; do
  timeout 300 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | sed "s/^/[default] /" | tee -a $O/enc_knobs.log
  WM_ENC_GEMM_RING=5 timeout 300 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | sed "s/^/[ring5] /" | tee -a $O/enc_knobs.log
  WM_ENC_GEMM_RING=3 timeout 300 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | sed "s/^/[ring3] /" | tee -a $O/enc_knobs.log
  WM_FLASH_VARIANT=1 timeout 300 python tests/microbench/r06_enc_time.py 2>&1 | grep "^lib=" | sed "s/^/[flash groups of 4] /" | tee -a $O/enc_knobs.log
done
