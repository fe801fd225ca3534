#!/bin/bash
# round 6, GPU call 27: encoder + decode bit-stability with two contexts running concurrently (the micro-batch pool's situation), 40 repetitions per library
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c27; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
for lib in f16 fl0; do
  WM_LIB_F16=$P/libwm_$lib.so timeout 900 python tests/microbench/r06_enc_concurrent.py --reps 40 2>&1 | grep "^lib=\|Error\|error" | tee -a $O/concurrent.log
done
