#!/bin/bash
# round 4, GPU call 6: in-kernel timeline of the single-stream chain on the round-4 kernels
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c6; mkdir -p $O
WM_LIB=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa/libwm_tl.so WM_ABI_ANY=1 timeout 200 python tests/microbench/timeline.py --out $O/timeline > $O/timeline.log 2>&1; echo rc $?
tail -45 $O/timeline.log
