#!/bin/bash
# round 6, GPU call 5: the whole GPU suite with the fp16 single-plane contract as the process default (WM_ACT=f16: engine and oracle)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c5; mkdir -p $O
export WM_ACT=f16
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu_f16.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu_f16.log | cut -c1-300 | tail -40
grep -h "parity ties" $O/pytest_gpu_f16.log | cut -c1-1500
