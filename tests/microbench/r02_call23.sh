#!/bin/bash
# round 2, GPU call 23: large-v2 and feature suites on the final encoder kernels
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c23; mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_large.py tests/test_gpu_features.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo "rc $?"; tail -4 $O/pytest.log
