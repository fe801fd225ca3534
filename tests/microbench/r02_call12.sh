#!/bin/bash
# round 2, GPU call 12: the whole GPU suite on the final decoder code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c12; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -q -x -p no:cacheprovider --durations=15 > $O/pytest_all.log 2>&1; echo "rc $?"
tail -30 $O/pytest_all.log
