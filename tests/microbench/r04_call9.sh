#!/bin/bash
# round 4, GPU call 9: the whole GPU test suite on the round-4 sources
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c9; mkdir -p $O
timeout 2300 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_full.log 2>&1; echo pytest rc $?; tail -12 $O/pytest_full.log
cp tests/parity_report.json $O/parity_report.json 2>/dev/null
