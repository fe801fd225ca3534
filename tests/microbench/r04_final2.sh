#!/bin/bash
# round 4, closing call: the whole GPU suite on the final sources (merged-step schedule on), then the measurement set of r04_final.sh
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04final2; mkdir -p $O
( time timeout 1750 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=15 > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
tail -30 $O/pytest_gpu.log
bash tests/microbench/r04_final.sh
