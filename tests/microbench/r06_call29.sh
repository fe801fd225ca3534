#!/bin/bash
# round 6, GPU call 29: does the pool / 12-stream mismatch of call 23 recur in suite context, and is it a followed tie when it does?  The files the suite runs ahead
# of test_gpu_large.py plus that file, three times, stdout kept (check_tokens prints every followed tie)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c29; mkdir -p $O
for rep in 1 2 3; do
( time timeout 1500 python -m pytest tests -m gpu -x -q -s -p no:cacheprovider > $O/suite_$rep.log 2>&1 ) 2>&1 | grep real
grep -h "^FAILED\|passed\|failed" $O/suite_$rep.log | cut -c1-200 | tail -3
grep -h "twelve streams\|parity\[" $O/suite_$rep.log | cut -c1-400 | tail -6
done
