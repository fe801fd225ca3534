#!/bin/bash
# round 6, GPU call 24: test_large_twelve_streams_equal_single_stream_runs failed once in call 23 (micro-batched generate against the 12-stream decode): does it follow
# the sibling rows (WM_SIBLINGS=5 / 0), and is it deterministic?
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c24; mkdir -p $O
for rep in 1 2; do for sib in 5 0; do
  echo "== WM_SIBLINGS=$sib rep $rep"
  WM_SIBLINGS=$sib timeout 600 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -k "twelve_streams" 2>&1 | grep -h "^FAILED\|passed\|failed\|At index\|Right contains\|AssertionError" | cut -c1-200 | tee -a $O/repeat.log
done; done
