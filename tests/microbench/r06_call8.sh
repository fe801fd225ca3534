#!/bin/bash
# round 6, GPU call 8: which half of the fold breaks the two-tile kernel on the f16 contract
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c8; mkdir -p $O
T="tests/test_gpu_features.py::test_long_prompts_match_the_oracle"
for arm in "WM_SKINNY2_FOLD=0" "WM_SKINNY2_RESFOLD=0" "WM_PLAN_WAVE_CAP=10"; do
  echo "== arm [$arm]"
  env $arm timeout 300 python -m pytest $T -m gpu -q -p no:cacheprovider -k "23 or 40" 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200
done 2>&1 | tee $O/bisect.log
