#!/bin/bash
# round 6, GPU call 7: bisecting the two f16-contract failures of call 6 (long prompt 23 + Medusa-Block, two streams; large-v2 clip alone after a two-stream run)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c7; mkdir -p $O
T="tests/test_gpu_features.py::test_long_prompts_match_the_oracle"
for arm in "" "WM_SKINNY2=0" "WM_LN_FOLD=0" "WM_PREFETCH=0" "WM_ACT=hilo" "WM_NO_GRAPH=1" "WM_NO_STEP=1" "WM_NO_CARRY=1"; do
  echo "== arm [$arm]"
  env $arm timeout 300 python -m pytest $T -m gpu -q -p no:cacheprovider -k "23 or 40" 2>&1 | grep -E "passed|failed|^FAILED" | cut -c1-200
done 2>&1 | tee $O/bisect.log
