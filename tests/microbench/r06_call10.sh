#!/bin/bash
# round 6, GPU call 10: after the contraction-proof fold arithmetic: cache probe per contract, the two failing tests, the two-contract file
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c10; mkdir -p $O
for arm in "" "WM_ACT=hilo"; do
  for heads in medusa_block base_head; do
    env $arm timeout 300 python tests/microbench/r06_dbg_probe.py $heads 4 2>&1 | grep "rows first pass"
    env $arm timeout 300 python tests/microbench/r06_dbg_probe.py $heads 10 3 2>&1 | grep "rows first pass"
  done
done | tee $O/probe.log | cut -c1-250
timeout 900 python -m pytest tests/test_gpu_features.py::test_long_prompts_match_the_oracle tests/test_gpu_large.py::test_large_greedy_equals_vanilla_and_batch_consistency tests/test_gpu_act.py tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider 2>&1 | tee $O/pytest.log | grep -E "passed|failed|^FAILED|^ERROR" | cut -c1-300
