#!/bin/bash
# round 6, GPU call 26: the twelve-streams test in the order the suite runs it (behind test_large_greedy_equals_vanilla_and_batch_consistency, same module fixture),
# 8 runs per arm, failing runs keep their full assertion text
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c26; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
run() {
  local name=$1; shift
  local fails=0
  for rep in 1 2 3 4 5 6 7 8; do
    env "$@" timeout 400 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -k "greedy_equals_vanilla or twelve_streams" > $O/${name}_$rep.log 2>&1 || fails=$((fails+1))
  done
  echo "$name: $fails failures of 8" | tee -a $O/summary.log
  grep -h "At index\|Right contains\|Left contains\|AssertionError: \|^FAILED" $O/${name}_*.log | cut -c1-200 | sort | uniq -c | tee -a $O/summary.log
}
run product WM_SIBLINGS=5
run nosib WM_SIBLINGS=0
run oldenc WM_SIBLINGS=5 WM_LIB_F16=$P/libwm_fl0.so
