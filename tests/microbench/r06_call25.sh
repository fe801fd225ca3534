#!/bin/bash
# round 6, GPU call 25: flake rate of test_large_twelve_streams_equal_single_stream_runs (failed once in call 23, passed 4 / 4 in call 24): 10 runs per arm —
# the product library, the same with WM_SIBLINGS=0, and a library with round 5's encoder kernels (libwm_fl0.so)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c25; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
run() { # name, env...
  local name=$1; shift
  local fails=0
  for rep in 1 2 3 4 5 6 7 8 9 10; do
    env "$@" timeout 300 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -k "twelve_streams" > $O/${name}_$rep.log 2>&1 || fails=$((fails+1))
  done
  echo "$name: $fails failures of 10" | tee -a $O/summary.log
  grep -h "At index\|Right contains\|Left contains\|AssertionError: " $O/${name}_*.log | cut -c1-160 | sort | uniq -c | tee -a $O/summary.log
}
run product WM_SIBLINGS=5
run nosib WM_SIBLINGS=0
run oldenc WM_SIBLINGS=5 WM_LIB_F16=$P/libwm_fl0.so
