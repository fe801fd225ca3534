#!/bin/bash
# round 5, call 2: the whole default GPU suite on the restructured test set (offline fp32 tables, slow marker, 16 oracle threads), then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c2; mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -q -p no:cacheprovider --durations=25 > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
tail -45 $O/pytest_gpu.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ) 2>&1 | grep real; tail -3 $O/smoke.log
cp gpurun_out/gpu_suite_durations.json $O/ 2>/dev/null; cp gpurun_out/parity_report.json $O/ 2>/dev/null
