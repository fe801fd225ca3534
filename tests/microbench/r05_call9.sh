#!/bin/bash
# round 5, call 9 (second session): the whole default GPU suite exactly as the driver runs it (-x), then smoke(), on the sources the round ships
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c9; mkdir -p $O
( time timeout 1100 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=25 > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
tail -45 $O/pytest_gpu.log
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ) 2>&1 | grep real; tail -3 $O/smoke.log
cp gpurun_out/gpu_suite_durations.json $O/ 2>/dev/null; cp gpurun_out/parity_report.json $O/ 2>/dev/null
