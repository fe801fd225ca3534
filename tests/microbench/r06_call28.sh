#!/bin/bash
# round 6, GPU call 28: the whole GPU suite as the driver runs it, twice (flake check after call 23's one-off), then smoke()
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c28; mkdir -p $O
for rep in 1 2; do
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu_$rep.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed\|parity\[" $O/pytest_gpu_$rep.log | cut -c1-300 | tail -8
grep -h "parity ties" $O/pytest_gpu_$rep.log | cut -c1-160
done
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ) 2>&1 | grep real; tail -2 $O/smoke.log | cut -c1-160
