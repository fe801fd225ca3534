#!/bin/bash
# round 4, GPU call 7: candidate trees at 32 streams (cost vs tokens per iteration)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c7; mkdir -p $O
timeout 400 python tests/microbench/r04_tree32.py > $O/tree32.txt 2> $O/tree32.err; echo rc $?; cat $O/tree32.txt; tail -3 $O/tree32.err
