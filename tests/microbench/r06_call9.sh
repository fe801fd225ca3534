#!/bin/bash
# round 6, GPU call 9: where a two-stream pass first differs from the single-stream passes (TEMP debug taps)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c9; mkdir -p $O
timeout 300 python tests/microbench/r06_dbg_stop.py base_head 7 2>&1 | grep "stop after\|Error\|error" | tee $O/stop18.log | cut -c1-250
timeout 300 python tests/microbench/r06_dbg_stop.py base_head 15 2>&1 | grep "stop after\|Error\|error" | tee $O/stop32.log | cut -c1-250
