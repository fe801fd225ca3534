#!/bin/bash
# round 3, GPU call 3: in-situ kernel trace of the 32-stream decode (what each launch costs inside the real chain, weights from HBM),
# and the pipelined encoder GEMM with parts switched off (which part bounds it)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c3; mkdir -p $O
echo "== encoder GEMM, parts switched off (32 clips)"; timeout 400 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-dbg --out $O/enc_dbg.json > $O/enc_dbg.log 2>&1; echo rc $?; grep -E "^encoder" $O/enc_dbg.log
echo "== kernel trace, 32 streams"
cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_b32 -o b32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/trace_b32.log 2>&1; echo rc $?
cd $R; DB=$(find $O/trace_b32 -name "*.db" | head -1); python tests/prof_summary.py $DB > $O/trace_b32.md 2>&1; head -45 $O/trace_b32.md
