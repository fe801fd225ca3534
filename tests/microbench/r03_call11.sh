#!/bin/bash
# round 3, GPU call 11: where the 32-clip encoder pass goes after call 10 (kernel trace), and whether the GEMM epilogues are cheaper
# when the blocks of an XCD are out of phase (every other block starts late)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c11; mkdir -p $O
echo "== encoder, odd blocks late"
timeout 600 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-stagger --out $O/enc_stagger.json > $O/enc_stagger.log 2>&1; echo rc $?; grep "^encoder" $O/enc_stagger.log
cd /tmp
echo "== kernel trace, encoder 32 clips"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kte -o kte -- python $R/tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-default --out $O/enc_traced.json > $O/kte.log 2>&1; echo rc $?
DB=$(find /tmp/kte -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r03_kernel_trace_encoder_b32.md | tail -2
head -30 $O/r03_kernel_trace_encoder_b32.md
