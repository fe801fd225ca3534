#!/bin/bash
# round 6, GPU call 21: what the sibling rows cost per verify pass (forced accept lengths, WM_SIBLINGS=5 against 0, interleaved), and a kernel trace of both
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06c21; mkdir -p $O
for rep in 1 2; do for sib in 5 0; do WM_SIBLINGS=$sib timeout 300 python tests/microbench/r06_sib_cost.py 2>&1 | grep "^WM_SIBLINGS" | tee -a $O/sib_cost.log; done; done
cd /tmp
for sib in 5 0; do
WM_SIBLINGS=$sib timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_$sib -o kt -- python $GRAFT_REPO_ROOT/tests/microbench/r06_sib_cost.py > $O/kt_$sib.log 2>&1
DB=$(find /tmp/kt_$sib -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tests/prof_summary.py $DB $O/kernel_trace_sib$sib.md | tail -1
head -30 $O/kernel_trace_sib$sib.md | cut -c1-150
done
