#!/bin/bash
# round 6, GPU call 22: sibling rows with the 32-slice candidate kernel: the sibling test file, cost at forced accept lengths and the free run (WM_SIBLINGS=5 / 0,
# interleaved), kernel time of k_sib_cand / k_kv_compact, then the one-stream bench both ways
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06c22; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_siblings.py -m gpu -q -p no:cacheprovider -s > $O/pytest_sib.log 2>&1 ) 2>&1 | grep real
grep -h "^FAILED\|^ERROR\|passed\|failed\|sibling rows" $O/pytest_sib.log | cut -c1-300 | tail -14
for rep in 1 2; do for sib in 5 0; do WM_SIBLINGS=$sib timeout 300 python tests/microbench/r06_sib_cost.py 2>&1 | grep "^WM_SIBLINGS" | tee -a $O/sib_cost.log; done; done
cd /tmp
WM_SIBLINGS=5 timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/kt_5 -o kt -- python $GRAFT_REPO_ROOT/tests/microbench/r06_sib_cost.py > $O/kt_5.log 2>&1
DB=$(find /tmp/kt_5 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tests/prof_summary.py $DB $O/kernel_trace_sib5.md | tail -1
grep "k_sib_cand\|k_kv_compact\|k_accept\|k_set_cand" $O/kernel_trace_sib5.md | cut -c1-150
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for sib in 5 0; do
    WM_SIBLINGS=$sib timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_sib${sib}_$rep.json 2> $O/b1_sib${sib}_$rep.err
    python - <<PY
import json
d = json.loads(open("$O/b1_sib${sib}_$rep.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("WM_SIBLINGS=$sib", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "frac_executed", r["frac_executed"], "passes/iter", r["passes_per_iteration"], "hits", d.get("sibling_hits"), "hist", d["accept_hist"])
PY
  done
done
