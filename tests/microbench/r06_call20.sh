#!/bin/bash
# round 6, GPU call 20: sibling rows (wm_config.sibling_rows, ABI v9): the new test file, the parity file, the large-v2 decode-loop tests; then the one-stream bench
# with the rows on / off (WM_SIBLINGS) interleaved
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c20; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_siblings.py -m gpu -q -p no:cacheprovider -s > $O/pytest_sib.log 2>&1 ) 2>&1 | grep real
grep -h "^FAILED\|^ERROR\|passed\|failed\|sibling rows" $O/pytest_sib.log | cut -c1-300 | tail -20
( time timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_act.py -m gpu -q -p no:cacheprovider > $O/pytest_par.log 2>&1 ) 2>&1 | grep real
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_par.log | cut -c1-300 | tail -8
( time timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -s -k "linear_decode_loop or block_decode_loop or natural_eos or hard_length" > $O/pytest_large.log 2>&1 ) 2>&1 | grep real
grep -h "^FAILED\|^ERROR\|passed\|failed\|sibling hits" $O/pytest_large.log | cut -c1-300 | tail -12
for rep in 1 2; do
  for sib in 5 0; do
    WM_SIBLINGS=$sib timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_sib${sib}_$rep.json 2> $O/b1_sib${sib}_$rep.err
    python - <<PY
import json
d = json.loads(open("$O/b1_sib${sib}_$rep.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("WM_SIBLINGS=$sib", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "frac_executed", r["frac_executed"], "passes/iter", r["passes_per_iteration"], "hits", d.get("sibling_hits"), "hist", d["accept_hist"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"] if d.get("vanilla_anchor") else None)
PY
  done
done
