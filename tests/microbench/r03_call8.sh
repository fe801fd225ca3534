#!/bin/bash
# round 3, GPU call 8: the whole GPU test suite + smoke + the default bench line on the round's code
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c8; mkdir -p $O
echo "== smoke"; timeout 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo rc $?; tail -2 $O/smoke.log
echo "== pytest -m gpu (all)"; ( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_all.log 2>&1 ) 2>&1 | grep real; echo rc $?; tail -12 $O/pytest_all.log
echo "== default bench"; ( time timeout 1200 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; python - <<PY
import json
try:
    d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
    print("value", d["value"], "ms/iter", d["roofline"]["ms_per_launch"], "frac", d["roofline"]["frac"], "vanilla", d["vanilla_anchor"], "prefill", d["roofline"]["prefill"])
    print("cpu", {k: d["cpu_baseline"].get(k) for k in ("value", "parity_checked", "cores")})
    for c in d.get("configs", []):
        print({k: c.get(k) for k in ("config", "tokens_per_sec", "ms_per_iteration", "tokens_per_iteration", "medusa_over_vanilla", "roofline_frac_hbm", "prefill_tflops", "parity_checked", "failed")})
except Exception as e: print("bench failed", e, open("$O/bench_default.err").read()[-1500:])
PY
