#!/bin/bash
# round 6, GPU call 11: the whole GPU suite on the contraction-proof fold arithmetic, then the default bench line (all legs)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c11; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu.log | cut -c1-300 | tail -20
grep -h "parity ties" $O/pytest_gpu.log | cut -c1-400
( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real
tail -c 600 $O/bench_default.err
python - <<PY
import json
d = json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("value", d["value"], d["unit"], "ms/step", d["ms_per_step"], "frac", r["frac"], "ms/iter", r.get("ms_per_launch"), "cpu", d["cpu_baseline"]["value"])
for c in d.get("configs", []):
    print({k: c.get(k) for k in ("config", "value", "ms_per_iteration", "medusa_over_vanilla", "roofline_frac", "prefill_tflops", "prefill_frac_mfma", "tokens_per_iteration", "parity_checked")})
PY
