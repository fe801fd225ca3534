#!/bin/bash
# round 2, GPU call 24: the default bench line on the final code (kept as profiles/r02_bench_default.json)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c24; mkdir -p $O
timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err; echo rc $?
python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1])
print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["roofline"]["traffic"], d["roofline"]["prefill"], d["vanilla_anchor"], d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_checked"])
for e in d["configs"]: print(e["config"][:40], e["tokens_per_sec"], e["ms_per_iteration"], e["medusa_over_vanilla"], e["prefill_tflops"])
PY
