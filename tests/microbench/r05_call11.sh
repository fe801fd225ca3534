#!/bin/bash
# round 5, GPU call 11: two row tiles per wave for the single-stream vocabulary projection (3242 row tiles: every block re-reads the 56 KB token
# operand for 40 KB of weights; WM_PLAN_RT2_MAXN16 lifts the N16 <= 1024 bound of the RT = 2 plan).  One-stream bench per setting + kernel trace.
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c11; mkdir -p $O
for arm in base rt2 base rt2; do
  if [ $arm = rt2 ]; then export WM_PLAN_RT2_MAXN16=1048576; else unset WM_PLAN_RT2_MAXN16; fi
  timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_$arm.json 2> $O/b1_$arm.err; echo $arm rc $?
  python - <<PY
import json
try:
    d = json.loads(open("$O/b1_$arm.json").read().strip().splitlines()[-1]); r = d["roofline"]
    print("$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "vanilla", d["vanilla_anchor"].get("ms_per_token"), "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "tok/it", d["tokens_per_iter"])
except Exception as e: print("$arm", "failed", e)
PY
done
cd /tmp
export WM_PLAN_RT2_MAXN16=1048576
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/kernel_trace_b1_rt2.md --hbm-large-v2-b1 | tail -3
grep -E "EpF32T<false>|EpHead" $O/kernel_trace_b1_rt2.md | head
cd $R
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "logits or bit_exact or carry" 2>&1 | tail -3
