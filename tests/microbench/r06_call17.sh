#!/bin/bash
# round 6, GPU call 17: closing set on the encoder changes (GEMM K-loop schedule 3, flash K fragments ahead): whole GPU suite as the driver runs it, smoke(),
# kernel trace and matrix-pipe busy at 32 streams (encoder kernels changed; decode sources unchanged: the traffic counters of call 12 stay valid), default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c17; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu.log | cut -c1-300 | tail -8
grep -h "parity ties" $O/pytest_gpu.log | cut -c1-200
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ) 2>&1 | grep real; tail -2 $O/smoke.log | cut -c1-200
cp gpurun_out/gpu_suite_durations.json $O/ 2>/dev/null; cp gpurun_out/parity_report.json $O/ 2>/dev/null
cd /tmp
B="--no-cpu-baseline --no-vanilla --no-extra-configs"
echo "== kernel trace b32"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 $B > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r06_kernel_trace_bench_b32.md | tail -1
echo "== mfma busy b32"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/mf -o mf -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --max-new 16 $B > $O/mf.log 2>&1; echo rc $?
DB=$(find /tmp/mf -name "*.db" | head -1); python $R/tests/mfma_summary.py $DB $O/r06_pmc_mfma_busy_bench_b32.md 2>/dev/null | head -8
cd $R
echo "== bench default"
( time timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench.err ) 2>&1 | grep real
python - <<PY
import json
try:
    d = json.loads(open("$O/r06_bench_default.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("b1", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "traffic", r.get("traffic"), "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "enc ms", d["ms_encode_per_step"])
    for c in d["configs"]: print("  ", c["config"][:60], c["tokens_per_sec"], c["ms_per_iteration"], c["medusa_over_vanilla"], c["roofline_frac_hbm"], c["prefill_tflops"], c["prefill_frac_mfma"], c["parity_checked"], c.get("parity_strict"), c.get("parity_ties_followed"))
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("threads_used"), d["cpu_baseline"]["parity_checked"])
except Exception as e:
    print("bench failed", e, open("$O/bench.err").read()[-800:])
PY
