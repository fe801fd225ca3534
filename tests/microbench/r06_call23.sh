#!/bin/bash
# round 6, GPU call 23: closing set on the sources with the sibling rows (ABI v9) and the encoder changes: whole GPU suite as the driver runs it, smoke(), HBM traffic
# counters at one stream (with / without prefetch blocks), kernel traces at 1 and 32 streams (bf16 and fp8 legs), matrix-pipe busy at 32 streams, default bench line
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${WM_CALL_DIR:-r06c23}; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider --durations=8 > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu.log | cut -c1-300 | tail -8
grep -h "parity ties" $O/pytest_gpu.log | cut -c1-200
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1 ) 2>&1 | grep real; tail -2 $O/smoke.log | cut -c1-200
cp gpurun_out/gpu_suite_durations.json $O/ 2>/dev/null; cp gpurun_out/parity_report.json $O/ 2>/dev/null
cd /tmp
B="--no-cpu-baseline --no-vanilla --no-extra-configs"
echo "== pmc fetch b1, no prefetch blocks"
WM_PREFETCH=0 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc2 -o pmc2 -- python $R/bench.py --steps 2 --warmup 1 $B > $O/pmc2.log 2>&1; echo rc $?
DB=$(find /tmp/pmc2 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r06_pmc_fetch_size_bench_b1_noprefetch.md $O/r06_pmc_traffic_noprefetch.json | tail -1
echo "== pmc fetch b1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 $B > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r06_pmc_fetch_size_bench_b1.md $O/r06_pmc_traffic.json $O/r06_pmc_traffic_noprefetch.json | tail -1
echo "== kernel trace b1"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 4 --warmup 1 $B > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r06_kernel_trace_bench_b1.md --hbm-large-v2-b1 | tail -4
echo "== kernel trace b32"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 $B > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r06_kernel_trace_bench_b32.md | tail -1
echo "== kernel trace b32 fp8"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt32f -o kt32f -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --fp8-weights $B > $O/kt32f.log 2>&1; echo rc $?
DB=$(find /tmp/kt32f -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r06_kernel_trace_bench_b32_fp8.md | tail -1
echo "== mfma busy b32"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/mf -o mf -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --max-new 16 $B > $O/mf.log 2>&1; echo rc $?
DB=$(find /tmp/mf -name "*.db" | head -1); python $R/tests/mfma_summary.py $DB $O/r06_pmc_mfma_busy_bench_b32.md 2>/dev/null | head -8
cd $R
cp $O/r06_pmc_traffic.json $R/profiles/r06_pmc_traffic.json
echo "== bench default"
( time timeout 900 python bench.py > $O/r06_bench_default.json 2> $O/bench.err ) 2>&1 | grep real
python - <<PY
import json
try:
    d = json.loads(open("$O/r06_bench_default.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("b1", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "frac_executed", r.get("frac_executed"), "traffic", r.get("traffic"), "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "enc ms", d["ms_encode_per_step"], "hits", d.get("sibling_hits"))
    for c in d["configs"]: print("  ", c["config"][:60], c["tokens_per_sec"], c["ms_per_iteration"], c["medusa_over_vanilla"], c["roofline_frac_hbm"], c["prefill_tflops"], c["prefill_frac_mfma"], c["parity_checked"], c.get("parity_strict"), str(c.get("parity_ties_followed"))[:80])
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["cores"], d["cpu_baseline"].get("threads_used"), d["cpu_baseline"]["parity_checked"])
    print("  sens", {k: (v["ms_per_iteration"], v.get("frac_hbm_executed")) for k, v in d["acceptance_sensitivity"].items()})
except Exception as e:
    print("bench failed", e, open("$O/bench.err").read()[-800:])
PY
