#!/bin/bash
# round 3, GPU call 14 (final code): HBM traffic counters and kernel traces of the bench commands, the default bench line, then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c14; mkdir -p $O
cd /tmp
echo "== pmc fetch b1"
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r03_pmc_fetch_size_bench_b1.md $O/r03_pmc_traffic.json | tail -2
cp $O/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json
echo "== kernel trace b1"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r03_kernel_trace_bench_b1.md | tail -1
echo "== kernel trace b32"
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r03_kernel_trace_bench_b32.md | tail -1
echo "== mfma busy b32"
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/mf -o mf -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/mf.log 2>&1; echo rc $?
DB=$(find /tmp/mf -name "*.db" | head -1); python $R/tests/mfma_summary.py $DB $O/r03_pmc_mfma_busy_bench_b32.md 2>/dev/null | head -12
cd $R
echo "== bench default"
( time timeout 600 python bench.py > $O/r03_bench_default.json 2> $O/bench.err ) 2>&1 | grep real
python - <<PY
import json
try:
    d = json.loads(open("$O/r03_bench_default.json").read().strip().splitlines()[-1])
    r = d["roofline"]
    print("b1", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "traffic", r.get("traffic"), "prefill", r["prefill"]["achieved"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"])
    for c in d["configs"]: print("  ", c["config"][:60], c["tokens_per_sec"], c["ms_per_iteration"], c["medusa_over_vanilla"], c["roofline_frac_hbm"], c["prefill_tflops"], c["parity_checked"])
    print("  cpu", d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_checked"])
except Exception as e:
    print("bench failed", e, open("$O/bench.err").read()[-800:])
PY
echo "== pytest -m gpu (whole suite)"
timeout 1000 python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/pytest_full.log 2>&1; echo rc $?; tail -6 $O/pytest_full.log
