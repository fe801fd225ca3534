#!/bin/bash
# round 2, GPU call 16: final numbers on the final decoder code — default bench (CPU baseline + extra configs), kernel trace, PMC traffic
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c16; mkdir -p $O
echo "== default bench"; ( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; python - <<PY
import json
d=json.loads(open("$O/bench_default.json").read().strip().splitlines()[-1]); print(d["value"], d["roofline"]["ms_per_launch"], d["roofline"]["frac"], d["vanilla_anchor"], d["cpu_baseline"]["value"], d["cpu_baseline"]["parity_checked"])
PY
cd /tmp
echo "== kernel trace b1"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r02_kernel_trace_bench_b1.md | tail -2
echo "== pmc fetch b1"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r02_pmc_fetch_size_bench_b1.md $O/r02_pmc_traffic.json | tail -2
echo "== pmc fetch b1 without prefetch blocks"
WM_PREFETCH=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc0 -o pmc0 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc0.log 2>&1; echo rc $?
DB=$(find /tmp/pmc0 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r02_pmc_fetch_size_bench_b1_noprefetch.md | tail -2
