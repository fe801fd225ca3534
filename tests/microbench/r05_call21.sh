#!/bin/bash
# round 5, GPU call 21: HBM traffic counters re-taken (bench.py refuses a traffic file whose decode-source hash differs: a comment changed after the closing call)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c21; mkdir -p $O
cd /tmp
WM_PREFETCH=0 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc2 -o pmc2 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc2.log 2>&1; echo rc $?
DB=$(find /tmp/pmc2 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r05_pmc_fetch_size_bench_b1_noprefetch.md $O/r05_pmc_traffic_noprefetch.json | tail -1
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r05_pmc_fetch_size_bench_b1.md $O/r05_pmc_traffic.json $O/r05_pmc_traffic_noprefetch.json | tail -1
cd $R
cp $O/r05_pmc_traffic.json profiles/r05_pmc_traffic.json
timeout 300 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']
print(d['value'], r['ms_per_launch'], r['frac'], 'traffic', r['traffic'], r['traffic_source'].get('stale'))"
