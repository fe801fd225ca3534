#!/bin/bash
# round 4, call 15 (diagnostic for the next round): in-kernel timeline of the merged-step schedule's 352-row launches at 32 streams (libwm_tl.so,
# k_rows_gemm probes: MFMAs issued / K-slice partials exchanged / exit), then the HBM traffic counters on the final decode sources (product library)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c15; mkdir -p $O
WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 150 python tests/microbench/timeline.py --batch 32 --max-new 20 --out $O/r04_timeline_b32 > $O/tl.log 2>&1; echo timeline rc $?; tail -28 $O/tl.log | cut -c1-220
rm -f $O/r04_timeline_b32.npz
cd /tmp
echo "== pmc fetch b1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r04_pmc_fetch_size_bench_b1.md $O/r04_pmc_traffic.json | tail -1
