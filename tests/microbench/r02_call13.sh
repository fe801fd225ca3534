#!/bin/bash
# round 2, GPU call 13: final measurements — default bench (with CPU baseline + extra configs), kernel traces, PMC traffic, timeline
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c13; mkdir -p $O
echo "== default bench"; ( time timeout 900 python bench.py > $O/bench_default.json 2> $O/bench_default.err ) 2>&1 | grep real; tail -c 600 $O/bench_default.json | head -c 600; echo
cd /tmp
echo "== kernel trace b1"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r02_kernel_trace_bench_b1.md | tail -3
echo "== pmc fetch b1"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r02_pmc_fetch_size_bench_b1.md $O/r02_pmc_traffic.json | tail -2
echo "== kernel trace b32"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r02_kernel_trace_bench_b32.md | tail -2
echo "== timeline"
cd $R
ls -la whisper-medusa_amd/whisper_medusa/libwm_tl.so
WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 300 python tests/microbench/timeline.py --out $O/r02_timeline_final > $O/timeline.log 2>&1; echo rc $?; tail -14 $O/timeline.log
