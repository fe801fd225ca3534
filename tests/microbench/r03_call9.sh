#!/bin/bash
# round 3, GPU call 9: profiles of the round's code — kernel traces (1 and 32 streams), PMC traffic (with / without prefetch blocks),
# MFMA-busy of the encoder, the CPU baseline over the whole 128-token budget; re-run of the tests touched after call 8
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c9; mkdir -p $O
echo "== pytest (api, trees)"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tree.py tests/test_bench_dist.py -m gpu -q -p no:cacheprovider -k "generate_api or tree_decode_tokens or streamer or bench_two_ranks" > $O/pytest.log 2>&1; echo rc $?; tail -4 $O/pytest.log
cd /tmp
echo "== kernel trace b1"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r03_kernel_trace_bench_b1.md | tail -2
echo "== kernel trace b32"
timeout 600 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r03_kernel_trace_bench_b32.md | tail -2
echo "== pmc fetch b1"
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r03_pmc_fetch_size_bench_b1.md $O/r03_pmc_traffic.json | tail -2
echo "== pmc fetch b1 without prefetch blocks"
WM_PREFETCH=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc0 -o pmc0 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc0.log 2>&1; echo rc $?
DB=$(find /tmp/pmc0 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r03_pmc_fetch_size_bench_b1_noprefetch.md | tail -2
echo "== mfma busy b32"
timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/mf -o mf -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/mf.log 2>&1; echo rc $?
DB=$(find /tmp/mf -name "*.db" | head -1); python $R/tests/mfma_summary.py $DB $O/r03_pmc_mfma_busy_bench_b32.md 2>/dev/null | head -14
echo "== cpu baseline, whole budget"
cd $R; ( time timeout 1500 python bench.py --steps 2 --warmup 1 --cpu-full --no-extra-configs > $O/r03_bench_cpu_full.json 2> $O/cpu_full.err ) 2>&1 | grep real
python - <<PY
import json
try:
    d=json.loads(open("$O/r03_bench_cpu_full.json").read().strip().splitlines()[-1]); print(d["cpu_baseline"])
except Exception as e: print("failed", e, open("$O/cpu_full.err").read()[-800:])
PY
