#!/bin/bash
# round 6, GPU call 12: the round's profile set on the current sources (fp16 single-plane contract, folded LayerNorm, fp8 cross-K/V):
# kernel traces at 1 and 32 streams (bf16 and fp8 legs), HBM traffic counters at one stream (with / without prefetch blocks), matrix-pipe busy at 32 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06c12; mkdir -p $O
cd /tmp
B="--no-cpu-baseline --no-vanilla --no-extra-configs"
echo "== kernel trace b1"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt1 -- python $R/bench.py --steps 4 --warmup 1 $B > $O/kt1.log 2>&1; echo rc $?
DB=$(find /tmp/kt1 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r06_kernel_trace_bench_b1.md --hbm-large-v2-b1 | tail -12
echo "== kernel trace b32"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $R/bench.py --batch 32 --steps 2 --warmup 1 $B > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r06_kernel_trace_bench_b32.md | tail -1
echo "== kernel trace b32 fp8"
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt32f -o kt32f -- python $R/bench.py --batch 32 --steps 2 --warmup 1 --fp8-weights $B > $O/kt32f.log 2>&1; echo rc $?
DB=$(find /tmp/kt32f -name "*.db" | head -1); python $R/tests/prof_summary.py $DB $O/r06_kernel_trace_bench_b32_fp8.md | tail -1
echo "== pmc fetch b1, no prefetch blocks"
WM_PREFETCH=0 timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc2 -o pmc2 -- python $R/bench.py --steps 2 --warmup 1 $B > $O/pmc2.log 2>&1; echo rc $?
DB=$(find /tmp/pmc2 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r06_pmc_fetch_size_bench_b1_noprefetch.md $O/r06_pmc_traffic_noprefetch.json | tail -1
echo "== pmc fetch b1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 $B > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r06_pmc_fetch_size_bench_b1.md $O/r06_pmc_traffic.json $O/r06_pmc_traffic_noprefetch.json | tail -1
echo "== mfma busy b32"
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/mf -o mf -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --max-new 16 $B > $O/mf.log 2>&1; echo rc $?
DB=$(find /tmp/mf -name "*.db" | head -1); python $R/tests/mfma_summary.py $DB $O/r06_pmc_mfma_busy_bench_b32.md 2>/dev/null | head -14
cd $R
echo "== 32-stream Linear leg alone (parity detail of stream 0)"
timeout 600 python - <<PY 2>&1 | tail -5
import json, sys, torch
sys.path.insert(0, "$R"); sys.path.insert(0, "$R/whisper-medusa_amd")
import bench
o = bench.extra_config("configs[1] shape at 32 streams (Medusa-Linear)", "base_head", 32, False, torch.device("cuda", 0), 4.5, 128, steps=2, f16=True)
print(json.dumps({k: v for k, v in o.items() if k.startswith("parity") or k in ("ms_per_iteration", "medusa_over_vanilla")}))
PY
ls $O
