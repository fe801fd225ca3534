#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r02c7; mkdir -p $O
python whisper-medusa_amd/build.py --force > $O/build.log 2>&1; tail -1 $O/build.log
cd /tmp
echo "== pmc SQ encoder b32"
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS --kernel-trace -d /tmp/pmc_sq -o pmc_sq -- python $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc_sq.log 2>&1; echo rc $?
DB=$(find /tmp/pmc_sq -name "*.db" | head -1); echo $DB
python $GRAFT_REPO_ROOT/tests/sq_summary.py $DB $O/r02_pmc_sq_bench_b32.md
echo "== kernel trace b32"
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt32 -o kt32 -- python $GRAFT_REPO_ROOT/bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/kt32.log 2>&1; echo rc $?
DB=$(find /tmp/kt32 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tests/prof_summary.py $DB $O/r02_kernel_trace_bench_b32_pre.md | head -40
