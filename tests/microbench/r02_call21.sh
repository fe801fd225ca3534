#!/bin/bash
# round 2, GPU call 21: MFMA-busy counters of the encoder at 32 clips, bf16 and fp8 paths
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c21; mkdir -p $O
cd /tmp
for v in bf16 fp8; do
  X=""; [ $v = fp8 ] && X="--fp8-weights"
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace -d /tmp/mf_$v -o mf -- python $R/bench.py --batch 32 --steps 1 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs $X > $O/mf_$v.log 2>&1; echo "$v rc $?"
  DB=$(find /tmp/mf_$v -name "*.db" | head -1); python $R/tests/mfma_summary.py $DB $O/r02_pmc_mfma_busy_bench_b32_$v.md 2>/dev/null | head -14
done
