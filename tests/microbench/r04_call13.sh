#!/bin/bash
# round 4, call 13 (diagnostic for the next round): where the 352-row decode GEMMs wait — L2 hits / misses and the wave-state counters of
# k_rows_gemm / k_ln_tiles on the layer microbenchmark (tests/microbench/r04_gemm_time.py), two counter passes
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c13; mkdir -p $O
cd /tmp
timeout 150 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace -d /tmp/l2 -o l2 -- python $R/tests/microbench/r04_gemm_time.py l2 > $O/l2.log 2>&1; echo l2 rc $?
DB=$(find /tmp/l2 -name "*.db" | head -1); python $R/tests/sq_summary.py $DB $O/r04_pmc_l2_rows_gemm.md "k_rows_gemm|k_ln_tiles|k_skinny2" | head -20
timeout 150 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES --kernel-trace -d /tmp/sq -o sq -- python $R/tests/microbench/r04_gemm_time.py sq > $O/sq.log 2>&1; echo sq rc $?
DB=$(find /tmp/sq -name "*.db" | head -1); python $R/tests/sq_summary.py $DB $O/r04_pmc_sq_rows_gemm.md "k_rows_gemm|k_ln_tiles|k_skinny2" | head -20
