#!/bin/bash
# round 6, GPU call 37: SQ counters of the encoder kernels at 32 clips on the final library (where do the flash-attention and GEMM waves wait?)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/r06c37; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/sq1 -o sq1 -- python $GRAFT_REPO_ROOT/tests/microbench/r06_enc_time.py --reps 2 > $O/sq1.log 2>&1; echo rc $?
DB=$(find /tmp/sq1 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tests/sq_summary.py $DB $O/r06_pmc_sq_encoder_b32.md "k_gemm_256p|k_flash|k_enc_ln" | cut -c1-400
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES --kernel-trace -d /tmp/sq2 -o sq2 -- python $GRAFT_REPO_ROOT/tests/microbench/r06_enc_time.py --reps 2 > $O/sq2.log 2>&1; echo rc $?
DB=$(find /tmp/sq2 -name "*.db" | head -1); python $GRAFT_REPO_ROOT/tests/sq_summary.py $DB $O/r06_pmc_sq2_encoder_b32.md "k_gemm_256p|k_flash|k_enc_ln" | cut -c1-400
