#!/bin/bash
# round 3, GPU call 18 (measurement only, final code): the pipelined GEMM with its parts switched off one at a time, and the SQ wait-state
# counters of the 32-clip encoder pass
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c18; mkdir -p $O
echo "== parts"
timeout 200 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-dbg --out $O/enc_parts.json > $O/enc_parts.log 2>&1; echo rc $?; grep "^encoder" $O/enc_parts.log
cd /tmp
echo "== SQ counters, encoder 32 clips"
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_LDS --kernel-trace -d /tmp/sq -o sq -- python $R/tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-default --out $O/enc_sq.json > $O/sq.log 2>&1; echo rc $?
DB=$(find /tmp/sq -name "*.db" | head -1); python $R/tests/sq_summary.py $DB $O/r03_pmc_sq_encoder_b32.md | tail -2; head -24 $O/r03_pmc_sq_encoder_b32.md
