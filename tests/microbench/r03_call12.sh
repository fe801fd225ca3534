#!/bin/bash
# round 3, GPU call 12: residual GEMMs with the accumulators initialised from the residual tile; flash-attention variants (groups of 4,
# three blocks per CU) — parity tests that cover the encoder, then timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c12; mkdir -p $O
echo "== pytest (encoder, fp8 encoder, logits, tokens: tiny shapes)"
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "encoder or logits or bit_exact or end_to_end or fp8" > $O/pytest_parity.log 2>&1; echo rc $?; tail -3 $O/pytest_parity.log
echo "== pytest (large-v2: big-batch encoder, prompt pass, batch consistency, fp8)"
timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "big_batch or prompt_pass or greedy_equals or twelve_streams or fp8_mfma or linear_decode" -s > $O/pytest_large.log 2>&1; echo rc $?; grep -i "max|d|\|passed\|failed" $O/pytest_large.log | tail -8
echo "== encoder (32 clips)"
timeout 600 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-r3b --out $O/enc.json > $O/enc.log 2>&1; echo rc $?; grep "^encoder" $O/enc.log
