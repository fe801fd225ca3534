#!/bin/bash
# round 3, GPU call 13: strip-major tile order of the pipelined GEMM (compact patches for every GEMM shape) — encoder parity, timings
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c13; mkdir -p $O
echo "== pytest (encoder: tiny shapes, large-v2 big batch, prompt pass)"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "encoder_output or big_batch or prompt_pass or fp8_mfma_encoder_and" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== encoder (32 clips)"
timeout 600 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-order --out $O/enc.json > $O/enc.log 2>&1; echo rc $?; grep "^encoder" $O/enc.log
