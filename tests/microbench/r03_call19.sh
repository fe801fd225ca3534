#!/bin/bash
# round 3, GPU call 19 (measurement only, final code): result-preserving knobs of the pipelined GEMM re-checked (ring depth, persistence, tile width)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03c19; mkdir -p $O
timeout 200 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-knobs --out $O/enc_knobs.json > $O/enc_knobs.log 2>&1; echo rc $?; grep "^encoder" $O/enc_knobs.log
for r in 3 5; do WM_ENC_GEMM_RING=$r timeout 100 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-default --out $O/enc_ring$r.json > $O/enc_ring$r.log 2>&1; echo "ring $r: $(grep '^encoder' $O/enc_ring$r.log)"; done
