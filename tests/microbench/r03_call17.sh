#!/bin/bash
# round 3, GPU call 17: one-clip encoder, the tile-shape knobs of round 2 re-swept on the round-3 epilogues (one process per setting)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r03c17; mkdir -p $O
i=0
for env in "X=0" "WM_ENC_BM64_BELOW=400" "WM_ENC_BM64_BELOW=600" "WM_ENC_GEMM_KSPLIT=1" "WM_FLASH_QT1_BELOW=0" "WM_ENC_BM64_BELOW=100"; do
  i=$((i+1))
  env $env timeout 120 python tests/microbench/r03_sweep.py --enc --enc-one-clip --out $O/enc_$i.json > $O/enc_$i.log 2>&1
  echo "$env: $(grep '^encoder' $O/enc_$i.log)"
done
