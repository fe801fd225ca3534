#!/bin/bash
# Encoder GEMM staging / K-split sweep: bash tests/microbench/enc_sweep.sh <batch> "<WM_ENC_GEMM_KSPLIT values>" "<WM_ENC_GEMM_STAGES values>"
B=${1:-1}
export TMPDIR=/tmp
for ks in ${2:-2}; do for stg in ${3:-2}; do
  export WM_ENC_GEMM_KSPLIT=$ks WM_ENC_GEMM_STAGES=$stg
  rm -rf /tmp/prof_e; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_e -- python /root/repo/bench.py --batch $B --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ksplit', $ks, 'stages', $stg, 'enc ms', d['ms_encode_per_step'], 'prefill TF/s', d['roofline']['prefill']['achieved'], 'tok/s', d['value'])")
  python /root/repo/tests/prof_summary.py $(ls /tmp/prof_e/*/*.db | head -1) | grep -E "flash|gemm_tiled" | cut -c1-110
done; done
