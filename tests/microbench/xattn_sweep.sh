#!/bin/bash
# Cross-attention launch-shape sweep at a given batch: bash tests/microbench/xattn_sweep.sh <batch>
# Prints the average duration of full-size k_attn_mfma<true,*> launches per setting.
B=${1:-32}
export TMPDIR=/tmp
for nt in ${NTS:-1 0}; do for t in ${TARGETS:-700 1300 1900 3800}; do
  export WM_XATTN_NT=$nt WM_XATTN_TARGET_BLOCKS=$t
  rm -rf /tmp/prof_x; (cd /tmp && timeout 300 rocprofv3 --kernel-trace -d /tmp/prof_x -- python /root/repo/bench.py --batch $B --steps 1 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nt', $nt, 'target', $t, 'ms/iter', d['roofline']['ms_per_launch'], 'tok/s', d['value'])")
  python /root/repo/tests/microbench/kernel_time.py $(ls /tmp/prof_x/*/*.db | head -1) "k_attn_mfma<true" ${2:-20}
done; done
