#!/bin/bash
# round 2, GPU call 9: candidate-tree parity + regression of the chain parity suite + single-stream bench
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_tree.py -x -q -m gpu -s > gpurun_out/c9_tree.log 2>&1; echo "tree rc=$?" >> gpurun_out/c9_tree.log
tail -25 gpurun_out/c9_tree.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/c9_parity.log 2>&1; echo "parity rc=$?" >> gpurun_out/c9_parity.log
tail -5 gpurun_out/c9_parity.log
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > gpurun_out/c9_bench.log 2>&1
tail -2 gpurun_out/c9_bench.log
