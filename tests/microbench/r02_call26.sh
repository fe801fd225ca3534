#!/bin/bash
# round 2, GPU call 26: generate()-level tests with the automatic micro-batch policy (2-3 clips -> one context per clip)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c26; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py -m gpu -q -x -p no:cacheprovider -k "generate or micro_batches or streamer or features or longform or language or sharded or packed or pretrained or stereo or silence" > $O/pytest.log 2>&1; echo "rc $?"; tail -4 $O/pytest.log
