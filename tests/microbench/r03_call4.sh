#!/bin/bash
# round 3, GPU call 4: batched epilogues (encoder tiles, decoder token-tile GEMM), done-flag checks behind the first loads,
# LayerNorm-fused two-tile kernel; cross-attention block balance at 32 streams
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c4; mkdir -p $O
echo "== pytest batched (micro / tiny shapes)"; timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py -m gpu -q -x -p no:cacheprovider -k "many_streams or wide_batch or micro_batches or batch_equals or encoder_output or fp8 or long_prompts or greedy_mode or cross_kv or logmel or decode_tokens" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== pytest large"; timeout 1200 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -k "twelve or four_stream or block_decode or thirty_two or big_batch or greedy_equals or end_to_end" > $O/pytest_large.log 2>&1; echo rc $?; tail -4 $O/pytest_large.log
echo "== encoder (32 clips)"; timeout 400 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-dbg --out $O/enc_dbg.json > $O/enc_dbg.log 2>&1; echo rc $?; grep -E "^encoder" $O/enc_dbg.log
run() { tag=$1; b=$2; shift; shift; env "$@" timeout 300 python bench.py --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$tag", d["roofline"]["ms_per_launch"], "ms/iter", d["value"], "tok/s; tok/iter", d["tokens_per_iter"], "vanilla ms/step", va.get("ms_per_token_step"), "medusa/vanilla", va.get("medusa_over_vanilla"), "prefill TF", d["roofline"]["prefill"]["achieved"])
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-800:])
PY
}
run b32 32 A=1
run b32_x3840 32 WM_XATTN_TARGET_BLOCKS=3840
run b32_x1920 32 WM_XATTN_TARGET_BLOCKS=1920
run b32_nofuse 32 WM_SKINNY2_NORM=0
run b1 1 A=1
