#!/bin/bash
# round 3, GPU call 6: candidate trees beyond 16 nodes (query tiles), encoder GEMM with 256 x 128 tiles / two blocks per CU
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c6; mkdir -p $O
echo "== pytest trees"; timeout 900 python -m pytest tests/test_gpu_tree.py -m gpu -q -x -p no:cacheprovider > $O/pytest_tree.log 2>&1; echo rc $?; tail -4 $O/pytest_tree.log
echo "== pytest parity (decode, single tile paths untouched?)"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "decode_tokens or carry or batch_equals or many_streams or forward" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== pytest large tree + linear"; timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -q -x -p no:cacheprovider -k "candidate_tree or linear_decode_loop or big_batch" > $O/pytest_large.log 2>&1; echo rc $?; tail -4 $O/pytest_large.log
echo "== encoder (32 clips)"; timeout 400 python tests/microbench/r03_sweep.py --enc --enc-only-batch --enc-knobs --out $O/enc.json > $O/enc.log 2>&1; echo rc $?; grep -E "^encoder" $O/enc.log
run() { tag=$1; b=$2; shift; shift; env "$@" timeout 300 python bench.py --batch $b --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/$tag.json 2> $O/$tag.err
python - <<PY
import json
try:
    d=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$tag", d["roofline"]["ms_per_launch"], "ms/iter", d["value"], "tok/s; tok/iter", d["tokens_per_iter"], "vanilla ms/step", va.get("ms_per_token_step"), "medusa/vanilla", va.get("medusa_over_vanilla"), "prefill TF", d["roofline"]["prefill"]["achieved"])
except Exception as e: print("$tag failed", e, open("$O/$tag.err").read()[-800:])
PY
}
run b1 1 A=1
