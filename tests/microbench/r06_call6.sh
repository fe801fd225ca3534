#!/bin/bash
# round 6, GPU call 6: the whole GPU suite on the new default (fp16 single-plane contract; both contracts in tests/test_gpu_act.py; cross_kv_fp8
# tests), then configs[4] with the decode loop on the fp8 cross-K/V copy against the bf16 cache (same library), and the Block leg
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c6; mkdir -p $O
( time timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu.log | cut -c1-300 | tail -20
grep -h "parity ties" $O/pytest_gpu.log | cut -c1-600
grep -h "rel to scale\|max rel" $O/pytest_gpu.log | cut -c1-300 | tail -12
for rep in 1 2; do
for arm in "fp8" "fp8 --bf16-cross-kv" "block"; do
  set -- $arm
  extra="--fp8-weights"; [ "$1" = block ] && extra="--heads block"; [ -n "$2" ] && extra="$extra $2"
  tag=$(echo $arm | tr -d ' -')
  timeout 300 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs $extra > $O/b32_$tag$rep.json 2> $O/b32_$tag$rep.err
  python - <<PY
import json
try:
    d = json.loads(open("$O/b32_$tag$rep.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "vanilla", v["ms_per_token_step"], "ratio", v["medusa_over_vanilla"], "tok/iter", d["tokens_per_iter"], "steps/iter", r["passes_per_iteration"], "prefill", r["prefill"]["achieved"], flush=True)
except Exception as e: print("$arm", "failed", e, open("$O/b32_$tag$rep.err").read()[-500:])
PY
done; done 2>&1 | tee $O/bench.log
