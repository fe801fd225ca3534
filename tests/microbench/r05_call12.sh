#!/bin/bash
# round 5, GPU call 12: the small operands of the next launch (LayerNorm gamma | beta, bias) touched by the in-launch prefetch blocks
# (WM_PREFETCH_SMALL=0/1): one-stream bench per setting, interleaved twice; bit-exactness tests on the default
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r05c12; mkdir -p $O
for arm in off on off on; do
  if [ $arm = off ]; then export WM_PREFETCH_SMALL=0; else unset WM_PREFETCH_SMALL; fi
  timeout 200 python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_$arm.json 2> $O/b1_$arm.err; echo $arm rc $?
  python - <<PY
import json
try:
    d = json.loads(open("$O/b1_$arm.json").read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
    print("$arm", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "vanilla ms/tok", v.get("ms_per_step"), v.get("tokens_per_sec"), "ratio", v["medusa_over_vanilla"], "tok/it", d["tokens_per_iter"])
except Exception as e: print("$arm", "failed", e)
PY
done
unset WM_PREFETCH_SMALL
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "logits or bit_exact or carry or batch" 2>&1 | tail -3
