#!/bin/bash
# round 4, GPU call 10: merged-step schedule at several streams (one pass per step) against the lock-step iteration: parity, 32-stream iteration, sensitivity
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c10; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "merged_step or batch or streams or carry or micro_batches or bit_exact" > $O/pytest.log 2>&1; echo pytest rc $?; tail -5 $O/pytest.log
for arm in step lock; do
  if [ $arm = lock ]; then export WM_NO_STEP=1; else unset WM_NO_STEP; fi
  timeout 200 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm rc $?
done
unset WM_NO_STEP
python - <<PY
import json
for arm in ("step", "lock"):
    try:
        d = json.loads(open("$O/b32_%s.json" % arm).read().strip().splitlines()[-1]); r = d["roofline"]
        print(arm, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "decode tok/s", d["decode_tokens_per_sec_per_gpu"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "tok/it", d["tokens_per_iter"])
        print("   sens", {k: v["ms_per_iteration"] for k, v in d.get("acceptance_sensitivity", {}).items()})
    except Exception as e: print(arm, "failed", e)
PY
