#!/bin/bash
# round 4, GPU call 2: fp8 encoder GEMMs on v_mfma_f32_16x16x128_f8f6f4 (parity vs the fp8 oracle, prefill against bf16 at 32 clips and one clip);
# forward() keyword surface on the GPU; 53-node tree; natural-EOS run at large-v2
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c2; mkdir -p $O
timeout 500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_features.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -s \
  -k "fp8_mfma or packed_fp8 or forward_with_encoder_outputs or natural_eos" > $O/pytest_fp8.log 2>&1; echo pytest fp8 rc $?; grep -i "max|d|\|passed\|failed\|error\|natural" $O/pytest_fp8.log | tail -12
timeout 300 python -m pytest tests/test_gpu_tree.py -m gpu -q -p no:cacheprovider -x -k "micro1442" > $O/pytest_tree.log 2>&1; echo pytest tree rc $?; tail -3 $O/pytest_tree.log
for arm in bf16 fp8; do
  F=""; [ $arm = fp8 ] && F="--fp8-weights"
  timeout 150 python bench.py $F --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm b32 rc $?
  timeout 100 python bench.py $F --steps 3 --warmup 1 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b1_$arm.json 2> $O/b1_$arm.err; echo $arm b1 rc $?
done
python - <<PY
import json
for arm in ("bf16", "fp8"):
    for b in ("b32", "b1"):
        try:
            d = json.loads(open("$O/%s_%s.json" % (b, arm)).read().strip().splitlines()[-1]); r = d["roofline"]
            print(arm, b, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "prefill TF/s", r["prefill"]["achieved"], "enc ms", d.get("ms_encode_per_step"), "frac_executed", r.get("frac_executed"))
        except Exception as e: print(arm, b, "failed", e)
PY
