#!/bin/bash
# round 3, GPU call 16: one-wave-per-row LayerNorm for one or two clips (the 16-row-group kernel is 96 blocks there)
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c16; mkdir -p $O
echo "== pytest (encoder: tiny shapes end to end, large-v2 big batch against one clip, prompt pass)"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "encoder_output or big_batch or prompt_pass or test_generate_api_end_to_end" > $O/pytest.log 2>&1; echo rc $?; tail -2 $O/pytest.log
for v in 256 0; do
  WM_ENC_LN_ROWS_BELOW=$v timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b1_$v.json 2> $O/b1_$v.err; echo rc $?
  python - <<PY
import json
d = json.loads(open("$O/b1_$v.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("rows_below=$v", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "prefill", r["prefill"]["achieved"], "enc ms", d.get("ms_encode_per_step"))
PY
done
