#!/bin/bash
# round 4, GPU call 4: staggered fp8 256 x 256 kernel (two wave halves half a step apart) against the lock-step one; packed-GELU FC1 epilogue (bf16 and fp8)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c4; mkdir -p $O
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -s \
  -k "fp8_mfma or encoder_output or big_batch or end_to_end_audio" > $O/pytest.log 2>&1; echo pytest rc $?; grep -i "max|d|\|passed\|failed\|error" $O/pytest.log | tail -14
for arm in bf16 fp8s fp8l; do
  F=""; unset WM_F8_STAGGER
  [ $arm = fp8s ] && F="--fp8-weights"
  [ $arm = fp8l ] && { F="--fp8-weights"; export WM_F8_STAGGER=0; }
  timeout 150 python bench.py $F --batch 32 --steps 2 --warmup 1 --max-new 32 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm b32 rc $?
done
unset WM_F8_STAGGER
python - <<PY
import json
for arm in ("bf16", "fp8s", "fp8l"):
    try:
        d = json.loads(open("$O/b32_%s.json" % arm).read().strip().splitlines()[-1]); r = d["roofline"]
        print(arm, "prefill TF/s", r["prefill"]["achieved"], "enc ms", d.get("ms_encode_per_step"))
    except Exception as e: print(arm, "failed", e)
PY
cd /tmp; export TMPDIR=/tmp
timeout 200 rocprofv3 --kernel-trace --stats -d /tmp/kt_fp8 -o fp8 -- python $GRAFT_REPO_ROOT/bench.py --fp8-weights --batch 32 --steps 1 --warmup 1 --max-new 16 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/prof_fp8.json 2> $O/prof_fp8.err; echo prof rc $?
cd "$GRAFT_REPO_ROOT"
DB=$(find /tmp/kt_fp8 -name "*.db" | head -1); python tests/prof_summary.py $DB $O/r04_kernel_trace_fp8_b32.md | head -16
