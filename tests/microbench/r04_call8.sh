#!/bin/bash
# round 4, GPU call 8: 96-row tiles for badly filling grids of the few-clip encoder GEMMs (one clip: QKV 360 -> 480 tiles on 512 slots)
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c8; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -s -k "encoder_output or cross_kv or big_batch or end_to_end_audio or logmel" > $O/pytest.log 2>&1; echo pytest rc $?; grep -i "max|d|\|passed\|failed\|error" $O/pytest.log | tail -8
for arm in on off; do
  if [ $arm = off ]; then export WM_ENC_BM96=0; else unset WM_ENC_BM96; fi
  for B in 1 2; do
    timeout 100 python bench.py --batch $B --steps 4 --warmup 1 --max-new 32 --no-cpu-baseline --no-extra-configs --no-vanilla > $O/b${B}_$arm.json 2> $O/b${B}_$arm.err; echo $arm b$B rc $?
  done
done
unset WM_ENC_BM96
python - <<PY
import json
for arm in ("on", "off"):
    for B in (1, 2):
        try:
            d = json.loads(open("$O/b%d_%s.json" % (B, arm)).read().strip().splitlines()[-1]); r = d["roofline"]
            print("bm96", arm, "clips", B, "prefill TF/s", r["prefill"]["achieved"], "enc ms", d.get("ms_encode_per_step"))
        except Exception as e: print(arm, B, "failed", e)
PY
