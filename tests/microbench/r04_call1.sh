#!/bin/bash
# round 4, GPU call 1: (a) the five prepared patches (libwm_r4p.so, WM_EP_WAIT_ONCE) against the round-3 library: 32-stream iteration,
# single-stream iteration, prefill at 32 clips / one clip; (b) f8f6f4 MFMA probe; (c) XCD-hierarchical grid barrier + 56 KB gather microbench
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c1; mkdir -p $O
PKG=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
timeout 60 tests/microbench/bin/mfma128_probe > $O/mfma128.txt 2>&1; echo probe rc $?; tail -30 $O/mfma128.txt
timeout 120 tests/microbench/bin/xcd_barrier > $O/xcd_barrier.txt 2>&1; echo barrier rc $?; cat $O/xcd_barrier.txt
for arm in base r4p; do
  if [ $arm = base ]; then unset WM_LIB; else export WM_LIB=$PKG/libwm_r4p.so; fi
  timeout 150 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm b32 rc $?
  timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b1_$arm.json 2> $O/b1_$arm.err; echo $arm b1 rc $?
done
python - <<PY
import json
for arm in ("base", "r4p"):
    for b in ("b32", "b1"):
        try:
            d = json.loads(open("$O/%s_%s.json" % (b, arm)).read().strip().splitlines()[-1]); r = d["roofline"]
            print(arm, b, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "prefill TF/s", r["prefill"]["achieved"], "enc ms", d.get("ms_encode_per_step"), "vanilla", d.get("vanilla_tokens_per_s"))
        except Exception as e: print(arm, b, "failed", e)
PY
export WM_LIB=$PKG/libwm_r4p.so
timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "encoder or bit_exact or batch or streams or wide or fp8_mfma" > $O/pytest.log 2>&1; echo pytest rc $?; tail -3 $O/pytest.log
