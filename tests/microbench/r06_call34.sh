#!/bin/bash
# round 6, GPU call 34: fp8 bytes widened by v_cvt_scalef32_pk_{bf16,f16}_fp8 (one instruction per pair) against the two-step form (libwm_prev.so = the commit
# before): fp8 tests, then the fp8 legs at 32 streams and one stream, interleaved twice
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c34; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
( time timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider -k "fp8" > $O/pytest_fp8.log 2>&1 ) 2>&1 | grep real
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_fp8.log | cut -c1-300 | tail -5
for rep in 1 2; do
  for lib in f16 prev; do
    WM_LIB_F16=$P/libwm_$lib.so timeout 400 python bench.py --batch 32 --fp8-weights --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b32_${lib}_$rep.json 2> $O/b32_${lib}_$rep.err
    WM_LIB_F16=$P/libwm_$lib.so timeout 400 python bench.py --fp8-weights --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_${lib}_$rep.json 2> $O/b1_${lib}_$rep.err
    python - <<PY
import json
for tag in ("b32", "b1"):
    d = json.loads(open("$O/%s_${lib}_$rep.json" % tag).read().strip().splitlines()[-1])
    r = d["roofline"]
    print("[$lib]", tag, "fp8", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "vanilla", d["vanilla_anchor"]["ms_per_token_step"], "hist", d["accept_hist"][:4])
PY
  done
done
