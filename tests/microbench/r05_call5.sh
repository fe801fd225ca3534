#!/bin/bash
# round 5, call 5: the hybrid arm of the k_rows_lds experiment (only the wide GEMMs — QKV, FC1: N16 >= 160 — on k_rows_lds with 8 waves x 1 row tile, the four narrow ones
# on the shipped k_rows_gemm) in the merged-step bench, against the shipped library in the same call; the new host-processor GPU tests
cd "$GRAFT_REPO_ROOT" || exit 1
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r05c5; mkdir -p $O
timeout 300 python -m pytest tests/test_gpu_features.py -m gpu -q -p no:cacheprovider -k "arbitrary" > $O/pytest_proc.log 2>&1; echo pytest rc $?; tail -3 $O/pytest_proc.log
for arm in shipped hybrid81 hybrid_auto; do
  unset WM_LIB WM_ROWS_LDS WM_LN_TAIL WM_RL_MIN_N16 WM_RL_NW WM_RL_FT
  case $arm in
    hybrid81) export WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_rowslds.so WM_ROWS_LDS=1 WM_LN_TAIL=0 WM_RL_MIN_N16=160 WM_RL_NW=8 WM_RL_FT=1;;
    hybrid_auto) export WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_rowslds.so WM_ROWS_LDS=1 WM_LN_TAIL=0 WM_RL_MIN_N16=160;;
  esac
  timeout 240 python bench.py --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm rc $?
done
unset WM_LIB WM_ROWS_LDS WM_LN_TAIL WM_RL_MIN_N16 WM_RL_NW WM_RL_FT
python - <<PY
import json
for arm in ("shipped", "hybrid81", "hybrid_auto"):
    try:
        d = json.loads(open("$O/b32_%s.json" % arm).read().strip().splitlines()[-1]); r = d["roofline"]
        print(arm, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "decode tok/s", d["decode_tokens_per_sec_per_gpu"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"], "hist", d.get("accept_hist"))
    except Exception as e: print(arm, "failed", e)
PY
