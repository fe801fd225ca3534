#!/bin/bash
# round 2, GPU call 15: token-operand loads unconditional (one memory batch per GEMM launch instead of two dependent rounds)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c15; mkdir -p $O
show() { python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$1", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], "tok/iter; vanilla ms/step", va.get("ms_per_token_step"), "frac", d["roofline"]["frac"])
except Exception as e: print("$1", "failed", e, open("$O/$1.err").read()[-800:])
PY
}
B="--steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs"
echo "== b1"; timeout 300 python bench.py $B > $O/b1.json 2> $O/b1.err; show b1
echo "== parity"; timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_tree.py tests/test_gpu_features.py -m gpu -q -x -p no:cacheprovider -k "not fp32_oracle_end_to_end" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
echo "== b32"; timeout 300 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32.json 2> $O/b32.err; show b32
echo "== timeline"
WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 300 python tests/microbench/timeline.py --out $O/r02_timeline_final > $O/timeline.log 2>&1; echo rc $?; tail -4 $O/timeline.log; grep -E " \| 11 \|" $O/r02_timeline_final.md
