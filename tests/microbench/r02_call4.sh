#!/bin/bash
# round-2 GPU call 4: in-launch prefetch on/off, kernarg preload on/off, timeline, parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c4; mkdir -p $O
python whisper-medusa_amd/build.py --force > $O/build.log 2>&1; python whisper-medusa_amd/build.py --timeline >> $O/build.log 2>&1; tail -2 $O/build.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs"
show() { python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], d["vanilla_anchor"])
except Exception as e: print("$1", "failed", e, open("$O/$1.err").read()[-800:])
PY
}
echo "== bench prefetch on (default)"; timeout 600 python bench.py $B > $O/bench_pf.json 2> $O/bench_pf.err; echo rc $?; show bench_pf
echo "== bench prefetch off"; WM_PREFETCH=0 timeout 600 python bench.py $B > $O/bench_nopf.json 2> $O/bench_nopf.err; echo rc $?; show bench_nopf
echo "== timeline prefetch on"; WM_LIB=whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 600 python tests/microbench/timeline.py --out $O/timeline_pf > $O/timeline_pf.log 2>&1; echo rc $?; tail -28 $O/timeline_pf.log
echo "== pytest parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo rc $?; tail -15 $O/pytest.log
echo "== rebuild without kernarg preload"; cp whisper-medusa_amd/whisper_medusa/libwm.so /tmp/libwm_pre.so
WM_NO_KERNARG_PRELOAD=1 python whisper-medusa_amd/build.py --force > $O/build2.log 2>&1; tail -1 $O/build2.log
echo "== bench no-preload, prefetch on"; timeout 600 python bench.py $B > $O/bench_nopre.json 2> $O/bench_nopre.err; echo rc $?; show bench_nopre
cp /tmp/libwm_pre.so whisper-medusa_amd/whisper_medusa/libwm.so
