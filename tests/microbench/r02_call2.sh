#!/bin/bash
# round-2 GPU call 2: divergence diagnosis, bench without / with the side-stream prefetch, timelines, parity subset
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c2; mkdir -p $O
python whisper-medusa_amd/build.py > $O/build.log 2>&1; python whisper-medusa_amd/build.py --timeline >> $O/build.log 2>&1; tail -2 $O/build.log
echo "== diag new"; timeout 300 python tests/microbench/diag_m10block.py > $O/diag_new.log 2>&1; echo rc $?; tail -30 $O/diag_new.log
echo "== diag v1"; WM_ABI_ANY=1 WM_LIB=tests/microbench/libwm_v1.so timeout 300 python tests/microbench/diag_m10block.py > $O/diag_v1.log 2>&1; echo rc $?; tail -12 $O/diag_v1.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs"
echo "== bench new"; timeout 600 python bench.py $B > $O/bench_new.json 2> $O/bench_new.err; echo rc $?; python - <<PY
import json
for f in ("bench_new",):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1]); print(f, d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], d["vanilla_anchor"])
    except Exception as e: print(f, "failed", e, open("$O/"+f+".err").read()[-800:])
PY
echo "== bench prefetch"; WM_PREFETCH=1 timeout 600 python bench.py $B > $O/bench_pf.json 2> $O/bench_pf.err; echo rc $?; python - <<PY
import json
for f in ("bench_pf",):
    try:
        d=json.loads(open("$O/"+f+".json").read().strip().splitlines()[-1]); print(f, d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], d["vanilla_anchor"])
    except Exception as e: print(f, "failed", e, open("$O/"+f+".err").read()[-800:])
PY
echo "== timeline new"; WM_LIB=whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 600 python tests/microbench/timeline.py --out $O/timeline_new > $O/timeline_new.log 2>&1; echo rc $?; tail -28 $O/timeline_new.log
echo "== timeline prefetch"; WM_PREFETCH=1 WM_LIB=whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 600 python tests/microbench/timeline.py --out $O/timeline_pf > $O/timeline_pf.log 2>&1; echo rc $?; tail -28 $O/timeline_pf.log
echo "== pytest parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo rc $?; tail -15 $O/pytest.log
