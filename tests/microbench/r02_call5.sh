#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r02c6; mkdir -p $O
python whisper-medusa_amd/build.py --force > $O/build.log 2>&1; python whisper-medusa_amd/build.py --timeline >> $O/build.log 2>&1; tail -2 $O/build.log
B="--steps 6 --warmup 2 --no-cpu-baseline --no-extra-configs"
show() { python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); print("$1", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], d["vanilla_anchor"], "prefill", d["roofline"]["prefill"]["achieved"], "ms_enc", d["ms_encode_per_step"])
except Exception as e: print("$1", "failed", e, open("$O/$1.err").read()[-800:])
PY
}
echo "== bench"; timeout 600 python bench.py $B > $O/bench.json 2> $O/bench.err; echo rc $?; show bench
echo "== timeline"; WM_LIB=whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 600 python tests/microbench/timeline.py --out $O/timeline > $O/timeline.log 2>&1; echo rc $?; tail -12 $O/timeline.log
echo "== pytest parity subset"; timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider > $O/pytest.log 2>&1; echo rc $?; tail -5 $O/pytest.log

echo "== bench nofuse"; WM_FUSE_CQ=0 timeout 600 python bench.py $B > $O/bench_nofuse.json 2> $O/bench_nofuse.err; echo rc $?; show bench_nofuse
