#!/bin/bash
# round 2, GPU call 14: attention kernels with their early arguments in the preloaded kernarg range; prefetch job granularity;
# FETCH_SIZE without the prefetch blocks
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r02c14; mkdir -p $O
show() { python - <<PY
import json
try:
    d=json.loads(open("$O/$1.json").read().strip().splitlines()[-1]); va=d.get("vanilla_anchor") or {}
    print("$1", d["value"], "tok/s", d["roofline"]["ms_per_launch"], "ms/iter", d["tokens_per_iter"], "tok/iter; vanilla ms/step", va.get("ms_per_token_step"))
except Exception as e: print("$1", "failed", e, open("$O/$1.err").read()[-800:])
PY
}
B="--steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs"
echo "== default"; timeout 300 python bench.py $B > $O/b1.json 2> $O/b1.err; show b1
echo "== pf jobs 240"; WM_PF_JOBS=240 timeout 300 python bench.py $B > $O/b1_pf240.json 2> $O/b1_pf240.err; show b1_pf240
echo "== pf jobs 480"; WM_PF_JOBS=480 timeout 300 python bench.py $B > $O/b1_pf480.json 2> $O/b1_pf480.err; show b1_pf480
echo "== quick parity"; timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -p no:cacheprovider -k "decode_tokens_bit_exact or batch_equals or wide_batch" > $O/pytest.log 2>&1; echo rc $?; tail -2 $O/pytest.log
cd /tmp
echo "== pmc fetch b1, no prefetch blocks"
WM_PREFETCH=0 timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc0 -o pmc0 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc0.log 2>&1; echo rc $?
DB=$(find /tmp/pmc0 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r02_pmc_fetch_size_bench_b1_noprefetch.md | tail -2
cd $R
echo "== timeline"
WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_tl.so timeout 300 python tests/microbench/timeline.py --out $O/r02_timeline_final > $O/timeline.log 2>&1; echo rc $?; tail -6 $O/timeline.log; grep -E "self-attn \| 11|cross-attn \| 11|FC2 \| 11" $O/r02_timeline_final.md
