#!/bin/bash
# round 4, call 16: LayerNorm in the tail of the producing GEMM (LN2 behind out-proj, LN3 behind cross-out: libwm_lntail.so, build.py --variant lntail
# -DWM_LN_TAIL) against the product library: bench at 32 streams, parity on the variant; then the HBM traffic counters on the final decode sources
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c16; mkdir -p $O
for arm in lntail; do
  if [ $arm = lntail ]; then export WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_lntail.so; else unset WM_LIB; fi
  timeout 200 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm rc $?
done
unset WM_LIB
python - <<PY
import json
for arm in ("base", "lntail"):
    try:
        d = json.loads(open("$O/b32_%s.json" % arm).read().strip().splitlines()[-1]); r = d["roofline"]
        print(arm, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "decode tok/s", d["decode_tokens_per_sec_per_gpu"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"])
    except Exception as e: print(arm, "failed", e)
PY
WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_lntail.so timeout 150 python -m pytest tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -k "one_stream_of_a_four_stream_batch" > $O/pytest_lntail.log 2>&1; echo pytest lntail rc $?; tail -2 $O/pytest_lntail.log
cd /tmp
echo "== pmc fetch b1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r04_pmc_fetch_size_bench_b1.md $O/r04_pmc_traffic.json | tail -1
