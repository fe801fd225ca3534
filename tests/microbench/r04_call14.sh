#!/bin/bash
# round 4, call 14: k_rows_gemm blocks carrying two feature groups that read the same token fragments (second read from the L1) — libwm_share.so
# (build.py --variant share -DWM_ROWS_SHARE) against the product library: layer GEMM times, bench at 32 streams, parity on the variant; then the HBM traffic counters
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r04c14; mkdir -p $O
: > $O/gemm_time.txt
timeout 120 python tests/microbench/r04_gemm_time.py base 2>&1 | grep "^\[" | tee -a $O/gemm_time.txt
WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_share.so timeout 120 python tests/microbench/r04_gemm_time.py share 2>&1 | grep "^\[" | tee -a $O/gemm_time.txt
for arm in share; do
  if [ $arm = share ]; then export WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_share.so; else unset WM_LIB; fi
  timeout 200 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$arm.json 2> $O/b32_$arm.err; echo $arm rc $?
done
unset WM_LIB
python - <<PY
import json
for arm in ("base", "share"):
    try:
        d = json.loads(open("$O/b32_%s.json" % arm).read().strip().splitlines()[-1]); r = d["roofline"]
        print(arm, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "decode tok/s", d["decode_tokens_per_sec_per_gpu"], "ratio", d["vanilla_anchor"]["medusa_over_vanilla"])
    except Exception as e: print(arm, "failed", e)
PY
WM_LIB=$R/whisper-medusa_amd/whisper_medusa/libwm_share.so timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -k "merged_step or batch_invariance or bit_exact" > $O/pytest_share.log 2>&1; echo pytest share rc $?; tail -2 $O/pytest_share.log
cd /tmp
echo "== pmc fetch b1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r04_pmc_fetch_size_bench_b1.md $O/r04_pmc_traffic.json | tail -1
