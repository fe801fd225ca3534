#!/bin/bash
# round 3, GPU call 15: residual epilogue of the few-clip kernels (one load batch, stores at the end); encoder-only wave epilogues moved
# out of the decode sources' hash -> HBM traffic counters re-taken on the final sources
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r03c15; mkdir -p $O
echo "== pytest (encoder: tiny shapes end to end, large-v2 big batch against one clip, prompt pass)"
timeout 300 python -m pytest tests/test_gpu_parity.py tests/test_gpu_large.py -m gpu -q -p no:cacheprovider -x -k "encoder_output or big_batch or prompt_pass or test_generate_api_end_to_end" > $O/pytest.log 2>&1; echo rc $?; tail -3 $O/pytest.log
cd /tmp
echo "== pmc fetch b1"
timeout 200 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pmc1 -o pmc1 -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-vanilla --no-extra-configs > $O/pmc1.log 2>&1; echo rc $?
DB=$(find /tmp/pmc1 -name "*.db" | head -1); python $R/tests/pmc_summary.py $DB $O/r03_pmc_fetch_size_bench_b1.md $O/r03_pmc_traffic.json | tail -1
cp $O/r03_pmc_traffic.json $R/profiles/r03_pmc_traffic.json
cd $R
echo "== bench b1"
timeout 200 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1.json 2> $O/b1.err; echo rc $?
python - <<PY
import json
d = json.loads(open("$O/b1.json").read().strip().splitlines()[-1]); r = d["roofline"]
print("b1", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "traffic", r.get("traffic"), "prefill", r["prefill"]["achieved"], "enc ms", d.get("ms_encode_per_step"))
PY
