#!/bin/bash
# round 6, GPU call 4: the fp16 single-plane decode contract (libwm_f16.so, wm_config.act_fp16) next to the hi / lo contract: the two-contract
# parity file, the large-v2 logit tests, per-GEMM times, one-stream and 32-stream bench lines
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c4; mkdir -p $O
( time timeout 1200 python -m pytest tests/test_gpu_act.py -m gpu -x -q -p no:cacheprovider -s > $O/pytest_act.log 2>&1 ) 2>&1 | grep real; echo pytest act rc $?
grep -h "max rel\|agree with\|passed\|failed\|Error\|assert" $O/pytest_act.log | cut -c1-300 | tail -25
( time timeout 900 python -m pytest tests/test_gpu_large.py -m gpu -x -q -p no:cacheprovider -k "prompt_pass or block_decode_loop" -s > $O/pytest_large.log 2>&1 ) 2>&1 | grep real
grep -h "rel to scale\|passed\|failed" $O/pytest_large.log | cut -c1-400
for act in f16 hilo; do
  WM_ACT=$act timeout 300 python tests/microbench/r06_gemm_time.py 2> $O/gt_$act.err | tee -a $O/gemm_time.log
done
for rep in 1 2; do
for act in f16 hilo; do
  timeout 300 python bench.py --act $act --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_$act$rep.json 2> $O/b1_$act$rep.err
  timeout 300 python bench.py --act $act --batch 32 --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32_$act$rep.json 2> $O/b32_$act$rep.err
  python - <<PY
import json
for tag in ("b1", "b32"):
    try:
        d = json.loads(open("$O/%s_$act$rep.json" % tag).read().strip().splitlines()[-1]); r = d["roofline"]; v = d["vanilla_anchor"]
        print("act=$act", tag, d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "vanilla", v["ms_per_token_step"], "ratio", v["medusa_over_vanilla"], "tok/iter", d["tokens_per_iter"], "steps/iter", r["passes_per_iteration"], flush=True)
    except Exception as e: print("act=$act", tag, "failed", e, open("$O/%s_$act$rep.err" % tag).read()[-600:])
PY
done; done 2>&1 | tee $O/bench.log
