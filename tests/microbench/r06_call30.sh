#!/bin/bash
# round 6, GPU call 30: candidate stage in two launches (k_select1 with head 1's slice winners + k_cand_fin) against the previous four (libwm_prev.so = the commit before):
# one-stream bench and forced-accept cost interleaved twice, 32-stream bench once per arm; then the whole GPU suite
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r06c30; mkdir -p $O
P=$GRAFT_REPO_ROOT/whisper-medusa_amd/whisper_medusa
for rep in 1 2; do
  for lib in f16 prev; do
    WM_LIB_F16=$P/libwm_$lib.so timeout 300 python tests/microbench/r06_sib_cost.py 2>&1 | grep "^WM_SIBLINGS" | sed "s/^/[$lib] /" | tee -a $O/sib_cost.log
    WM_LIB_F16=$P/libwm_$lib.so timeout 400 python bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b1_${lib}_$rep.json 2> $O/b1_${lib}_$rep.err
    python - <<PY
import json
d = json.loads(open("$O/b1_${lib}_$rep.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("[$lib] b1", d["value"], "tok/s", r["ms_per_launch"], "ms/iter frac", r["frac"], "hits", d.get("sibling_hits"), "hist", d["accept_hist"])
PY
  done
done
for lib in f16 prev; do
  WM_LIB_F16=$P/libwm_$lib.so timeout 400 python bench.py --batch 32 --steps 8 --warmup 2 --no-cpu-baseline --no-extra-configs > $O/b32_$lib.json 2> $O/b32_$lib.err
  python - <<PY
import json
d = json.loads(open("$O/b32_$lib.json").read().strip().splitlines()[-1])
r = d["roofline"]
print("[$lib] b32", d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "hist", d["accept_hist"])
PY
done
( time timeout 1500 python -m pytest tests -m gpu -x -q -p no:cacheprovider > $O/pytest_gpu.log 2>&1 ) 2>&1 | grep real; echo pytest rc $?
grep -h "^FAILED\|^ERROR\|passed\|failed" $O/pytest_gpu.log | cut -c1-300 | tail -6
grep -h "parity ties" $O/pytest_gpu.log | cut -c1-160
