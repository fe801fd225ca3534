#!/bin/bash
# round 4, GPU call 5: LayerNorm-fused launches without LDS-DMA / exec-masked loads (request batch in one go, counted waits): parity + single-stream and 32-stream iteration
cd "$GRAFT_REPO_ROOT" || exit 1
O=$GRAFT_REPO_ROOT/gpurun_out/r04c5; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -p no:cacheprovider -x -k "not fp8 and not audio and not wav and not pinned" > $O/pytest.log 2>&1; echo pytest rc $?; tail -3 $O/pytest.log
timeout 120 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b1.json 2> $O/b1.err; echo b1 rc $?
timeout 150 python bench.py --batch 32 --steps 2 --warmup 1 --no-cpu-baseline --no-extra-configs > $O/b32.json 2> $O/b32.err; echo b32 rc $?
python - <<PY
import json
for b in ("b1", "b32"):
    try:
        d = json.loads(open("$O/%s.json" % b).read().strip().splitlines()[-1]); r = d["roofline"]
        print(b, d["value"], "tok/s", r["ms_per_launch"], "ms/iter", "frac", r["frac"], "frac_executed", r.get("frac_executed"), "vanilla", d["vanilla_anchor"], "layer gemms", r.get("layer_gemms"))
    except Exception as e: print(b, "failed", e)
PY
