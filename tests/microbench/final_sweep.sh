#!/bin/bash
# The batch / head-type / weight-format table of DESIGN.md §7: one bench.py JSON line per row into gpurun_out/sweep/.
mkdir -p gpurun_out/sweep
run() { # batch micro heads extra-flag tag
  timeout 400 python bench.py --batch $1 --micro-batches $2 --heads $3 $4 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 > gpurun_out/sweep/$5.json
  python - "$5" <<'PY'
import json, sys
d = json.load(open(f"gpurun_out/sweep/{sys.argv[1]}.json"))
v = d["vanilla_anchor"]
print(f"| {sys.argv[1]} | {d['value']:.0f} | {d['roofline']['ms_per_launch']:.2f} | {d['tokens_per_iter']:.2f} | {v['tokens_per_sec_per_gpu']:.0f} | {d['x_realtime']:.0f} | {d['roofline']['frac']:.3f} | {d['roofline']['prefill']['achieved']:.0f} |")
PY
}
echo "| run | tokens/s | ms/iteration | tokens/iteration | vanilla tokens/s | x real time | HBM roofline frac | prefill TFLOP/s |"
echo "|---|---|---|---|---|---|---|---|"
for b in 1 2 4 8 16 32; do run $b 1 linear "" linear_b$b; done
run 32 2 linear "" linear_b32_mb2
run 1 1 block "" block_b1; run 32 1 block "" block_b32; run 32 2 block "" block_b32_mb2
run 1 1 linear --fp8-weights fp8_b1; run 32 2 linear --fp8-weights fp8_b32_mb2
