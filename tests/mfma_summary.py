"""MFMA utilisation per kernel from a `rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES --kernel-trace` run.
SQ_VALU_MFMA_BUSY_CYCLES summed over all counter instances of a dispatch = 16 cycles x number of
v_mfma_f32_16x16x32_bf16 wave-instructions (calibrated on the FC1 GEMM: M/16 x N/16 x K/32 MFMAs), so
utilisation = busy cycles / (kernel duration x 2.4 GHz x 1024 SIMDs).  The fp8 kernels' v_mfma_f32_16x16x32_fp8_fp8 occupies the
pipe for the same 16 cycles (non-scaled fp8 runs at the bf16 rate on gfx950).   python tests/mfma_summary.py <db> [out.md]"""
import re
import sqlite3
import sys

CLK, SIMDS = 2.4e9, 1024
c = sqlite3.connect(sys.argv[1])
cols = [r[1] for r in c.execute("pragma table_info(pmc_events)").fetchall()]
did = next((x for x in ("dispatch_id", "event_id", "kernel_id", "id") if x in cols), None)
print("pmc_events columns:", cols, "-> dispatch key:", did, file=sys.stderr)
q = (f"select name, {did}, sum(counter_value), max(end-start) from pmc_events where counter_name='SQ_VALU_MFMA_BUSY_CYCLES' "
     f"group by name, {did}")
agg = {}
for n, d, v, t in c.execute(q):
    a = agg.setdefault(re.sub(r"\(.*", "", n)[:80], [0, 0.0, 0.0])
    a[0] += 1; a[1] += v; a[2] += t * 1e-9
lines = ["| kernel | dispatches | avg us | MFMA wave-instructions / dispatch | MFMA utilisation |", "|---|---|---|---|---|"]
for n, (k, busy, sec) in sorted(agg.items(), key=lambda kv: -kv[1][2]):
    if not re.search(r"k_gemm_tiled|k_gemm_256|k_gemm_f8|k_flash_enc|k_rows_gemm|k_skinny|k_attn", n) or sec <= 0:
        continue
    lines.append(f"| `{n}` | {k} | {sec / k * 1e6:.1f} | {busy / 16 / k:.4g} | {busy / (sec * CLK * SIMDS):.3f} |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
