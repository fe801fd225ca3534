"""Per-kernel sums of the SQ counters of a `rocprofv3 --pmc ... --kernel-trace` run (one pass, <= 8 SQ counters), as fractions
of SQ_WAVE_CYCLES where that is meaningful (WAIT_ANY + WAIT_INST_ANY + ACTIVE_INST_ANY ~ WAVE_CYCLES, MI355X_MICROARCH.md).
    python tests/sq_summary.py <db> [out.md] [kernel-regex]"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
pat = re.compile(sys.argv[3] if len(sys.argv) > 3 else r"k_gemm|k_flash|k_rows_gemm|k_rows_lds|k_skinny|k_attn|k_ln_tiles")
rows = c.execute("select name, counter_name, sum(counter_value), count(distinct dispatch_id) from pmc_events group by name, counter_name").fetchall()
agg = {}
for n, cn, v, k in rows:
    n = re.sub(r"\(.*", "", n)[:70]
    if pat.search(n):
        agg.setdefault(n, {})[cn] = v
        agg[n]["_n"] = k
ctrs = sorted({k for d in agg.values() for k in d if k != "_n"})
lines = ["| kernel | dispatches | " + " | ".join(ctrs) + " |", "|---|---|" + "---|" * len(ctrs)]
for n, d in sorted(agg.items(), key=lambda kv: -kv[1].get("SQ_WAVE_CYCLES", 0)):
    wc = d.get("SQ_WAVE_CYCLES", 0) or 1
    cells = []
    for cn in ctrs:
        v = d.get(cn, 0)
        cells.append(f"{v:.4g}" + (f" ({v / wc:.2f})" if cn.startswith("SQ_WAIT") or cn.startswith("SQ_ACTIVE") else ""))
    lines.append(f"| `{n}` | {d['_n']} | " + " | ".join(cells) + " |")
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
