"""Summarise a rocprofv3 rocpd database (kernel-trace) per kernel: python tests/prof_summary.py <db> [out.md]"""
import re
import sqlite3
import sys

db = sys.argv[1]
c = sqlite3.connect(db)
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                 "from kernels group by name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows)
lines = [f"total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches", "",
         "| % | calls | avg us | min us | max us | kernel |", "|---|---|---|---|---|---|"]
for r in rows[:45]:
    name = re.sub(r"\(.*", "", r[0])[:100]
    lines.append(f"| {r[2] / tot * 100:.1f} | {r[1]} | {r[3]:.2f} | {r[4]:.2f} | {r[5]:.2f} | `{name}` |")
# reconciliation with bench.py's hipEvent timing: decode-path kernel time per Medusa iteration (run bench.py --no-vanilla)
dec = c.execute("select sum(end-start)/1e3, count(*) from kernels where name like '%k_skinny%' or name like '%k_attn%' or name like '%k_rows_gemm%' "
                "or name like '%k_ln_tiles%' or name like '%k_tile_gemm%' or name like '%k_select%' or name like '%k_embed%' or name like '%k_rows_norm%' "
                "or name like '%k_accept%' or name like '%k_set_cand%'").fetchone()
n_iter = c.execute("select count(*) from kernels where name like '%k_accept%' and name not like '%vanilla%'").fetchone()[0]
n_van = c.execute("select count(*) from kernels where name like '%k_accept_vanilla%'").fetchone()[0]
if n_iter and not n_van:
    lines += ["", f"decode path: {dec[1]} kernel launches, {dec[0] / 1e3:.3f} ms of kernel time over {n_iter} Medusa iterations = "
                  f"{dec[0] / n_iter:.1f} us busy and {dec[1] / n_iter:.0f} launches per iteration "
                  "(bench.py's ms_per_launch additionally contains the launch boundaries between them)"]
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
