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
out = "\n".join(lines)
print(out)
if len(sys.argv) > 2:
    open(sys.argv[2], "w").write(out + "\n")
