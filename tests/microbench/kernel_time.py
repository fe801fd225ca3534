"""Average duration of kernels matching a substring from a rocprofv3 rocpd db, only launches > min_us."""
import sqlite3, sys
db, pat, min_us = sys.argv[1], sys.argv[2], float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
c = sqlite3.connect(db)
r = c.execute("select count(*), avg(end-start)/1e3 from kernels where name like ? and (end-start)/1e3 > ?", (f"%{pat}%", min_us)).fetchone()
print(pat, "calls", r[0], "avg us", r[1])
