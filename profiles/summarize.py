"""Turn a rocprofv3 (ROCm 7.2, rocpd sqlite) kernel trace into the per-kernel stats table that
`--stats` prints: python profiles/summarize.py <results.db> [include-regex [exclude-regex]] [> profiles/<name>.txt]
(the regexes select kernels by name: e.g. k_dec_ for the decode table, '' k_dec_ for everything else)"""
import re
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
rows = c.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, "
                 "max(end-start)/1e3 from kernels group by name order by 3 desc").fetchall()
inc = re.compile(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] else None
exc = re.compile(sys.argv[3]) if len(sys.argv) > 3 and sys.argv[3] else None
rows = [r for r in rows if (inc is None or inc.search(r[0])) and (exc is None or not exc.search(r[0]))]
tot = sum(r[2] for r in rows)
print(f"# rocprofv3 --kernel-trace --stats summary of {sys.argv[1]}" + (f"  [kernels matching {sys.argv[2]!r}" + (f", not {sys.argv[3]!r}" if exc else "") + "]" if inc or exc else ""))
print(f"# total kernel time {tot / 1e3:.3f} ms over {sum(r[1] for r in rows)} dispatches")
print(f"{'pct':>7} {'calls':>7} {'total_us':>12} {'avg_us':>10} {'min_us':>9} {'max_us':>9}  kernel")
for r in rows:
    print(f"{r[2] / tot * 100:7.2f} {r[1]:7d} {r[2]:12.1f} {r[3]:10.2f} {r[4]:9.2f} {r[5]:9.2f}  {r[0][:110]}")
