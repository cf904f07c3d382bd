#!/usr/bin/env python
"""Per-kernel summary (calls, total/avg/min/max duration) from a rocprofv3 `--kernel-trace --stats` rocpd SQLite
database, written as the text table kept under profiles/.   usage: rocprof_summary.py results.db > summary.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
tot = cur.execute("select sum(end-start)/1e6 from rocpd_kernel_dispatch").fetchone()[0]
print(f"# source: {sys.argv[1]}   total kernel time {tot:.2f} ms")
print(f"{'kernel':72s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'min_us':>9s} {'max_us':>10s} {'pct':>6s}")
q = """select s.kernel_name, count(*), sum(d.end-d.start)/1e6, avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3,
       max(d.end-d.start)/1e3 from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id=s.id
       group by s.kernel_name order by 3 desc"""
for r in cur.execute(q):
    print(f"{r[0][:72]:72s} {r[1]:7d} {r[2]:10.2f} {r[3]:10.1f} {r[4]:9.1f} {r[5]:10.1f} {100 * r[2] / tot:6.1f}")
