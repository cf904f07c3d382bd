#!/usr/bin/env python
"""Per-(kernel, grid) averages of the PMC counters in rocprofv3 rocpd databases (one --pmc pass per DB).
usage: pmc_summary.py a.db [b.db ...] > summary.txt      Counter values are summed over the XCD instances of a dispatch."""
import sqlite3
import sys

for name in sys.argv[1:]:
    cur = sqlite3.connect(name).cursor()
    print("#", name)
    q = """select s.kernel_name, d.grid_size_x, d.workgroup_size_x, p.name, count(distinct d.id), sum(e.value) * 1.0 / count(distinct d.id),
                  avg(d.end - d.start) / 1e3
           from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id
           join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           group by s.kernel_name, d.grid_size_x, p.name order by 7 desc, 1, 2, 4"""
    for r in cur.execute(q):
        if r[6] < 40:
            continue
        print(f"{r[0][:44]:44s} grid={r[1]:8d} wg={r[2]:4d} {r[3]:28s} n={r[4]:4d} per_dispatch={r[5]:.5g} avg_us={r[6]:.1f}")
