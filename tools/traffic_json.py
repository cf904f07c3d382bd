#!/usr/bin/env python
"""HBM-side counter bytes per launch of one kernel from two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE need separate
passes: TCC slots), with the gfx950 correction of /opt/skills/guides/MI355X_MICROARCH.md (FETCH_SIZE reports half of the
bytes of wide coalesced reads: doubled; WRITE_SIZE as reported; both in KB per dispatch, summed over the XCD instances).
usage: traffic_json.py fetch.db write.db <kernel substring> <bench config> <source label>  > profiles/traffic_latest.json"""
import json
import sqlite3
import sys


def per_dispatch(db, counter, kernel):
    cur = sqlite3.connect(db).cursor()
    q = """select count(distinct d.id), sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           where p.name = ? and s.kernel_name like ?"""
    n, tot = cur.execute(q, (counter, f"%{kernel}%")).fetchone()
    return n, (tot or 0.0) / max(n, 1)


fetch_db, write_db, kernel, config, source = sys.argv[1:6]
nf, f_kb = per_dispatch(fetch_db, "FETCH_SIZE", kernel)
nw, w_kb = per_dispatch(write_db, "WRITE_SIZE", kernel)
print(json.dumps({"kernel": kernel, "config": config, "launches_fetch_pass": nf, "launches_write_pass": nw,
                  "fetch_size_kb_per_launch_raw": f_kb, "write_size_kb_per_launch": w_kb,
                  "bytes_per_launch": (2.0 * f_kb + w_kb) * 1024.0,
                  "correction": "FETCH_SIZE x 2 (gfx950: 128-B requests tallied at 64 B), WRITE_SIZE as reported (uncalibrated)",
                  "source": source}, indent=1))
