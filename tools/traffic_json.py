#!/usr/bin/env python
"""Counter traffic of one kernel per launch from separate FETCH_SIZE / WRITE_SIZE (/ TCC_HIT + TCC_MISS) rocprofv3 --pmc passes (rocpd DBs).
usage: traffic_json.py fetch.db write.db <kernel-substring> <config> <source-note> [l2.db] > traffic.json
bytes = FETCH_SIZE[KB] * 1024 * 2 (gfx950: FETCH_SIZE counts half of a wide coalesced stream, MI355X_MICROARCH.md HBM section)
      + WRITE_SIZE[KB] * 1024, launch-weighted mean over all dispatches of the kernel; l2_hit_rate = TCC_HIT / (TCC_HIT + TCC_MISS)."""
import hashlib
import json
import os
import sqlite3
import sys


def counter(db, name, sub):
    cur = sqlite3.connect(db).cursor()
    q = """select count(distinct d.id), sum(e.value) from rocpd_pmc_event e join rocpd_info_pmc p on e.pmc_id = p.id
           join rocpd_kernel_dispatch d on e.event_id = d.event_id join rocpd_info_kernel_symbol s on d.kernel_id = s.id
           where p.name = ? and s.kernel_name like ?"""
    n, tot = cur.execute(q, (name, f"%{sub}%")).fetchone()
    return n, (tot or 0.0) / max(n, 1)


fetch_db, write_db, sub, config, note = sys.argv[1:6]
n, fkb = counter(fetch_db, "FETCH_SIZE", sub)
_, wkb = counter(write_db, "WRITE_SIZE", sub)
# (the timer row of bench.py is "k_conv_fused" for both dispatches: k_conv_fused per (layer, group) and k_conv_grouped per layer --
# pass the substring "k_conv_" to count whichever the run launched)
out = {"kernel": "k_conv_fused" if sub.startswith("k_conv") else sub, "kernel_match": sub, "config": config, "launches": n, "fetch_size_kb_per_launch": fkb, "write_size_kb_per_launch": wkb,
       "bytes_per_launch": fkb * 1024 * 2 + wkb * 1024,
       "source": note,
       "corrections": "FETCH_SIZE doubled (gfx950 counts half of a wide coalesced read); WRITE_SIZE uncalibrated; counters come "
                      "from the L2's fabric side, so Infinity-Cache hits are included"}
# the kernel source the counters were collected on: bench.py refuses to replay the record next to a different k_conv_tile.h
src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "diffdock_amd", "csrc", "k_conv_tile.h")
out["kernel_source"] = "diffdock_amd/csrc/k_conv_tile.h"
out["kernel_source_sha256"] = hashlib.sha256(open(src, "rb").read()).hexdigest()
# the scatter stage (k_reduce_bn) from the same passes: bench.py's roofline_scatter.traffic
ns, sfkb = counter(fetch_db, "FETCH_SIZE", "k_reduce_bn")
_, swkb = counter(write_db, "WRITE_SIZE", "k_reduce_bn")
if ns:
    out["scatter"] = {"kernel": "k_reduce_bn", "launches": ns, "fetch_size_kb_per_launch": sfkb, "write_size_kb_per_launch": swkb,
                      "bytes_per_launch": sfkb * 1024 * 2 + swkb * 1024}
if len(sys.argv) > 6:
    _, hit = counter(sys.argv[6], "TCC_HIT_sum", sub)
    _, miss = counter(sys.argv[6], "TCC_MISS_sum", sub)
    out["l2_hit_rate"] = hit / max(hit + miss, 1.0)
    out["tcc_hit_per_launch"], out["tcc_miss_per_launch"] = hit, miss
print(json.dumps(out, indent=1))
