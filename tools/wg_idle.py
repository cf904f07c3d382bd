#!/usr/bin/env python
"""CU-idle time inside the fused-convolution launches, from the workgroup stamps of a profiling build.

A `-DDDMI_PROFILING=2` build of libddmi (tools/build_variant.sh) stamps the constant-rate clock (s_memrealtime, 100 MHz) at the
start and the end of every live k_conv_fused workgroup and, with DDMI_WG_DUMP=<file>, appends the records at every report
(ddmi_set_kernel_timing).  A fused workgroup owns its CU (125-158 KB of LDS), so the number of workgroups in flight is the number
of busy CUs.  This script sweeps the intervals and prints, over the time during which AT LEAST ONE fused workgroup runs:

  busy window          total time with >= 1 workgroup in flight
  CU-idle fraction     integral of (CUS - active) / (CUS x busy window)   -- launch ramps, tails, holes between the two streams
  packed bound         sum of workgroup durations / CUS                   -- what a perfectly packed walk of the same workgroups takes

Usage: python tools/wg_idle.py <dump> [--cus 256] [--forwards N]
"""
import argparse
import sys

import numpy as np


def analyse(path, cus=256, forwards=None, out=sys.stdout, records=None):
    raw = np.fromfile(path, dtype=np.uint64)
    rec = raw.reshape(-1, 2)
    if records:            # the first report of a run only (bench.py: warm-up + timed steps; later reports hold the HIP-event pass)
        rec = rec[:records]
    t0 = rec[:, 0].astype(np.int64) & ((1 << 52) - 1)
    t1 = (rec[:, 1] & np.uint64((1 << 52) - 1)).astype(np.int64)
    slot = ((rec[:, 1] >> np.uint64(56)) & np.uint64(15)).astype(np.int64)
    ok = t1 >= t0
    t0, t1, slot = t0[ok], t1[ok], slot[ok]
    n = len(t0)
    dur = (t1 - t0).astype(np.float64) * 1e-2          # us (100 MHz)
    ev_t = np.concatenate([t0, t1])
    ev_d = np.concatenate([np.ones(n, np.int64), -np.ones(n, np.int64)])
    order = np.lexsort((-ev_d, ev_t))                  # starts before ends at equal stamps
    ev_t, ev_d = ev_t[order], ev_d[order]
    active = np.cumsum(ev_d)
    dt = np.diff(ev_t).astype(np.float64) * 1e-2       # us; active[i] holds on [ev_t[i], ev_t[i+1])
    a = active[:-1]
    busy = a > 0
    busy_us = float(dt[busy].sum())
    over = int(a.max())
    capped = np.minimum(a, cus)
    idle_us_cu = float(((cus - capped) * dt)[busy].sum())
    work_us_cu = float(dur.sum())
    span_us = float((ev_t[-1] - ev_t[0]) * 1e-2)
    print(f"# {path}: {n} workgroups, span {span_us / 1e3:.2f} ms, max in flight {over} (CUS = {cus})", file=out)
    f = forwards or 1
    unit = "per forward" if forwards else "total"
    print(f"busy window (>= 1 fused workgroup)  : {busy_us / 1e3 / f:9.3f} ms {unit}", file=out)
    print(f"no fused workgroup in flight        : {(span_us - busy_us) / 1e3 / f:9.3f} ms {unit}", file=out)
    print(f"sum of workgroup durations / CUS    : {work_us_cu / cus / 1e3 / f:9.3f} ms {unit}   (perfectly packed walk of the same workgroups)", file=out)
    print(f"CU-idle inside the busy window      : {idle_us_cu / cus / 1e3 / f:9.3f} ms {unit} = {idle_us_cu / (cus * busy_us):.3f} of it", file=out)
    # where the idle CU-time sits: by number of workgroups in flight
    edges = [1, cus // 8, cus // 4, cus // 2, 3 * cus // 4, cus * 15 // 16, cus, 10 ** 9]
    print("busy-window time and idle CU-time by workgroups in flight:", file=out)
    for lo, hi in zip(edges[:-1], edges[1:]):
        m = busy & (a >= lo) & (a < hi)
        if not m.any():
            continue
        t = float(dt[m].sum())
        idle = float(((cus - capped) * dt)[m].sum())
        label = f"[{lo}, {min(hi, over + 1)})"
        print(f"    {label:14s} {t / 1e3 / f:8.3f} ms   idle {idle / cus / 1e3 / f:8.3f} ms-chip", file=out)
    print("per edge-group slot: workgroups, mean / p95 / max duration (us), share of the workgroup time", file=out)
    for s in np.unique(slot):
        d = dur[slot == s]
        print(f"    slot {s}: {len(d):8d}  {d.mean():8.1f} {np.percentile(d, 95):8.1f} {d.max():8.1f}   {d.sum() / work_us_cu:.3f}", file=out)
    return {"busy_ms": busy_us / 1e3, "idle_frac": idle_us_cu / (cus * busy_us), "packed_ms": work_us_cu / cus / 1e3}


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("dump")
    ap.add_argument("--cus", type=int, default=256)
    ap.add_argument("--forwards", type=int, default=None, help="forwards behind the dump: print per-forward figures")
    ap.add_argument("--records", type=int, default=None, help="use the first N records (the first report appended to the dump)")
    args = ap.parse_args()
    analyse(args.dump, args.cus, args.forwards, records=args.records)
