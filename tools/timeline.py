#!/usr/bin/env python
"""Where the wall clock of a forward goes when no k_conv_fused launch is running: from a rocprofv3 --kernel-trace rocpd
database, over a window of whole forwards (delimited by the once-per-step kernel `marker`), the time covered by at least one
k_conv_fused dispatch, and for the rest ("exposed") the kernel that is running (earliest-started one) or idle.
usage: timeline.py results.db [marker=k_perturb] [n_forwards=10] [skip_last=0] > table.txt
skip_last: forwards at the END of the trace to leave out -- bench.py's last sampling run (20 forwards) is its HIP-event pass, with an
event pair around every kernel: its gaps are not those of the timed region (round 5: idle 0.40 ms per forward there at every batch size)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
marker = sys.argv[2] if len(sys.argv) > 2 else "k_perturb"
nfw = int(sys.argv[3]) if len(sys.argv) > 3 else 10
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0
rows = db.execute("""select d.start, d.end, s.kernel_name from rocpd_kernel_dispatch d
                     join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start""").fetchall()


import re


def short(n):   # rocpd keeps mangled names (_ZN4ddmi12k_conv_fusedILi3E...): the k_* identifier is enough here
    m = re.search(r"k_[a-z0-9_]+", n)
    return m.group(0) if m else n[:40]


marks = [r[0] for r in rows if marker in r[2]]
if len(marks) < nfw + 2 + skip:
    print(f"# only {len(marks)} marker dispatches ({marker}); using the whole trace")
    t0, t1, nf = rows[0][0], rows[-1][1], max(1, len(marks))
else:
    t0, t1, nf = marks[-nfw - 1 - skip], marks[-1 - skip], nfw
ev = [(a, b, short(n)) for a, b, n in rows if b > t0 and a < t1]
# sweep over interval boundaries
pts = sorted(set([t0, t1] + [min(max(x, t0), t1) for a, b, _ in ev for x in (a, b)]))
active = []
ev.sort()
i = 0
fused_cov = 0
exposed = defaultdict(int)
nover = defaultdict(int)
gaps = defaultdict(lambda: [0, 0])      # (kernel that ended last -> kernel that starts next): [idle ns, count]
last_ended = "(start)"
for p, q in zip(pts[:-1], pts[1:]):
    while i < len(ev) and ev[i][0] <= p:
        active.append(ev[i]); i += 1
    for e in active:
        if e[1] <= p:
            last_ended = e[2]
    active = [e for e in active if e[1] > p]
    if q <= p:
        continue
    names = [e[2] for e in active]
    if not names:
        nxt = ev[i][2] if i < len(ev) else "(end)"
        g = gaps[(last_ended, nxt)]
        g[0] += q - p; g[1] += 1
    if any(n.startswith("k_conv_fused") for n in names):
        fused_cov += q - p
        for n in set(names):
            if not n.startswith("k_conv_fused"):
                nover[n] += q - p
    elif names:
        exposed[min(active)[2]] += q - p
    else:
        exposed["(idle)"] += q - p
span = (t1 - t0) / 1e6 / nf
print(f"# {sys.argv[1]}: window of {nf} forwards ({skip} trailing forwards skipped), {span:.3f} ms per forward")
print(f"covered by >= 1 k_conv_fused dispatch : {fused_cov / 1e6 / nf:8.3f} ms per forward")
print(f"exposed (no k_conv_fused running)     : {sum(exposed.values()) / 1e6 / nf:8.3f} ms per forward")
for n, v in sorted(exposed.items(), key=lambda kv: -kv[1]):
    print(f"    {n:40s} {v / 1e6 / nf:8.3f}")
print("time other kernels ran next to a k_conv_fused dispatch (ms per forward):")
for n, v in sorted(nover.items(), key=lambda kv: -kv[1])[:12]:
    print(f"    {n:40s} {v / 1e6 / nf:8.3f}")
tot = defaultdict(int); cnt = defaultdict(int)
for a, b, n in ev:
    tot[n] += b - a; cnt[n] += 1
print("per-kernel totals in the window (ms per forward, launches per forward):")
for n, v in sorted(tot.items(), key=lambda kv: -kv[1])[:20]:
    print(f"    {n:40s} {v / 1e6 / nf:8.3f} {cnt[n] / nf:7.1f}")
print("idle time by (kernel that ended last -> kernel that starts next): ms per forward, gaps per forward, mean us")
for (a, b), (v, c) in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
    print(f"    {a:28s} -> {b:28s} {v / 1e6 / nf:8.3f} {c / nf:7.1f} {v / 1e3 / max(c, 1):7.1f}")
