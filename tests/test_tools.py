"""Measurement tools that the round's numbers rest on (tools/): checked on synthetic inputs, no GPU."""
import os
import sqlite3
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _trace(path, forwards, with_events):
    """A rocpd-like kernel trace: per forward  k_perturb | k_edge_hidden_mm | gap | k_conv_fused (+ k_reduce_bn behind a gap).
    with_events: the trailing `with_events` forwards carry a 10-us hole in front of every kernel (bench.py's HIP-event pass)."""
    db = sqlite3.connect(path)
    db.execute("create table rocpd_info_kernel_symbol (id integer, kernel_name text)")
    db.execute("create table rocpd_kernel_dispatch (start integer, end integer, kernel_id integer)")
    names = ["_ZN4ddmi9k_perturbEv", "_ZN4ddmi16k_edge_hidden_mmILi3EEEvNS_14EdgeHiddenArgsE", "_ZN4ddmi12k_conv_fusedILi3ELi4ELi3ELi5ELb0EEEvNS_13FusedConvArgsE",
             "_ZN4ddmi11k_reduce_bnEv"]
    for i, n in enumerate(names):
        db.execute("insert into rocpd_info_kernel_symbol values (?, ?)", (i, n))
    t = 1000
    for f in range(forwards):
        hole = 10_000 if f >= forwards - with_events else 0
        for kid, dur, gap in ((0, 5_000, 0), (1, 100_000, 2_000), (2, 800_000, 3_000), (3, 60_000, 7_000)):
            t += gap + hole
            db.execute("insert into rocpd_kernel_dispatch values (?, ?, ?)", (t, t + dur, kid))
            t += dur
    db.commit(); db.close()


def _run(db, *args):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "timeline.py"), db, "k_perturb", *map(str, args)], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return r.stdout


def test_timeline_attributes_idle_gaps_and_skips_the_event_pass(tmp_path):
    """tools/timeline.py: covered / exposed / idle per forward, the idle time by (kernel before -> kernel after), and the
    skip_last argument that keeps bench.py's trailing HIP-event pass (holes in front of every kernel) out of the window."""
    db = str(tmp_path / "kt.db")
    _trace(db, forwards=40, with_events=20)
    timed = _run(db, 10, 22)
    evpass = _run(db, 10)

    def field(out, key):
        line = next(l for l in out.splitlines() if key in l)
        return float(line.split()[-1] if key != "covered" else line.split(":")[1].split()[0])
    # timed region: per forward 0.8 ms covered, gaps 2 + 3 + 7 us = 0.012 ms idle
    assert abs(field(timed, "covered") - 0.800) < 1e-3
    assert abs(field(timed, "(idle)") - 0.012) < 1e-3
    # event pass: + 4 x 10 us of holes per forward
    assert abs(field(evpass, "(idle)") - 0.052) < 1e-3
    assert "22 trailing forwards skipped" in timed
    gaps = timed[timed.index("idle time by"):]
    row = next(l for l in gaps.splitlines() if "k_conv_fused" in l and "-> k_reduce_bn" in l)
    assert abs(float(row.split()[-1]) - 7.0) < 0.1            # mean us of that gap
    assert abs(float(row.split()[-2]) - 1.0) < 1e-6           # one such gap per forward
