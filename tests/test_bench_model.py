"""bench.py's work model (algorithmic flops / bytes behind the `roofline` object) -- host logic, no GPU."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load_bench():
    spec = importlib.util.spec_from_file_location("bench_module", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules["bench_module"] = mod
    spec.loader.exec_module(mod)
    return mod


def test_conv_work_counts_every_group_launch_once():
    b = load_bench()
    cfg = b.bench_cfg()
    nL, nR, e_ll, e_lr, e_rr = 40 * 30, 40 * 300, 21456, 325563, 288000
    fused = b.conv_work(cfg, nL, nR, e_ll, e_lr, e_rr)
    # 4 groups in layers 0..4, the two ligand-target groups in the last layer (cg_model.py:331-347)
    assert len(fused) == 4 * (cfg.num_conv_layers - 1) + 2 and all(set(w) == {"k_conv_fused"} for w in fused)
    # full-width layer (156 -> 156): 2 * 145 * 524 flop per edge + 2 * 145 * 9648 per gather node
    rr = [w["k_conv_fused"] for w in fused][4 * 3 + 2]          # layer 3, group rec-rec
    assert abs(rr["flops"] - (2 * 145 * 524 * e_rr + 2 * 145 * 9648 * nR)) < 1e-6 * rr["flops"]
    assert abs(rr["bytes"] - (nR * 156 * 4 + e_rr * (144 + 156) * 4)) < 1.0


def test_reference_association_flops_equal_the_survey_figure():
    """2*(3ns*3ns + 3ns*W) + 6W flop per edge in the REFERENCE's association = 8.3 TFLOP per forward at
    configs[2] (1.02 M edges per layer) -- reported next to the re-associated count, never used as the numerator of frac."""
    b = load_bench()
    w = b.conv_work(b.bench_cfg(), 1200, 12000, 40 * 30 * 14, 360000, 288000)
    ref = sum(x["k_conv_fused"]["ref_flops"] for x in w)
    mine = sum(x["k_conv_fused"]["flops"] for x in w)
    assert abs(ref / 1e12 - 8.3) < 0.05 and 8.5 < ref / mine < 10.0


def test_workloads_cover_the_baseline_configs():
    b = load_bench()
    assert b.WORKLOADS["configs2"]["complexes"] == [(300, 30, 0)] and b.WORKLOADS["configs2"]["samples"] == 40
    assert b.WORKLOADS["configs1"]["samples"] == 10 and b.WORKLOADS["configs4"]["complexes"][0][:2] == (1500, 80)
    mix = b.WORKLOADS["mix"]["complexes"]
    assert sorted({c[0] for c in mix}) == [150, 300, 500] and sorted({c[1] for c in mix}) == [20, 30, 45] and len(mix) == 9


def test_committed_traffic_record_is_what_bench_reads():
    """profiles/traffic_latest.json (written by tools/round_profile.sh from the FETCH_SIZE / WRITE_SIZE passes) carries the keys
    bench.py copies into `roofline.traffic`, for the default workload and the dominant kernel."""
    import json
    b = load_bench()
    rec = json.load(open(b.TRAFFIC_RECORD))
    assert rec["kernel"] == "k_conv_fused" and rec["config"] == "configs2"
    assert rec["bytes_per_launch"] == (2.0 * rec["fetch_size_kb_per_launch"] + rec["write_size_kb_per_launch"]) * 1024.0
    assert 1e8 < rec["bytes_per_launch"] < 1e10 and rec["source"].startswith("profiles/")


def test_traffic_record_of_another_kernel_source_is_not_replayed(tmp_path):
    """roofline.traffic is replayed from a committed counter record: only when the record names the k_conv.hip in the tree."""
    import hashlib
    b = load_bench()
    src = tmp_path / "k.hip"
    src.write_text("kernel v1")
    rec = {"bytes_per_launch": 1.0e9, "source": "profiles/x.txt", "l2_hit_rate": 0.5,
           "kernel_source_sha256": hashlib.sha256(b"kernel v1").hexdigest()}
    assert b.traffic_from_record(rec, str(src)) == (1.0e9, "profiles/x.txt", 0.5)
    src.write_text("kernel v2")
    t, note, hit = b.traffic_from_record(rec, str(src))
    assert t is None and hit is None and note.startswith("stale")
    assert b.traffic_from_record({k: v for k, v in rec.items() if k != "kernel_source_sha256"}, str(src))[0] is None


def test_roofline_constants_match_the_microarchitecture_guide():
    b = load_bench()
    assert b.MFMA_F32_PEAK_TFLOPS == 157.3 and b.HBM_PEAK_GBS == 8000.0
