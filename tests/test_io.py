"""diffdock_amd.io: PDB / SDF readers and graph construction on the reference's example complex."""
import os
import subprocess

import numpy as np
import pytest
import torch

import cases
from diffdock_amd.config import TINY
from diffdock_amd.io import complex_graph, ligand_bond_arrays, read_sdf, receptor_graph, transformation_mask
from diffdock_amd.model import MIScoreModel
from diffdock_amd.synth import make_complex, receptor_contact_graph
from util import load_fixture, tables

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EMU = os.path.join(ROOT, "tests", "hipemu", "libddmi_emu.so")
REF_DATA = "/root/reference/data/1a0q"


def test_fixture_has_the_documented_sizes():
    d = load_fixture("1a0q_graph")
    assert d["rec_pos"].shape == (416, 3) and d["lig_pos"].shape == (23, 3)        # 416 residues, 23 heavy atoms
    assert d["bond_index"].shape[1] == 46 and d["bond_attr"].sum(1).eq(1).all()
    assert abs(float(d["rec_pos"].mean())) < 1e-4                                   # centred on the C-alpha centroid
    deg = torch.bincount(d["rec_edge_index"][1], minlength=416)
    assert int(deg.max()) == 24 and int(deg.min()) >= 1                             # 24 nearest within 15 A
    src, dst = d["rec_edge_index"]
    assert float((d["rec_pos"][src] - d["rec_pos"][dst]).norm(dim=1).max()) < 15.0
    assert int(d["edge_mask"].sum()) == d["mask_rotate"].shape[0]


@pytest.mark.skipif(not os.path.isdir(REF_DATA), reason="the reference's data directory exists only in the build container")
def test_io_reproduces_the_reference_executed_fixture_bit_for_bit():
    """diffdock_amd.io against the fixture whose receptor graph / features / positions and rotatable-bond masks were produced
    by EXECUTING the reference (datasets/process_mols.py new_extract_receptor_structure, utils/torsion.py
    get_transformation_mask, the centring of datasets/pdbbind.py; tests/golden/make_1a0q.py)."""
    g = complex_graph(f"{REF_DATA}/1a0q_protein_processed.pdb", f"{REF_DATA}/1a0q_ligand.sdf", lm_dim=0)
    d = load_fixture("1a0q_graph")
    assert "process_mols.py new_extract_receptor_structure executed" in d["provenance"]
    assert torch.equal(g["receptor"].pos, d["rec_pos"]) and torch.equal(g["receptor", "receptor"].edge_index, d["rec_edge_index"])
    assert torch.equal(g["receptor"].x, d["rec_x"]) and torch.equal(g.original_center, d["original_center"])
    assert torch.equal(g["ligand"].pos, d["lig_pos"]) and torch.equal(g["ligand", "ligand"].edge_index, d["bond_index"])
    assert torch.equal(g["ligand"].edge_mask, d["edge_mask"])
    assert np.array_equal(np.asarray(g["ligand"].mask_rotate[0]), d["mask_rotate"].numpy())
    xyz, z, bonds = read_sdf(f"{REF_DATA}/1a0q_ligand.sdf", remove_hs=False)
    assert len(z) == 45 and int((z == 1).sum()) == 22 and len(bonds) == 45


def test_graph_builders_agree_with_the_synthetic_generators():
    """The networkx restatement of get_transformation_mask and the receptor graph against the independent implementations
    in diffdock_amd.synth (which the reference-executed fixtures were generated with)."""
    for seed in range(5):
        c = make_complex(seed=seed, n_res=40, n_lig=22 + seed)
        ei = c["ligand", "ligand"].edge_index.numpy()
        me, mr = transformation_mask(22 + seed, ei)
        assert np.array_equal(me, c["ligand"].edge_mask.numpy()) and np.array_equal(mr, c["ligand"].mask_rotate[0])
        rc = c["receptor"].pos.numpy()
        a, b = receptor_graph(rc, 15.0, 24), receptor_contact_graph(rc, 15.0, 24)   # torch.cdist float32 vs the generator's distances
        assert a.shape == b.shape and np.array_equal(a[1], b[1])
        sets = lambda e: [frozenset(e[0][e[1] == i]) for i in range(len(rc))]     # order inside exact ties (3.8 A chain steps) is free
        assert sum(x != y for x, y in zip(sets(a), sets(b))) <= 1
    ei, attr = ligand_bond_arrays([(0, 1, 1), (1, 2, 2), (2, 3, 4)])
    assert ei.tolist() == [[0, 1, 1, 2, 2, 3], [1, 0, 2, 1, 3, 2]] and attr.argmax(1).tolist() == [0, 0, 1, 1, 3, 3]


def test_configs0_plumbing_on_the_cpu_emulation():
    """BASELINE configs[0] geometry (data/1a0q, 2 samples) with the small preset through the emulated kernels, 2 steps."""
    r = subprocess.run(["make", "-j8", "-C", os.path.join(ROOT, "diffdock_amd", "csrc"), "emu"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]

    def make(cfg, sd):
        m = MIScoreModel(cfg, device="cpu", lib_path=EMU)
        m.load_state_dict(sd)
        m.set_tables(*tables())
        return m
    cases.config0_case(make, lambda b: b, TINY.replace(lm_embedding_type=None, num_conv_layers=3), steps=2)   # (4 steps, full width: -m gpu)
