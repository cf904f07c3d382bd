"""Minimal featurisation I/O: PDB -> C-alpha receptor graph, SDF/MOL (V2000) -> ligand geometry graph.

What the reference does with ProDy + RDKit (datasets/process_mols.py:117-200 receptor, :279-301 ligand,
utils/torsion.py:15-45 rotatable-bond masks) restated for the parts that are plain geometry / graph theory:

  read_pdb_calpha      residues that have a CA atom (ProDy's `pdb.ca`), residue-type index in
                       `possible_amino_acids` (process_mols.py:48-50; non-standard names -> 'misc')
  receptor_graph       `new_extract_receptor_structure` (:161-200): neighbours within `neighbor_cutoff`, the
                       `max_neighbors` nearest if there are more, the nearest one if there is none; edge_index = [neighbour; centre]
  read_sdf             V2000 atom / bond blocks; hydrogens dropped like `remove_hs=True`
  ligand_graph         get_lig_graph (:279-301): both directions interleaved, one-hot bond type (single, double, triple, aromatic)
  transformation_mask  get_transformation_mask (utils/torsion.py:15-45) with networkx, same component choice

  ligand_atom_features lig_atom_featurizer (:97-120) for the columns that follow from the connection table alone: atomic number,
                       total degree, formal charge, hydrogen count, aromatic flag, ring count and ring-size flags

NOT restated (declared external inputs): the RDKit chemistry-PERCEPTION columns of the 16 ligand atom features -- chirality
tag (column 1), implicit valence (4), radical electrons (6, read from the file's charge code / `M  RAD` only) and
hybridisation (7) stay 0 unless `atom_features` is passed in; aromaticity is taken from the file's bond type 4 (a kekulised
file keeps its single / double types and its atoms stay non-aromatic -- RDKit would perceive them); rings are a minimum
cycle basis (networkx), which equals RDKit's SSSR except for the symmetrised extra rings of cage systems; hydrogens are counted
only when the file writes them (a file without hydrogens gives degree = heavy neighbours, numH = 0 -- RDKit adds implicit
ones from its valence model).  The ESM language-model embeddings (`lm_embeddings`) are external as well.
"""
from __future__ import annotations

import numpy as np
import torch

from .hetero import HeteroData

POSSIBLE_AMINO_ACIDS = ['ALA', 'ARG', 'ASN', 'ASP', 'CYS', 'GLN', 'GLU', 'GLY', 'HIS', 'ILE', 'LEU', 'LYS', 'MET',
                        'PHE', 'PRO', 'SER', 'THR', 'TRP', 'TYR', 'VAL', 'HIP', 'HIE', 'TPO', 'HID', 'LEV', 'MEU',
                        'PTR', 'GLV', 'CYT', 'SEP', 'HIZ', 'CYM', 'GLM', 'ASQ', 'TYS', 'CYX', 'GLZ', 'misc']
_STANDARD = set(POSSIBLE_AMINO_ACIDS[:20])
_ELEMENTS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y Zr "
             "Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re Os Ir "
             "Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr Rf Db Sg Bh Hs Mt Ds").split()
_Z = {e.upper(): i + 1 for i, e in enumerate(_ELEMENTS)}
BOND_TYPES = {1: 0, 2: 1, 3: 2, 4: 3}     # SDF bond order -> index in {SINGLE, DOUBLE, TRIPLE, AROMATIC} (process_mols.py:21)


# Modified residues that ProDy's `protein` selection keeps although they are written as HETATM records.  LIMITATION: this is a
# hand-picked subset of ProDy's non-standard amino-acid table (not importable here); a file with another modified residue
# (MLZ, M3L, CSX, OCS, ...) yields FEWER residues than the reference pipeline, which would shift the language-model
# embedding alignment -- the length check of `complex_graph` (language-model rows vs residues) raises in that case instead of
# misaligning.  Pass ProDy's list through `extra_het_residues` of the readers when such files appear.
_HET_RESIDUES = {"MSE", "SEP", "TPO", "PTR", "CSO", "HYP", "MLY", "KCX", "CME", "CSD", "SEC", "PYL"}


def _pdb_residues(path, extra_het_residues=()):
    """Residues with a CA atom, in file order: {name, chain, N, CA, C}; alternate locations other than ' ' / 'A' are
    skipped; HETATM records count only for the modified amino acids ProDy's `protein` selection keeps (MSE, SEP, ...)."""
    res, order = {}, []
    with open(path) as f:
        for line in f:
            het = line.startswith("HETATM")
            if not (line.startswith("ATOM") or (het and (line[17:20].strip() in _HET_RESIDUES or line[17:20].strip() in extra_het_residues))):
                continue
            atom = line[12:16].strip()
            if atom not in ("N", "CA", "C") or line[16] not in (" ", "A"):
                continue
            key = (line[21], line[22:27])
            if key not in res:
                res[key] = {"name": line[17:20].strip(), "chain": line[21]}
                order.append(key)
            res[key].setdefault(atom, [float(line[30:38]), float(line[38:46]), float(line[46:54])])
    return [res[k] for k in order if "CA" in res[k]]


def read_pdb_calpha(path, extra_het_residues=()):
    """-> (coords float64 [n,3], residue type index int64 [n], chain ids list[str]) for every residue with a CA atom, in
    file order."""
    rs = _pdb_residues(path, extra_het_residues)
    # the reference goes through one-letter codes (pdb.ca.getSequence -> aa_short2long): anything ProDy does not map
    # to one of the 20 letters becomes 'misc'
    types = [POSSIBLE_AMINO_ACIDS.index(r["name"]) if r["name"] in _STANDARD else len(POSSIBLE_AMINO_ACIDS) - 1 for r in rs]
    return (np.asarray([r["CA"] for r in rs], dtype=np.float64).reshape(-1, 3), np.asarray(types, dtype=np.int64),
            [r["chain"] for r in rs])


def read_pdb_backbone(path, extra_het_residues=()):
    """-> (N / CA / C coordinates float64 [n, 3, 3] (nan where an atom is missing), residue names) of the same residues."""
    rs = _pdb_residues(path, extra_het_residues)
    nan = [float("nan")] * 3
    return np.asarray([[r.get("N", nan), r["CA"], r.get("C", nan)] for r in rs], dtype=np.float64).reshape(-1, 3, 3), [r["name"] for r in rs]


def receptor_graph(coords, neighbor_cutoff=15.0, max_neighbors=24):
    """process_mols.py:171-192 (non-kNN branch): `torch.cdist` distances in float32 on the float32 (uncentred) coordinates,
    exactly the reference's call -- neighbour sets at the cutoff and the order of near-ties follow that arithmetic."""
    d = torch.cdist(torch.as_tensor(np.asarray(coords), dtype=torch.float32), torch.as_tensor(np.asarray(coords), dtype=torch.float32)).numpy()
    c = d
    src_list, dst_list = [], []
    for i in range(len(c)):
        dst = list(np.where(d[i] < neighbor_cutoff)[0])
        dst.remove(i)
        cap = max_neighbors if max_neighbors else 1000
        if len(dst) > cap:
            dst = list(np.argsort(d[i]))[1:cap + 1]
        if len(dst) == 0:
            dst = list(np.argsort(d[i]))[1:2]
        src_list += [i] * len(dst)
        dst_list += [int(x) for x in dst]
    return np.asarray([dst_list, src_list], dtype=np.int64)


def read_sdf(path, remove_hs=True):
    """First molecule of a V2000 mol / sdf file -> (coords [n,3], atomic numbers [n], bonds [(a, b, order)])."""
    with open(path) as f:
        lines = f.read().splitlines()
    counts = lines[3]
    na, nb = int(counts[0:3]), int(counts[3:6])
    if "V2000" not in counts:
        raise ValueError("only V2000 connection tables are read")
    xyz, z = [], []
    for l in lines[4:4 + na]:
        xyz.append([float(l[0:10]), float(l[10:20]), float(l[20:30])])
        z.append(_Z.get(l[31:34].strip().upper(), 0))
    bonds = []
    for l in lines[4 + na:4 + na + nb]:
        bonds.append((int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])))
    xyz, z = np.asarray(xyz, dtype=np.float64), np.asarray(z, dtype=np.int64)
    if remove_hs:
        keep = np.where(z != 1)[0]
        remap = -np.ones(na, dtype=np.int64)
        remap[keep] = np.arange(len(keep))
        bonds = [(int(remap[a]), int(remap[b]), o) for a, b, o in bonds if remap[a] >= 0 and remap[b] >= 0]
        xyz, z = xyz[keep], z[keep]
    return xyz, z, bonds


_CHARGE_CODE = {0: 0, 1: 3, 2: 2, 3: 1, 4: 0, 5: -1, 6: -2, 7: -3}     # V2000 atom-block charge field (4 = doublet radical)


def ligand_atom_features(path):
    """[n_heavy, 16] int64: the columns of `lig_atom_featurizer` (datasets/process_mols.py:97-120, feature order of
    `lig_feature_dims` :65-82) that follow from the V2000 connection table without chemistry perception; the others stay 0
    (module docstring).  Rows are the heavy atoms in file order, as `read_sdf(remove_hs=True)` returns them.

      0 atomic number          safe_index(range(1, 119) + ['misc'], Z)
      2 total degree           bonded atoms in the file, hydrogens included (GetTotalDegree; 11 = 'misc' above 10)
      3 formal charge          atom-block charge code, overridden by `M  CHG` lines; index into [-5 .. 5, 'misc']
      5 hydrogen count         bonded hydrogen atoms in the file (GetTotalNumHs; 9 = 'misc' above 8)
      6 radical electrons      `M  RAD` (1 singlet -> 2 electrons as RDKit, 2 doublet -> 1, 3 triplet -> 2) / charge code 4
      8 aromatic               the atom has a bond of type 4
      9 ring count             cycles of a minimum cycle basis through the atom (7 = 'misc' above 6)
      10..15 in a ring of size 3..8"""
    import networkx as nx
    with open(path) as f:
        lines = f.read().splitlines()
    counts = lines[3]
    if "V2000" not in counts:
        raise ValueError("only V2000 connection tables are read")
    na, nb = int(counts[0:3]), int(counts[3:6])
    z, charge, rad = [], [], []
    for l in lines[4:4 + na]:
        z.append(_Z.get(l[31:34].strip().upper(), 0))
        code = int(l[36:39]) if len(l) >= 39 and l[36:39].strip() else 0
        charge.append(_CHARGE_CODE.get(code, 0))
        rad.append(1 if code == 4 else 0)
    bonds = [(int(l[0:3]) - 1, int(l[3:6]) - 1, int(l[6:9])) for l in lines[4 + na:4 + na + nb]]
    chg_lines = [l for l in lines[4 + na + nb:] if l.startswith("M  CHG")]
    if chg_lines:   # (a CHG property line supersedes every atom-block charge, ctfile specification)
        charge = [0] * na
    for l in lines[4 + na + nb:]:
        if l.startswith("M  END"):
            break
        if l.startswith("M  CHG") or l.startswith("M  RAD"):
            f_ = l.split()
            for a, v in zip(f_[3::2], f_[4::2]):
                if l.startswith("M  CHG"):
                    charge[int(a) - 1] = int(v)
                else:
                    rad[int(a) - 1] = {1: 2, 2: 1, 3: 2}.get(int(v), 0)
    z = np.asarray(z, dtype=np.int64)
    heavy = np.where(z != 1)[0]
    deg = np.zeros(na, dtype=np.int64); nh = np.zeros(na, dtype=np.int64); arom = np.zeros(na, dtype=bool)
    G = nx.Graph()
    G.add_nodes_from(int(i) for i in heavy)
    for a, b, o in bonds:
        deg[a] += 1; deg[b] += 1
        if z[b] == 1: nh[a] += 1
        if z[a] == 1: nh[b] += 1
        if o == 4: arom[a] = arom[b] = True
        if z[a] != 1 and z[b] != 1:
            G.add_edge(a, b)
    rings = [list(c) for c in nx.minimum_cycle_basis(G)]
    feats = np.zeros((len(heavy), 16), dtype=np.int64)
    for r, i in enumerate(heavy):
        mine = [c for c in rings if i in c]
        feats[r, 0] = z[i] - 1 if 1 <= z[i] <= 118 else 118
        feats[r, 2] = min(deg[i], 11)
        feats[r, 3] = charge[i] + 5 if -5 <= charge[i] <= 5 else 11
        feats[r, 5] = min(nh[i], 9)
        feats[r, 6] = min(rad[i], 5)
        feats[r, 8] = int(arom[i])
        feats[r, 9] = min(len(mine), 7)
        for size in range(3, 9):
            feats[r, 7 + size] = int(any(len(c) == size for c in mine))
    return feats


def ligand_bond_arrays(bonds):
    """get_lig_graph: edge_index [2, 2*nb] (both directions interleaved) and one-hot edge_attr [2*nb, 4]."""
    row, col, et = [], [], []
    for a, b, o in bonds:
        row += [a, b]
        col += [b, a]
        et += 2 * [BOND_TYPES.get(o, 0)]
    attr = np.zeros((len(et), 4), dtype=np.float32)
    attr[np.arange(len(et)), et] = 1.0
    return np.asarray([row, col], dtype=np.int64).reshape(2, -1), attr


def transformation_mask(n_atoms, edge_index):
    """utils/torsion.py:15-45 with networkx (same calls: undirected copy, remove_edge, is_connected,
    sorted(connected_components, key=len)[0]).  edge_index: [2, 2*nb], directed pairs interleaved."""
    import networkx as nx
    edges = np.asarray(edge_index).T
    G = nx.DiGraph()
    G.add_nodes_from(range(n_atoms))
    G.add_edges_from((int(a), int(b)) for a, b in edges)
    to_rotate = []
    for i in range(0, edges.shape[0], 2):
        assert edges[i, 0] == edges[i + 1, 1]
        G2 = G.to_undirected()
        G2.remove_edge(*edges[i])
        if not nx.is_connected(G2):
            l = list(sorted(nx.connected_components(G2), key=len)[0])
            if len(l) > 1:
                if edges[i, 0] in l:
                    to_rotate += [[], l]
                else:
                    to_rotate += [l, []]
                continue
        to_rotate += [[], []]
    mask_edges = np.asarray([len(l) > 0 for l in to_rotate], dtype=bool)
    mask_rotate = np.zeros((int(mask_edges.sum()), n_atoms), dtype=bool)
    idx = 0
    for i, l in enumerate(to_rotate):
        if mask_edges[i]:
            mask_rotate[idx][np.asarray(l, dtype=int)] = True
            idx += 1
    return mask_edges, mask_rotate


def complex_graph(pdb_path, sdf_path, receptor_radius=15.0, c_alpha_max_neighbors=24, lm_embeddings=None, atom_features=None,
                  lm_dim=1280, name=None) -> HeteroData:
    """HeteroData with the schema of the reference's preprocessed complex: receptor C-alpha graph, ligand
    heavy-atom graph with rotatable-bond masks, both centred on the receptor's C-alpha centroid (`original_center`,
    datasets/pdbbind.py get_complex).  `lm_embeddings` [n_res, 1280] and `atom_features` [n_lig, 16] are the external
    inputs (ESM, RDKit); zeros / the connection-table columns of `ligand_atom_features` when absent."""
    rc, rtype, _ = read_pdb_calpha(pdb_path)
    lc, z, bonds = read_sdf(sdf_path)
    g = HeteroData()
    # the reference builds the graph on the uncentred float32 coordinates (process_mols.py:167-192) and centres afterwards
    # with a float32 mean (datasets/pdbbind.py:405-416)
    rpos = torch.from_numpy(rc.astype(np.float32))
    if lm_embeddings is not None and len(lm_embeddings) != len(rc):
        raise ValueError(f"{len(lm_embeddings)} language-model embedding rows for {len(rc)} residues with a CA atom")
    lm = torch.zeros(len(rc), lm_dim) if lm_embeddings is None else torch.as_tensor(lm_embeddings, dtype=torch.float32)
    g["receptor"].x = torch.cat([torch.from_numpy(rtype.astype(np.float32))[:, None], lm], 1)
    g["receptor"].side_chain_vecs = torch.zeros(len(rc), 10)
    g["receptor", "rec_contact", "receptor"].edge_index = torch.from_numpy(
        receptor_graph(rpos.numpy(), receptor_radius, c_alpha_max_neighbors))
    center = torch.mean(rpos, dim=0, keepdim=True)
    g["receptor"].pos = rpos - center
    ei, attr = ligand_bond_arrays(bonds)
    if atom_features is None:       # the connection-table columns; the perception columns stay 0 (module docstring)
        feats = ligand_atom_features(sdf_path)
        assert feats.shape[0] == len(z) and np.array_equal(feats[:, 0], np.where((z >= 1) & (z <= 118), z - 1, 118))
    else:
        feats = np.asarray(atom_features, dtype=np.int64)
    mask_edges, mask_rotate = transformation_mask(len(z), ei)
    g["ligand"].x = torch.from_numpy(feats)
    g["ligand"].pos = torch.from_numpy(lc.astype(np.float32)) - center
    g["ligand"].edge_mask = torch.from_numpy(mask_edges)
    g["ligand"].mask_rotate = [mask_rotate]
    g["ligand", "lig_bond", "ligand"].edge_index = torch.from_numpy(ei)
    g["ligand", "lig_bond", "ligand"].edge_attr = torch.from_numpy(attr)
    g.name = name or "complex"
    g.original_center = center
    return g
