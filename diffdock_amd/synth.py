"""Seeded synthetic complexes with the schema of the reference's preprocessed
`HeteroData`.  There is no RDKit / ProDy / ESM in this image and no
network, so inputs of PDBBind-like shape are generated:

  receptor  compact C-alpha walk (3.8 A steps), residue type + 1280-d "ESM" features,
            contact graph = neighbours within `receptor_radius`, at most
            `c_alpha_max_neighbors` nearest (datasets/process_mols.py:171-192:
            row 0 = neighbour, row 1 = centre)
  ligand    random tree with ring closures, 1.5 A bonds, categorical atom features
            within the reference vocabularies, both-direction interleaved bonds with
            one-hot types (process_mols.py:279-295), rotatable-bond `edge_mask` and
            `mask_rotate` with the semantics of utils/torsion.py:15-45
  pose      `randomize_position` semantics (utils/sampling.py:16-58)
"""
from __future__ import annotations

import numpy as np
import torch

from .config import LIG_FEATURE_DIMS, LM_EMBEDDING_DIM
from .hetero import HeteroData


def _receptor_coords(rng, nr):
    radius = 2.2 * nr ** (1 / 3) + 4.0
    pts = [np.zeros(3)]
    tries = 0
    while len(pts) < nr:
        d = rng.normal(size=3)
        d /= np.linalg.norm(d)
        cand = pts[-1] + 3.8 * d
        tries += 1
        if np.linalg.norm(cand) > radius and tries < 50:
            continue
        arr = np.asarray(pts)
        if tries < 50 and np.min(np.linalg.norm(arr - cand, axis=1)) < 3.0:
            continue
        pts.append(cand)
        tries = 0
    pts = np.asarray(pts, dtype=np.float64)
    return (pts - pts.mean(0)).astype(np.float32)


def receptor_contact_graph(coords, cutoff, max_neighbors):
    """process_mols.py:174-192 (non-kNN branch)."""
    d = np.linalg.norm(coords[:, None] - coords[None], axis=-1)
    src_list, dst_list = [], []
    for i in range(len(coords)):
        dst = list(np.where(d[i] < cutoff)[0])
        dst.remove(i)
        if max_neighbors is not None and len(dst) > max_neighbors:
            dst = list(np.argsort(d[i], kind="stable"))[1:max_neighbors + 1]
        if len(dst) == 0:
            dst = list(np.argsort(d[i], kind="stable"))[1:2]
        src_list += [i] * len(dst)
        dst_list += [int(x) for x in dst]
    return np.asarray([dst_list, src_list], dtype=np.int64)


def _ligand_topology(rng, nl):
    """Random tree grown atom by atom (degree <= 4) plus a few ring closures."""
    bonds, deg = [], np.zeros(nl, dtype=int)
    for i in range(1, nl):
        cand = [j for j in range(max(0, i - 6), i) if deg[j] < 3]
        j = int(rng.choice(cand)) if cand else i - 1
        bonds.append((j, i))
        deg[j] += 1
        deg[i] += 1
    # ring closures between atoms 4-5 bonds apart along the index chain
    for _ in range(max(1, nl // 10)):
        a = int(rng.integers(0, max(1, nl - 5)))
        b = min(nl - 1, a + int(rng.integers(4, 6)))
        if a != b and (a, b) not in bonds and (b, a) not in bonds and deg[a] < 4 and deg[b] < 4:
            bonds.append((a, b))
            deg[a] += 1
            deg[b] += 1
    return bonds


def _ligand_coords(rng, nl, bonds):
    """Place atoms along the tree with ~1.5 A bonds and soft self-avoidance; ring-closing
    bonds only constrain topology (geometry quality is irrelevant to the arithmetic)."""
    pos = np.zeros((nl, 3))
    placed = {0}
    for (j, i) in bonds:
        if i in placed:
            continue
        best, best_d = None, -1.0
        for _ in range(30):
            d = rng.normal(size=3)
            d /= np.linalg.norm(d)
            cand = pos[j] + 1.5 * d
            others = pos[sorted(placed - {j})] if len(placed) > 1 else None
            md = np.min(np.linalg.norm(others - cand, axis=1)) if others is not None and len(others) else 9.0
            if md > best_d:
                best, best_d = cand, md
            if md > 2.1:
                break
        pos[i] = best
        placed.add(i)
    return (pos - pos.mean(0)).astype(np.float32)


def transformation_mask(nl, bonds):
    """utils/torsion.py:15-45: a bond is rotatable iff removing it disconnects the graph
    and the smaller side has more than one atom; the directed edge whose head (v) lies in
    the rotated side carries the mask."""
    adj = [[] for _ in range(nl)]
    for a, b in bonds:
        adj[a].append(b)
        adj[b].append(a)

    def component(start, banned):
        seen, stack = {start}, [start]
        while stack:
            x = stack.pop()
            for y in adj[x]:
                if (x, y) == banned or (y, x) == banned:
                    continue
                if y not in seen:
                    seen.add(y)
                    stack.append(y)
        return seen

    to_rotate = []
    for (a, b) in bonds:  # directed edges 2k = (a,b), 2k+1 = (b,a)
        ca = component(a, (a, b))
        if b in ca:
            to_rotate += [[], []]
            continue
        cb = set(range(nl)) - ca
        small = sorted(ca) if len(ca) <= len(cb) else sorted(cb)
        # reference sorts components by len (stable) and takes the first; on ties the
        # component found first by networkx wins -- avoid ties mattering: either is valid.
        if len(small) > 1:
            if a in small:
                to_rotate += [[], small]
            else:
                to_rotate += [small, []]
        else:
            to_rotate += [[], []]
    mask_edges = np.asarray([len(l) > 0 for l in to_rotate], dtype=bool)
    mask_rotate = np.zeros((int(mask_edges.sum()), nl), dtype=bool)
    idx = 0
    for i, l in enumerate(to_rotate):
        if mask_edges[i]:
            mask_rotate[idx][np.asarray(l, dtype=int)] = True
            idx += 1
    return mask_edges, mask_rotate


def receptor_atoms(rng, rc, restype, atoms_per_res=(4, 11), atom_radius=5.0, atom_max_neighbors=8):
    """Heavy atoms of the all-atom graphs (datasets/process_mols.py:204-239 semantics): a few atoms around every
    C-alpha, categorical features (amino acid, atomic number, atom_type_2, atom_type_3), atom-atom edges = neighbours
    within atom_radius capped at the atom_max_neighbors nearest (nearest one if none), edge_index = [neighbour; atom],
    and one atom -> own residue edge per atom."""
    from .config import REC_ATOM_FEATURE_DIMS
    n_res = len(rc)
    counts = rng.integers(atoms_per_res[0], atoms_per_res[1] + 1, size=n_res)
    res_of = np.repeat(np.arange(n_res), counts)
    na = len(res_of)
    pos = rc[res_of] + rng.normal(size=(na, 3)) * 1.3
    first = np.concatenate([[0], np.cumsum(counts)[:-1]])
    pos[first] = rc                                   # the C-alpha itself
    feats = np.stack([restype[res_of, 0].astype(np.int64)] +
                     [rng.integers(0, d, size=na) for d in REC_ATOM_FEATURE_DIMS[1:]], 1)
    feats[:, 1] = rng.choice([5, 6, 7, 15], size=na, p=[0.62, 0.17, 0.2, 0.01])
    d = np.linalg.norm(pos[:, None, :] - pos[None, :, :], axis=-1)
    src, dst = [], []
    for i in range(na):
        nb = [j for j in np.where(d[i] < atom_radius)[0] if j != i]
        if len(nb) > atom_max_neighbors:
            nb = list(np.argsort(d[i])[1:atom_max_neighbors + 1])
        if len(nb) == 0:
            nb = list(np.argsort(d[i])[1:2])
        src += [i] * len(nb)
        dst += [int(j) for j in nb]
    aa = np.asarray([dst, src], dtype=np.int64)
    ar = np.stack([np.arange(na), res_of]).astype(np.int64)
    return pos.astype(np.float32), feats.astype(np.float32), aa, ar


def make_complex(seed=0, n_res=300, n_lig=30, receptor_radius=15.0, c_alpha_max_neighbors=24,
                 lm_dim=LM_EMBEDDING_DIM, name=None, all_atoms=False, atoms_per_res=(4, 11)) -> HeteroData:
    rng = np.random.default_rng(seed)
    g = HeteroData()
    # receptor ------------------------------------------------------------
    rc = _receptor_coords(rng, n_res)
    restype = rng.integers(0, 20, size=(n_res, 1)).astype(np.float32)
    esm = (rng.normal(size=(n_res, lm_dim)) * 0.2).astype(np.float32)
    g["receptor"].x = torch.from_numpy(np.concatenate([restype, esm], 1))
    g["receptor"].pos = torch.from_numpy(rc)
    g["receptor"].side_chain_vecs = torch.zeros(n_res, 10)
    g["receptor", "rec_contact", "receptor"].edge_index = torch.from_numpy(
        receptor_contact_graph(rc, receptor_radius, c_alpha_max_neighbors))
    if all_atoms:
        ap, af, aa, ar = receptor_atoms(np.random.default_rng(seed + 7919), rc, restype, atoms_per_res)
        g["atom"].x = torch.from_numpy(af)
        g["atom"].pos = torch.from_numpy(ap)
        g["atom", "atom_contact", "atom"].edge_index = torch.from_numpy(aa)
        g["atom", "atom_rec_contact", "receptor"].edge_index = torch.from_numpy(ar)
    # ligand --------------------------------------------------------------
    bonds = _ligand_topology(rng, n_lig)
    lc = _ligand_coords(rng, n_lig, bonds)
    feats = np.stack([rng.integers(0, d, size=n_lig) for d in LIG_FEATURE_DIMS], 1)
    feats[:, 0] = rng.choice([5, 6, 7, 15], size=n_lig, p=[0.7, 0.12, 0.15, 0.03])  # C N O S
    row, col, et = [], [], []
    for (a, b) in bonds:
        row += [a, b]
        col += [b, a]
        et += 2 * [int(rng.choice(4, p=[0.7, 0.1, 0.02, 0.18]))]
    edge_attr = np.zeros((len(et), 4), dtype=np.float32)
    edge_attr[np.arange(len(et)), et] = 1.0
    mask_edges, mask_rotate = transformation_mask(n_lig, bonds)
    g["ligand"].x = torch.from_numpy(feats.astype(np.int64))
    g["ligand"].pos = torch.from_numpy(lc)
    g["ligand"].edge_mask = torch.from_numpy(mask_edges)
    g["ligand"].mask_rotate = [mask_rotate]  # list-of-one, as after a batch_size=1 PyG loader
    g["ligand", "lig_bond", "ligand"].edge_index = torch.tensor([row, col], dtype=torch.long)
    g["ligand", "lig_bond", "ligand"].edge_attr = torch.from_numpy(edge_attr)
    g.name = name or f"synth_{seed}_{n_res}_{n_lig}"
    g.original_center = torch.zeros(1, 3)
    return g


def _rotvec_to_matrix(v):
    th = np.linalg.norm(v)
    if th < 1e-12:
        return np.eye(3)
    k = v / th
    K = np.array([[0, -k[2], k[1]], [k[2], 0, -k[0]], [-k[1], k[0], 0]])
    return np.eye(3) + np.sin(th) * K + (1 - np.cos(th)) * K @ K


def _torsion_update_numpy(pos, rot_edges, mask_rotate, updates):
    """utils/torsion.py:48-72 (numpy, sequential)."""
    pos = pos.copy()
    for idx, (u, v) in enumerate(rot_edges):
        if updates[idx] == 0:
            continue
        vec = pos[u] - pos[v]
        vec = vec * updates[idx] / np.linalg.norm(vec)
        R = _rotvec_to_matrix(vec)
        m = mask_rotate[idx]
        pos[m] = (pos[m] - pos[v]) @ R.T + pos[v]
    return pos


def randomize_position(data_list, no_torsion, no_random, tr_sigma_max, initial_noise_std_proportion=-1.0, seed=0, draws=None):
    """utils/sampling.py:16-58 with an explicit seeded generator (the reference uses the global numpy / scipy / torch
    RNGs).  `draws` = dict(torsion=[per complex: R angles], rotation=[per complex: 3x3 matrix], tr=[per complex: (1,3)
    translation, already scaled]) replaces the generator -- that is how the function is pinned to the reference-executed
    fixture tests/golden/randpos.pt, which records the reference's own draws."""
    rng = np.random.default_rng(seed)
    center_pocket = data_list[0]["receptor"].pos.mean(dim=0)
    for i, g in enumerate(data_list):
        pos = g["ligand"].pos.double().numpy()
        if not no_torsion:
            ei = g["ligand", "ligand"].edge_index.T.numpy()[g["ligand"].edge_mask.numpy()]
            upd = rng.uniform(-np.pi, np.pi, size=len(ei)) if draws is None else np.asarray(draws["torsion"][i], dtype=np.float64)
            pos = _torsion_update_numpy(pos, ei, g["ligand"].mask_rotate[0], upd)
        if draws is None:
            q = rng.normal(size=4)
            q /= np.linalg.norm(q)
            w, x, y, z = q
            R = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                          [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                          [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
        else:
            R = np.asarray(draws["rotation"][i], dtype=np.float64)
        pos = (pos - pos.mean(0, keepdims=True)) @ R.T + center_pocket.double().numpy()
        if not no_random:
            if draws is not None:
                pos = pos + np.asarray(draws["tr"][i], dtype=np.float64).reshape(1, 3)
            else:
                if initial_noise_std_proportion >= 0.0:
                    std_rec = float(torch.sqrt(torch.mean(torch.sum(g["receptor"].pos ** 2, dim=1))))
                    std = std_rec * initial_noise_std_proportion / 1.73
                else:
                    std = -initial_noise_std_proportion * tr_sigma_max
                pos = pos + rng.normal(size=(1, 3)) * std
        g["ligand"].pos = torch.from_numpy(pos.astype(np.float32))
    return data_list


def make_pose_list(complex_graph, n_samples, tr_sigma_max=19.0, seed=0, initial_noise_std_proportion=1.46,
                   no_torsion=False):
    """inference.py:239-242: N deep copies of one complex, each with a random initial pose."""
    lst = [complex_graph.clone() for _ in range(n_samples)]
    return randomize_position(lst, no_torsion, False, tr_sigma_max,
                              initial_noise_std_proportion=initial_noise_std_proportion, seed=seed)
