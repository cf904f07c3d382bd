"""Minimal heterogeneous-graph containers with the attribute surface the reference's
hot path touches on a PyG `HeteroData` / `Batch`:

    data['ligand'].pos / .x / .batch / .edge_mask / .mask_rotate / .node_t / .num_nodes
    data['ligand', 'ligand'].edge_index / .edge_attr / .num_edges     (-> lig_bond relation)
    data['receptor'].pos / .x / .batch / .side_chain_vecs
    data['receptor', 'receptor'].edge_index                            (-> rec_contact relation)
    data['atom'].pos / .x / .batch, data['atom', 'atom'].edge_index, data['atom', 'receptor'].edge_index   (all_atoms)
    data.num_graphs, data.complex_t, data['name'], data.to(device), data.to_data_list()

torch_geometric is not installed in this image; real PyG batches expose the same
attributes, so everything that consumes these containers is duck-typed and accepts both.
"""
from __future__ import annotations

import copy
from typing import Dict, List

import numpy as np
import torch


class Store:
    """Attribute bag (node store or edge store)."""

    def __init__(self, **kw):
        self.__dict__.update(kw)

    # PyG-like conveniences ------------------------------------------------
    @property
    def num_nodes(self):
        return self.pos.shape[0] if "pos" in self.__dict__ else self.x.shape[0]

    @property
    def num_edges(self):
        return self.edge_index.shape[1]

    def keys(self):
        return self.__dict__.keys()

    def __contains__(self, k):
        return k in self.__dict__

    def __getitem__(self, k):
        return self.__dict__[k]

    def __setitem__(self, k, v):
        self.__dict__[k] = v


_REL = {("ligand", "ligand"): ("ligand", "lig_bond", "ligand"),
        ("receptor", "receptor"): ("receptor", "rec_contact", "receptor"),
        ("atom", "atom"): ("atom", "atom_contact", "atom"),                   # all-atom graphs (datasets/process_mols.py:238-239)
        ("atom", "receptor"): ("atom", "atom_rec_contact", "receptor")}


class HeteroData:
    """One complex (or, as HeteroBatch, a collated batch)."""

    def __init__(self):
        object.__setattr__(self, "_stores", {})
        object.__setattr__(self, "_globals", {})

    def _key(self, key):
        if isinstance(key, tuple):
            if len(key) == 2:
                key = _REL.get(key, (key[0], "to", key[1]))
            return key
        return key

    def __getitem__(self, key):
        key = self._key(key)
        if isinstance(key, str) and key in self._globals:
            return self._globals[key]
        if key not in self._stores:
            self._stores[key] = Store()
        return self._stores[key]

    def __setitem__(self, key, value):
        self._globals[key] = value

    def __getattr__(self, name):
        g = object.__getattribute__(self, "_globals")
        if name in g:
            return g[name]
        raise AttributeError(name)

    def __setattr__(self, name, value):
        self._globals[name] = value

    def __contains__(self, name):
        return name in self._globals

    @property
    def node_types(self):
        return [k for k in self._stores if isinstance(k, str)]

    @property
    def edge_types(self):
        return [k for k in self._stores if isinstance(k, tuple)]

    def to(self, device):
        def mv(v):
            if torch.is_tensor(v):
                return v.to(device)
            if isinstance(v, dict):
                return {k: mv(x) for k, x in v.items()}
            return v
        for st in self._stores.values():
            for k, v in list(st.__dict__.items()):
                st.__dict__[k] = mv(v)
        for k, v in list(self._globals.items()):
            self._globals[k] = mv(v)
        return self

    def clone(self):
        return copy.deepcopy(self)


class HeteroBatch(HeteroData):
    """Collation of HeteroData graphs with PyG `Batch.from_data_list` semantics:
    node/edge tensors concatenated along dim 0, edge_index offset by cumulative node
    counts, `.batch` vectors, non-tensor attributes collected into lists."""

    @classmethod
    def from_data_list(cls, data_list: List[HeteroData]) -> "HeteroBatch":
        out = cls()
        B = len(data_list)
        node_types = data_list[0].node_types
        offsets: Dict[str, List[int]] = {}
        for nt in node_types:
            offs, acc = [], 0
            for d in data_list:
                offs.append(acc)
                acc += d[nt].num_nodes
            offsets[nt] = offs
            keys = list(data_list[0][nt].keys())
            st = out[nt]
            for k in keys:
                vals = [d[nt][k] for d in data_list]
                if torch.is_tensor(vals[0]):
                    st[k] = torch.cat(vals, 0)
                else:
                    st[k] = vals
            st["batch"] = torch.cat([torch.full((d[nt].num_nodes,), i, dtype=torch.long)
                                     for i, d in enumerate(data_list)])
            st["ptr"] = torch.tensor(offs + [acc], dtype=torch.long)
        for et in data_list[0].edge_types:
            st = out[et]
            for k in list(data_list[0][et].keys()):
                vals = [d[et][k] for d in data_list]
                if k == "edge_index":
                    so, do = offsets[et[0]], offsets[et[2]]
                    st[k] = torch.cat([v + torch.tensor([[so[i]], [do[i]]], dtype=v.dtype)
                                       for i, v in enumerate(vals)], 1)
                elif torch.is_tensor(vals[0]):
                    st[k] = torch.cat(vals, 0)
                else:
                    st[k] = vals
        for k in data_list[0]._globals:
            vals = [d._globals[k] for d in data_list]
            out._globals[k] = torch.cat(vals, 0) if torch.is_tensor(vals[0]) and vals[0].dim() > 0 else vals
        out._globals["num_graphs"] = B
        out._globals["_slices"] = {nt: offsets[nt] + [out[nt].num_nodes] for nt in node_types}
        return out

    def to_data_list(self) -> List[HeteroData]:
        B = self.num_graphs
        sl = self._globals["_slices"]
        res = []
        for i in range(B):
            d = HeteroData()
            for nt in self.node_types:
                lo, hi = sl[nt][i], sl[nt][i + 1]
                n_tot = self[nt].num_nodes
                for k, v in self[nt].__dict__.items():
                    if k in ("batch", "ptr"):
                        continue
                    if torch.is_tensor(v):
                        if v.shape[0] == n_tot:
                            d[nt][k] = v[lo:hi]
                        else:  # edge-length tensor living in a node store (ligand.edge_mask)
                            per = v.shape[0] // B
                            d[nt][k] = v[i * per:(i + 1) * per]
                    elif isinstance(v, list):
                        d[nt][k] = v[i]
                    elif isinstance(v, dict):
                        d[nt][k] = {kk: vv[lo:hi] for kk, vv in v.items()}
            for et in self.edge_types:
                ei = self[et].edge_index
                slo, shi = sl[et[0]][i], sl[et[0]][i + 1]
                m = (ei[0] >= slo) & (ei[0] < shi)
                for k, v in self[et].__dict__.items():
                    if k == "edge_index":
                        d[et][k] = v[:, m] - torch.tensor([[slo], [sl[et[2]][i]]], dtype=v.dtype, device=v.device)
                    elif torch.is_tensor(v):
                        d[et][k] = v[m]
                    elif isinstance(v, list):
                        d[et][k] = v[i]
            for k, v in self._globals.items():
                if k in ("num_graphs", "_slices", "complex_t"):
                    continue
                d._globals[k] = v[i] if isinstance(v, list) else v
            res.append(d)
        return res


class DataLoader:
    """torch_geometric.loader.DataLoader stand-in: sequential mini-batches of
    HeteroBatch (reference utils/sampling.py:80)."""

    def __init__(self, data_list, batch_size=1, shuffle=False):
        assert not shuffle
        self.data_list, self.batch_size = list(data_list), batch_size

    def __iter__(self):
        for i in range(0, len(self.data_list), self.batch_size):
            yield HeteroBatch.from_data_list(self.data_list[i:i + self.batch_size])

    def __len__(self):
        return (len(self.data_list) + self.batch_size - 1) // self.batch_size


def set_time(batch, t_tr, t_rot, t_tor, batchsize, device=None):
    """utils/diffusion_utils.py:146-168 ('atom' nodes receive node_t when the graph has them = all_atoms)."""
    device = device or batch["ligand"].pos.device
    for nt in ("ligand", "receptor") + (("atom",) if "atom" in batch.node_types else ()):
        n = batch[nt].num_nodes
        batch[nt].node_t = {"tr": t_tr * torch.ones(n, device=device),
                            "rot": t_rot * torch.ones(n, device=device),
                            "tor": t_tor * torch.ones(n, device=device)}
    batch.complex_t = {"tr": t_tr * torch.ones(batchsize, device=device),
                       "rot": t_rot * torch.ones(batchsize, device=device),
                       "tor": t_tor * torch.ones(batchsize, device=device)}
    return batch


def as_numpy_mask(mask_rotate):
    """mask_rotate is a numpy bool [R, Nl] on a single graph and a list of them on a
    batch (Appendix A.9); sampling() uses data_list[0]['ligand'].mask_rotate[0]."""
    while isinstance(mask_rotate, (list, tuple)):
        mask_rotate = mask_rotate[0]
    if torch.is_tensor(mask_rotate):
        mask_rotate = mask_rotate.cpu().numpy()
    return np.asarray(mask_rotate, dtype=bool)
