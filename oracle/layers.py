"""Oracle restatement of models/layers.py and models/tensor_layers.py.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Functional style over a
reference-keyed state_dict `sd` (prefix = module path).

  linear / fc_block            models/layers.py:10-17 (FCBlock = Linear, act, Dropout, ..., Linear)
  gaussian_smearing            models/layers.py:20-30
  atom_encoder                 models/layers.py:33-67
  faster_tensor_product        models/tensor_layers.py:44-122
  tp_conv_layer                models/tensor_layers.py:125-231 (tp_scatter_*), :309-335 (layer forward)
"""
import math

import torch
import torch.nn.functional as F

from .e3nn_lite import Irreps, FullyConnectedTensorProduct, batch_norm_eval
from .graph_ops import scatter


def linear(sd, name, x):
    w = sd[name + ".weight"]
    b = sd.get(name + ".bias")
    return F.linear(x, w, b)


def mlp2(sd, name, x, act=torch.relu, second="3"):
    """nn.Sequential(Linear, ReLU, Dropout, Linear) in eval mode (keys .0 / .3); the
    tr/rot final layers are (Linear, Dropout, ReLU, Linear) -- same keys, same math."""
    return linear(sd, f"{name}.{second}", act(linear(sd, f"{name}.0", x)))


def fc_block(sd, name, x, layers):
    """FCBlock(in, hidden, out, layers) in eval mode (models/layers.py:10-17): Linear keys 0, 3, .. 3(layers-1), ReLU between."""
    for j in range(layers - 1):
        x = torch.relu(linear(sd, f"{name}.{3 * j}", x))
    return linear(sd, f"{name}.{3 * (layers - 1)}", x)


def gaussian_smearing(offset, dist):
    coeff = -0.5 / (offset[1] - offset[0]).item() ** 2
    d = dist.reshape(-1, 1) - offset.reshape(1, -1)
    return torch.exp(coeff * d.pow(2))


def atom_encoder(sd, name, x, n_cat):
    emb = 0
    for i in range(n_cat):
        emb = emb + sd[f"{name}.atom_embedding_list.{i}.weight"][x[:, i].long()]
    if f"{name}.additional_features_embedder.weight" in sd:
        emb = linear(sd, f"{name}.additional_features_embedder", torch.cat([emb, x[:, n_cat:].to(emb.dtype)], 1))
    return emb


_F_TYPES = ("0e", "1o", "1e", "0o")


def faster_tensor_product(in_irreps, out_irreps, x, sh, weight):
    """models/tensor_layers.py:71-122, sh = [1x0e + 1x1o]."""
    in_irreps, out_irreps = Irreps(in_irreps), Irreps(out_irreps)
    ind, im, om = {}, {t: 0 for t in _F_TYPES}, {t: 0 for t in _F_TYPES}
    for mi, sl in zip(in_irreps, in_irreps.slices()):
        v = x[..., sl]
        if mi.ir.l == 1:
            v = v.reshape(v.shape[0], mi.mul, 3)
        ind[str(mi.ir)] = v
        im[str(mi.ir)] = mi.mul
    for mi in out_irreps:
        om[str(mi.ir)] = mi.mul
    s0, s1 = sh[:, 0], sh[:, 1:]
    terms = {t: [] for t in _F_TYPES}
    if "0e" in ind:
        terms["0e"].append(ind["0e"] * s0[:, None])
        terms["1o"].append(ind["0e"][:, :, None] * s1[:, None, :])
    if "1o" in ind:
        terms["0e"].append((ind["1o"] * s1[:, None, :]).sum(-1) / math.sqrt(3))
        terms["1o"].append(ind["1o"] * s0[:, None, None])
        terms["1e"].append(torch.linalg.cross(ind["1o"], s1[:, None, :].expand_as(ind["1o"]), dim=-1) / math.sqrt(2))
    if "1e" in ind:
        terms["1o"].append(torch.linalg.cross(ind["1e"], s1[:, None, :].expand_as(ind["1e"]), dim=-1) / math.sqrt(2))
        terms["1e"].append(ind["1e"] * s0[:, None, None])
        terms["0o"].append((ind["1e"] * s1[:, None, :]).sum(-1) / math.sqrt(3))
    if "0o" in ind:
        terms["1e"].append(ind["0o"][:, :, None] * s1[:, None, :])
        terms["0o"].append(ind["0o"] * s0[:, None])
    shapes = {"0e": (im["0e"] + im["1o"], om["0e"]), "1o": (im["0e"] + im["1o"] + im["1e"], om["1o"]),
              "1e": (im["1o"] + im["1e"] + im["0o"], om["1e"]), "0o": (im["1e"] + im["0o"], om["0o"])}
    wd, start = {}, 0
    for t in _F_TYPES:
        a, b = shapes[t]
        wd[t] = weight[:, start:start + a * b].reshape(weight.shape[0], a, b) / math.sqrt(a) if a > 0 else None
        start += a * b
    outd = {}
    for t in _F_TYPES:
        if not terms[t]:
            continue
        if t in ("0e", "0o"):
            z = torch.cat(terms[t], -1)                      # [E, fan]
            outd[t] = torch.einsum("eu,euw->ew", z, wd[t])
        else:
            z = torch.cat(terms[t], -2)                      # [E, fan, 3]
            outd[t] = torch.einsum("eum,euw->ewm", z, wd[t]).reshape(z.shape[0], 3 * om[t])
    return torch.cat([outd[str(mi.ir)] for mi in out_irreps], -1)


def faster_weight_numel(in_irreps, out_irreps):
    im, om = {t: 0 for t in _F_TYPES}, {t: 0 for t in _F_TYPES}
    for mi in Irreps(in_irreps):
        im[str(mi.ir)] = mi.mul
    for mi in Irreps(out_irreps):
        om[str(mi.ir)] = mi.mul
    return ((im["0e"] + im["1o"]) * om["0e"] + (im["0e"] + im["1o"] + im["1e"]) * om["1o"]
            + (im["1o"] + im["1e"] + im["0o"]) * om["1e"] + (im["1e"] + im["0o"]) * om["0o"])


class TPConv:
    """One TensorProductConvLayer (eval mode) bound to a state_dict prefix."""

    def __init__(self, sd, name, in_irreps, sh_irreps, out_irreps, residual=True, batch_norm=True,
                 faster=False, edge_groups=1, tp_weights_layers=2, depthwise=False):
        self.tp_weights_layers = tp_weights_layers
        self.sd, self.name = sd, name
        self.in_irreps, self.sh_irreps, self.out_irreps = Irreps(in_irreps), Irreps(sh_irreps), Irreps(out_irreps)
        self.residual, self.batch_norm, self.faster, self.edge_groups = residual, batch_norm, faster, edge_groups
        self.depthwise = depthwise
        if depthwise:   # models/tensor_layers.py:248-290: 'uvu' TensorProduct into the sorted mid irreps, then linear_2 (replaces FasterTP)
            from .e3nn_lite import Linear, TensorProduct
            mid, instr = [], []
            for i, a in enumerate(self.in_irreps):
                for j, b in enumerate(self.sh_irreps):
                    for ir_out in a.ir * b.ir:
                        if ir_out in self.out_irreps:
                            instr.append((i, j, len(mid), "uvu", True))
                            mid.append((a.mul, ir_out))
            mid, p, _ = Irreps(mid).sort()
            instr = [(i1, i2, p[io], mode, train) for i1, i2, io, mode, train in instr]
            self.faster = False
            self.tp = TensorProduct(self.in_irreps, self.sh_irreps, mid, instr)
            self.weight_numel = self.tp.weight_numel
            self.linear_2 = Linear(mid.simplify(), self.out_irreps)
            self.linear_2.weight.data = sd[f"{name}.linear_2.weight"]
            self.mid_irreps = mid
        elif faster:
            self.weight_numel = faster_weight_numel(in_irreps, out_irreps)
        else:
            self.tp = FullyConnectedTensorProduct(in_irreps, sh_irreps, out_irreps)
            self.weight_numel = self.tp.weight_numel

    def _tp(self, x, sh, w):
        if self.faster:
            return faster_tensor_product(self.in_irreps, self.out_irreps, x, sh, w)
        return self.tp(x, sh, w)

    def _fc(self, g, edge_attr):
        pre = f"{self.name}.fc" if self.edge_groups == 1 else f"{self.name}.fc.{g}"
        return fc_block(self.sd, pre, edge_attr, self.tp_weights_layers)

    def __call__(self, node_attr, edge_index, edge_attr, edge_sh, out_nodes=None, reduce="mean", edge_weight=1.0):
        if edge_index.shape[1] == 0 and node_attr.shape[0] == 0:
            raise ValueError("No edges and no nodes")
        n_out = out_nodes or node_attr.shape[0]
        tp_dim = self.mid_irreps.dim if self.depthwise else self.out_irreps.dim
        if edge_index.shape[1] == 0:
            out = node_attr.new_zeros(node_attr.shape[0], self.out_irreps.dim)
        else:
            src, dst = edge_index
            if self.edge_groups == 1:
                w = self._fc(0, edge_attr) * edge_weight
                out = scatter(self._tp(node_attr[dst], edge_sh, w), src, 0, n_out, reduce)
            else:
                assert isinstance(edge_attr, list) and len(edge_attr) == self.edge_groups
                out = node_attr.new_zeros(n_out, tp_dim)
                div = node_attr.new_zeros(n_out)
                start = 0
                for g, ea in enumerate(edge_attr):
                    sl = slice(start, start + ea.shape[0])
                    start += ea.shape[0]
                    w = self._fc(g, ea)
                    w = w * (edge_weight[sl] if torch.is_tensor(edge_weight) else edge_weight)
                    out = out + scatter(self._tp(node_attr[dst[sl]], edge_sh[sl], w), src[sl], 0, n_out, "sum")
                    div = div + torch.bincount(src[sl], minlength=n_out).to(div.dtype)
                assert start == edge_index.shape[1]
                if reduce == "mean":
                    out = out / div.clamp(min=torch.finfo(div.dtype).eps)[:, None]
            if self.depthwise:   # tensor_layers.py:324-325
                with torch.no_grad():
                    out = self.linear_2(out)
            if self.batch_norm:
                n = f"{self.name}.batch_norm"
                out = batch_norm_eval(self.out_irreps, out, self.sd[n + ".running_mean"], self.sd[n + ".running_var"],
                                      self.sd[n + ".weight"], self.sd[n + ".bias"])
        if self.residual:
            out = out + F.pad(node_attr, (0, out.shape[-1] - node_attr.shape[-1]))
        return out
