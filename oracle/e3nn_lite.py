"""Oracle restatement of the e3nn==0.5.x pieces the DiffDock hot path calls.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  e3nn is a third-party
dependency of the reference (requirements.txt:7 pins e3nn==0.5.0,
environment.yml:23 pins 0.5.1) and is NOT vendored under /root/reference, so this
file restates its published algorithms.  Reference call sites:

  o3.Irreps / Irreps.spherical_harmonics      models/cg_model.py:47, tensor_layers.py:54-55
  o3.spherical_harmonics(normalize=True,
        normalization='component')            models/cg_model.py:411,494,511,556-557,622,636
  o3.FullyConnectedTensorProduct(
        shared_weights=False)                 models/tensor_layers.py:299
  o3.FullTensorProduct(sh, "2e")              models/cg_model.py:240,412
  e3nn.nn.BatchNorm (eval)                    models/tensor_layers.py:307,327-328

The only in-repo pin of these conventions is FasterTensorProduct
(models/tensor_layers.py:44-122); tests/test_oracle_e3nn.py checks that the generic
FullyConnectedTensorProduct below reproduces it after the weight-slot remap.
"""
from __future__ import annotations

import math
from fractions import Fraction
from functools import lru_cache
from typing import List, Tuple

import torch


# ----------------------------------------------------------------------------- Irreps
class Irrep(tuple):
    """(l, p) with p=+1 ('e') / -1 ('o'); tuple ordering is e3nn's sort order."""

    def __new__(cls, l, p=None):
        if p is None:
            if isinstance(l, Irrep):
                return l
            if isinstance(l, str):
                s = l.strip()
                return super().__new__(cls, (int(s[:-1]), {"e": 1, "o": -1}[s[-1]]))
            l, p = l
        return super().__new__(cls, (int(l), int(p)))

    @property
    def l(self):
        return self[0]

    @property
    def p(self):
        return self[1]

    @property
    def dim(self):
        return 2 * self.l + 1

    def is_scalar(self):
        return self.l == 0 and self.p == 1

    def __mul__(self, other):
        other = Irrep(other)
        return [Irrep(l, self.p * other.p) for l in range(abs(self.l - other.l), self.l + other.l + 1)]

    def __repr__(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"

    __str__ = __repr__


class _MulIr(tuple):
    def __new__(cls, mul, ir):
        return super().__new__(cls, (int(mul), Irrep(ir)))

    @property
    def mul(self):
        return self[0]

    @property
    def ir(self):
        return self[1]

    @property
    def dim(self):
        return self.mul * self.ir.dim


class Irreps(tuple):
    """Minimal o3.Irreps: parse '48x0e + 10x1o', iterate (mul, ir), slices, sort."""

    def __new__(cls, irreps=None):
        if isinstance(irreps, Irreps):
            return irreps
        out = []
        if isinstance(irreps, str):
            for term in irreps.split("+"):
                term = term.strip()
                if not term:
                    continue
                if "x" in term:
                    mul, ir = term.split("x")
                else:
                    mul, ir = 1, term
                out.append(_MulIr(int(mul), Irrep(ir.strip())))
        elif irreps is not None:
            for item in irreps:
                if isinstance(item, _MulIr):
                    out.append(item)
                elif isinstance(item, Irrep):
                    out.append(_MulIr(1, item))
                else:
                    mul, ir = item
                    out.append(_MulIr(mul, Irrep(ir)))
        return super().__new__(cls, out)

    @staticmethod
    def spherical_harmonics(lmax, p=-1):
        return Irreps([(1, (l, p ** l)) for l in range(lmax + 1)])

    @property
    def dim(self):
        return sum(mi.dim for mi in self)

    @property
    def num_irreps(self):
        return sum(mi.mul for mi in self)

    def slices(self):
        out, i = [], 0
        for mi in self:
            out.append(slice(i, i + mi.dim))
            i += mi.dim
        return out

    def sort(self):
        """e3nn Irreps.sort(): stable sort of (ir, original index); returns (irreps, p, inv)."""
        order = sorted((mi.ir, i, mi.mul) for i, mi in enumerate(self))
        inv = tuple(i for _, i, _ in order)
        p = [0] * len(inv)
        for new, old in enumerate(inv):
            p[old] = new
        return Irreps([(mul, ir) for ir, _, mul in order]), tuple(p), inv

    def __contains__(self, ir):
        ir = Irrep(ir)
        return any(mi.ir == ir for mi in self)

    def __eq__(self, other):
        return tuple(self) == tuple(Irreps(other))

    def __hash__(self):
        return hash(tuple(self))

    def __repr__(self):
        return "+".join(f"{mi.mul}x{mi.ir}" for mi in self)


# ------------------------------------------------------------------------- wigner 3j
def _f(n):
    return math.factorial(round(n))


def _su2_cg_coeff(j1, m1, j2, m2, j3, m3):
    """<j1 m1 j2 m2 | j3 m3>, Racah's formula as used by e3nn.o3._wigner."""
    if m3 != m1 + m2:
        return 0.0
    vmin = int(max(-j1 + j2 + m3, -j1 + m1, 0))
    vmax = int(min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3))
    C = ((2.0 * j3 + 1.0) * Fraction(
        _f(j3 + j1 - j2) * _f(j3 - j1 + j2) * _f(j1 + j2 - j3) * _f(j3 + m3) * _f(j3 - m3),
        _f(j1 + j2 + j3 + 1) * _f(j1 - m1) * _f(j1 + m1) * _f(j2 - m2) * _f(j2 + m2))) ** 0.5
    S = 0
    for v in range(vmin, vmax + 1):
        S += (-1) ** int(v + j2 + m2) * Fraction(
            _f(j2 + j3 + m1 - v) * _f(j1 - m1 + v),
            _f(v) * _f(j3 - j1 + j2 - v) * _f(j3 + m3 - v) * _f(v + j1 - j2 - m3))
    return float(C * S)


def _su2_cg(j1, j2, j3):
    mat = torch.zeros(2 * j1 + 1, 2 * j2 + 1, 2 * j3 + 1, dtype=torch.float64)
    if abs(j1 - j2) <= j3 <= j1 + j2:
        for m1 in range(-j1, j1 + 1):
            for m2 in range(-j2, j2 + 1):
                if abs(m1 + m2) <= j3:
                    mat[j1 + m1, j2 + m2, j3 + m1 + m2] = _su2_cg_coeff(j1, m1, j2, m2, j3, m1 + m2)
    return mat


def _real_to_complex(l):
    q = torch.zeros(2 * l + 1, 2 * l + 1, dtype=torch.complex128)
    s = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l + abs(m)] = s
        q[l + m, l - abs(m)] = -1j * s
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + abs(m)] = (-1) ** m * s
        q[l + m, l - abs(m)] = 1j * (-1) ** m * s
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def _w3j_cached(l1, l2, l3):
    Q1, Q2, Q3 = _real_to_complex(l1), _real_to_complex(l2), _real_to_complex(l3)
    C = _su2_cg(l1, l2, l3).to(torch.complex128)
    C = torch.einsum("ij,kl,mn,ikn->jlm", Q1, Q2, torch.conj(Q3.T), C)
    assert torch.all(C.imag.abs() < 1e-9)
    C = C.real
    return C / C.norm()


def wigner_3j(l1, l2, l3, dtype=torch.float64):
    """Real-basis Wigner 3j, Frobenius norm 1 (e3nn.o3.wigner_3j)."""
    assert abs(l2 - l3) <= l1 <= l2 + l3
    return _w3j_cached(int(l1), int(l2), int(l3)).to(dtype).clone()


# --------------------------------------------------------------- spherical harmonics
def spherical_harmonics(irreps, vec, normalize=True, normalization="component"):
    """o3.spherical_harmonics for l<=2 (all the path uses), 'component' normalisation.

    irreps may be an Irreps / str ('2e') / list of l.  Zero vectors stay zero
    (torch.nn.functional.normalize semantics: x / max(|x|, 1e-12)).
    """
    assert normalization == "component"
    if isinstance(irreps, (str, Irreps)):
        ls = [mi.ir.l for mi in Irreps(irreps) for _ in range(mi.mul)]
    else:
        ls = list(irreps)
    if normalize:
        vec = torch.nn.functional.normalize(vec, dim=-1)
    x, y, z = vec[..., 0], vec[..., 1], vec[..., 2]
    out = []
    for l in ls:
        if l == 0:
            out.append(torch.ones_like(x).unsqueeze(-1))
        elif l == 1:
            out.append(math.sqrt(3) * torch.stack([x, y, z], -1))
        elif l == 2:
            s3 = math.sqrt(3)
            out.append(math.sqrt(5) * torch.stack([
                s3 * x * z, s3 * x * y, y * y - 0.5 * (x * x + z * z), s3 * y * z,
                (s3 / 2) * (z * z - x * x)], -1))
        else:
            raise NotImplementedError("oracle SH restated for l<=2 only")
    return torch.cat(out, -1)


# ------------------------------------------------------------------ tensor products
class FullyConnectedTensorProduct(torch.nn.Module):
    """o3.FullyConnectedTensorProduct(in1, in2, out, shared_weights=False).

    Instructions (= weight slots, in this nesting order, shape (mul1, mul2, mul_out)
    row-major): for i1 in in1, for i2 in in2, for io in out if out[io].ir in ir1*ir2.
    irrep_normalization='component', path_normalization='element':
        coeff_io = sqrt((2 l_out + 1) / sum_{slots into io} mul1*mul2)
    out[io][w,k] += coeff * sum_{u,v,i,j} W[u,v,w] x1[u,i] x2[v,j] w3j(l1,l2,lo)[i,j,k]
    """

    def __init__(self, irreps_in1, irreps_in2, irreps_out, shared_weights=False):
        super().__init__()
        assert not shared_weights
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        self.instructions: List[Tuple[int, int, int]] = []
        for i1, a in enumerate(self.irreps_in1):
            for i2, b in enumerate(self.irreps_in2):
                for io, c in enumerate(self.irreps_out):
                    if c.ir in a.ir * b.ir:
                        self.instructions.append((i1, i2, io))
        fan = {}
        for i1, i2, io in self.instructions:
            fan[io] = fan.get(io, 0) + self.irreps_in1[i1].mul * self.irreps_in2[i2].mul
        self.coeffs = [math.sqrt(self.irreps_out[io].ir.dim / fan[io]) for _, _, io in self.instructions]
        self.slot_shapes = [(self.irreps_in1[i1].mul, self.irreps_in2[i2].mul, self.irreps_out[io].mul)
                            for i1, i2, io in self.instructions]
        self.weight_numel = sum(a * b * c for a, b, c in self.slot_shapes)

    def forward(self, x1, x2, weight):
        E = x1.shape[0]
        s1, s2 = self.irreps_in1.slices(), self.irreps_in2.slices()
        outs = [x1.new_zeros(E, mi.mul, mi.ir.dim) for mi in self.irreps_out]
        off = 0
        for (i1, i2, io), c, (m1, m2, mo) in zip(self.instructions, self.coeffs, self.slot_shapes):
            a, b, o = self.irreps_in1[i1], self.irreps_in2[i2], self.irreps_out[io]
            W = weight[:, off:off + m1 * m2 * mo].reshape(E, m1, m2, mo)
            off += m1 * m2 * mo
            X1 = x1[:, s1[i1]].reshape(E, m1, a.ir.dim)
            X2 = x2[:, s2[i2]].reshape(E, m2, b.ir.dim)
            C = wigner_3j(a.ir.l, b.ir.l, o.ir.l, dtype=x1.dtype)
            xx = torch.einsum("eui,evj,ijk->euvk", X1, X2, C)
            outs[io] = outs[io] + c * torch.einsum("euvw,euvk->ewk", W, xx)
        return torch.cat([o.reshape(E, mi.dim) for o, mi in zip(outs, self.irreps_out)], -1)


class FullTensorProduct(torch.nn.Module):
    """o3.FullTensorProduct(in1, in2): weight-free 'uvuv' paths, coefficient
    sqrt(2 l_out + 1) * w3j, output blocks sorted with Irreps.sort()."""

    def __init__(self, irreps_in1, irreps_in2):
        super().__init__()
        self.irreps_in1, self.irreps_in2 = Irreps(irreps_in1), Irreps(irreps_in2)
        out, instr = [], []
        for i1, a in enumerate(self.irreps_in1):
            for i2, b in enumerate(self.irreps_in2):
                for ir in a.ir * b.ir:
                    instr.append((i1, i2, len(out)))
                    out.append((a.mul * b.mul, ir))
        out = Irreps(out)
        self.irreps_out, p, _ = out.sort()
        self.instructions = [(i1, i2, p[io]) for i1, i2, io in instr]

    def forward(self, x1, x2):
        E = x1.shape[0]
        s1, s2, so = self.irreps_in1.slices(), self.irreps_in2.slices(), self.irreps_out.slices()
        out = x1.new_zeros(E, self.irreps_out.dim)
        for i1, i2, io in self.instructions:
            a, b, o = self.irreps_in1[i1], self.irreps_in2[i2], self.irreps_out[io]
            X1 = x1[:, s1[i1]].reshape(E, a.mul, a.ir.dim)
            X2 = x2[:, s2[i2]].reshape(E, b.mul, b.ir.dim)
            C = wigner_3j(a.ir.l, b.ir.l, o.ir.l, dtype=x1.dtype) * math.sqrt(o.ir.dim)
            out[:, so[io]] = torch.einsum("eui,evj,ijk->euvk", X1, X2, C).reshape(E, o.dim)
        return out


class BatchNorm(torch.nn.Module):
    """e3nn.nn.BatchNorm(irreps), eval-mode arithmetic (eps=1e-5, affine, 'component').

    running_mean / bias: one per 0e channel; running_var / weight: one per channel."""

    def __init__(self, irreps, eps=1e-5):
        super().__init__()
        self.irreps = Irreps(irreps)
        self.eps = eps
        ns = sum(mi.mul for mi in self.irreps if mi.ir.is_scalar())
        nf = self.irreps.num_irreps
        self.register_buffer("running_mean", torch.zeros(ns))
        self.register_buffer("running_var", torch.ones(nf))
        self.weight = torch.nn.Parameter(torch.ones(nf))
        self.bias = torch.nn.Parameter(torch.zeros(ns))

    def forward(self, x):
        return batch_norm_eval(self.irreps, x, self.running_mean, self.running_var, self.weight, self.bias, self.eps)


def batch_norm_eval(irreps, x, running_mean, running_var, weight, bias, eps=1e-5):
    irreps = Irreps(irreps)
    N = x.shape[0]
    out, ix, iv, im = [], 0, 0, 0
    for mi in irreps:
        d = mi.ir.dim
        f = x[:, ix:ix + mi.mul * d].reshape(N, mi.mul, d)
        ix += mi.mul * d
        if mi.ir.is_scalar():
            f = f - running_mean[im:im + mi.mul].reshape(1, -1, 1)
        scale = (running_var[iv:iv + mi.mul] + eps).pow(-0.5) * weight[iv:iv + mi.mul]
        f = f * scale.reshape(1, -1, 1)
        if mi.ir.is_scalar():
            f = f + bias[im:im + mi.mul].reshape(1, -1, 1)
            im += mi.mul
        iv += mi.mul
        out.append(f.reshape(N, mi.mul * d))
    return torch.cat(out, -1)


# ------------------------------------------------------------------ o3.Linear / o3.TensorProduct('uvu')
def simplify(irreps):
    """Irreps.simplify(): adjacent entries of the same irrep merged (no sort)."""
    out = []
    for mi in Irreps(irreps):
        if out and out[-1][1] == mi.ir:
            out[-1] = (out[-1][0] + mi.mul, mi.ir)
        elif mi.mul > 0:
            out.append((mi.mul, mi.ir))
    return Irreps(out)


Irreps.simplify = lambda self: simplify(self)


class Linear(torch.nn.Module):
    """o3.Linear(irreps_in, irreps_out, internal_weights=True, shared_weights=True), no biases (e3nn/o3/_linear.py).

    Reference call sites: CGModel.sidechain_predictor (models/cg_model.py:173-178,401) and the second stage of a depthwise
    TensorProductConvLayer (models/tensor_layers.py:281-290,324-325).
    Instructions (weight slots, [mul_in, mul_out] row-major, in this order): for i_in in irreps_in, for i_out in irreps_out if the
    irreps agree.  path_normalization='element': every slot into output block i_out is scaled by
    (sum of mul_in over the slots into i_out) ** -0.5.  out[i_out][w, m] += scale * sum_u W[u, w] x[i_in][u, m]."""

    def __init__(self, irreps_in, irreps_out, internal_weights=True, shared_weights=True, **kw):
        super().__init__()
        assert internal_weights and shared_weights and not kw.get("biases", False)
        self.irreps_in, self.irreps_out = Irreps(irreps_in), Irreps(irreps_out)
        self.instructions = [(i, o) for i, a in enumerate(self.irreps_in) for o, b in enumerate(self.irreps_out) if a.ir == b.ir]
        fan = {}
        for i, o in self.instructions:
            fan[o] = fan.get(o, 0) + self.irreps_in[i].mul
        self.scales = [fan[o] ** -0.5 for _, o in self.instructions]
        self.weight_numel = sum(self.irreps_in[i].mul * self.irreps_out[o].mul for i, o in self.instructions)
        self.weight = torch.nn.Parameter(torch.randn(self.weight_numel))

    def forward(self, x):
        N = x.shape[0]
        si = self.irreps_in.slices()
        outs = [x.new_zeros(N, mi.mul, mi.ir.dim) for mi in self.irreps_out]
        off = 0
        for (i, o), sc in zip(self.instructions, self.scales):
            a, b = self.irreps_in[i], self.irreps_out[o]
            W = self.weight[off:off + a.mul * b.mul].reshape(a.mul, b.mul).to(x.dtype)
            off += a.mul * b.mul
            outs[o] = outs[o] + sc * torch.einsum("uw,num->nwm", W, x[:, si[i]].reshape(N, a.mul, a.ir.dim))
        return torch.cat([t.reshape(N, mi.dim) for t, mi in zip(outs, self.irreps_out)], -1)


class TensorProduct(torch.nn.Module):
    """o3.TensorProduct(in1, in2, out, instructions, shared_weights=False, internal_weights=False) for the 'uvu' connection mode
    with trainable paths -- what a depthwise TensorProductConvLayer builds (models/tensor_layers.py:248-279): one weight per
    (path, u) [mul_in2 is 1 for spherical harmonics], weight slots in instruction order, [mul1, mul2] row-major.
    irrep_normalization='component', path_normalization='element' (e3nn/o3/_tensor_product/_tensor_product.py):
        coeff = sqrt((2 l_out + 1) / sum over the instructions into the same output block of mul2)
    out[i_out][u, k] += coeff * sum_{v, i, j} W[u, v] x1[u, i] x2[v, j] w3j(l1, l2, l_out)[i, j, k]"""

    def __init__(self, irreps_in1, irreps_in2, irreps_out, instructions, shared_weights=False, internal_weights=False):
        super().__init__()
        assert not shared_weights and not internal_weights
        self.irreps_in1, self.irreps_in2, self.irreps_out = Irreps(irreps_in1), Irreps(irreps_in2), Irreps(irreps_out)
        self.instructions = []
        for ins in instructions:
            i1, i2, io, mode, train = ins[:5]
            assert mode == "uvu" and train
            assert self.irreps_in1[i1].mul == self.irreps_out[io].mul
            self.instructions.append((i1, i2, io))
        fan = {}
        for i1, i2, io in self.instructions:
            fan[io] = fan.get(io, 0) + self.irreps_in2[i2].mul
        self.coeffs = [math.sqrt(self.irreps_out[io].ir.dim / fan[io]) for _, _, io in self.instructions]
        self.slot_shapes = [(self.irreps_in1[i1].mul, self.irreps_in2[i2].mul) for i1, i2, _ in self.instructions]
        self.weight_numel = sum(a * b for a, b in self.slot_shapes)

    def forward(self, x1, x2, weight):
        E = x1.shape[0]
        s1, s2 = self.irreps_in1.slices(), self.irreps_in2.slices()
        outs = [x1.new_zeros(E, mi.mul, mi.ir.dim) for mi in self.irreps_out]
        off = 0
        for (i1, i2, io), c, (m1, m2) in zip(self.instructions, self.coeffs, self.slot_shapes):
            a, b, o = self.irreps_in1[i1], self.irreps_in2[i2], self.irreps_out[io]
            W = weight[:, off:off + m1 * m2].reshape(E, m1, m2)
            off += m1 * m2
            X1 = x1[:, s1[i1]].reshape(E, m1, a.ir.dim)
            X2 = x2[:, s2[i2]].reshape(E, m2, b.ir.dim)
            C = wigner_3j(a.ir.l, b.ir.l, o.ir.l, dtype=x1.dtype)
            outs[io] = outs[io] + c * torch.einsum("euv,eui,evj,ijk->euk", W, X1, X2, C)
        return torch.cat([t.reshape(E, mi.dim) for t, mi in zip(outs, self.irreps_out)], -1)
