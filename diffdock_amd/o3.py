"""Host-side O(3) constants for weight pre-packing: real-basis Wigner-3j tensors and the
dense coupling tables the HIP kernels stage in LDS.

The reference gets these from e3nn (third-party, e3nn==0.5.x `o3.wigner_3j`, used inside
FullyConnectedTensorProduct / FullTensorProduct, models/tensor_layers.py:299,
models/cg_model.py:240).  Algorithm (e3nn `_so3_clebsch_gordan`): SU(2) Clebsch-Gordan
coefficients, conjugated into the real spherical-harmonic basis (y polar axis, l=1 order
x,y,z), real part, Frobenius-normalised.  Computed here in float64 with numpy.
"""
from __future__ import annotations

import math
from functools import lru_cache

import numpy as np

from .irreps import FASTER_TERMS, FASTER_TYPES, faster_weight_shapes, fctp_paths, parse_irreps


def _fact(n: int) -> int:
    return math.factorial(int(round(n)))


def _cg(j1, m1, j2, m2, j3, m3) -> float:
    """<j1 m1; j2 m2 | j3 m3> (Condon-Shortley), integer arithmetic under the root."""
    if m1 + m2 != m3 or not (abs(j1 - j2) <= j3 <= j1 + j2):
        return 0.0
    pref_num = (2 * j3 + 1) * _fact(j3 + j1 - j2) * _fact(j3 - j1 + j2) * _fact(j1 + j2 - j3) * _fact(j3 + m3) * _fact(j3 - m3)
    pref_den = _fact(j1 + j2 + j3 + 1) * _fact(j1 - m1) * _fact(j1 + m1) * _fact(j2 - m2) * _fact(j2 + m2)
    s = 0.0
    lo = max(-j1 + j2 + m3, -j1 + m1, 0)
    hi = min(j2 + j3 + m1, j3 - j1 + j2, j3 + m3)
    for v in range(lo, hi + 1):
        num = _fact(j2 + j3 + m1 - v) * _fact(j1 - m1 + v)
        den = _fact(v) * _fact(j3 - j1 + j2 - v) * _fact(j3 + m3 - v) * _fact(v + j1 - j2 - m3)
        s += (-1) ** (v + j2 + m2) * num / den
    return math.sqrt(pref_num / pref_den) * s


def _q(l: int) -> np.ndarray:
    """real -> complex change of basis (rows m=-l..l), including the (-i)^l phase."""
    q = np.zeros((2 * l + 1, 2 * l + 1), dtype=np.complex128)
    r = 1 / math.sqrt(2)
    for m in range(-l, 0):
        q[l + m, l - m] = r
        q[l + m, l + m] = -1j * r
    q[l, l] = 1
    for m in range(1, l + 1):
        q[l + m, l + m] = (-1) ** m * r
        q[l + m, l - m] = 1j * (-1) ** m * r
    return (-1j) ** l * q


@lru_cache(maxsize=None)
def wigner_3j(l1: int, l2: int, l3: int) -> np.ndarray:
    c = np.zeros((2 * l1 + 1, 2 * l2 + 1, 2 * l3 + 1))
    for m1 in range(-l1, l1 + 1):
        for m2 in range(-l2, l2 + 1):
            if abs(m1 + m2) <= l3:
                c[l1 + m1, l2 + m2, l3 + m1 + m2] = _cg(l1, m1, l2, m2, l3, m1 + m2)
    t = np.einsum("ij,kl,mn,ikn->jlm", _q(l1), _q(l2), np.conj(_q(l3).T), c.astype(np.complex128))
    assert np.abs(t.imag).max() < 1e-9
    t = t.real
    return t / np.linalg.norm(t)


# ------------------------------------------------------------------------------------
# Unified path table.  Every tensor product on the path (FasterTensorProduct and e3nn
# FullyConnectedTensorProduct) is a list of paths
#     out[o_off + w*do + k] += sum_{u,i,j} Wt[w_off + u*mulo + w] * x[i_off + u*di + i] * sh[s_off + j] * C[i,j,k]
# with a dense coupling tensor C (coefficient folded in).  `Wt` = per-edge weight row.
class TPPath:
    __slots__ = ("i_off", "mul_in", "di", "s_off", "ds", "o_off", "mul_out", "do", "w_off", "C")

    def __init__(self, i_off, mul_in, di, s_off, ds, o_off, mul_out, do, w_off, C):
        self.i_off, self.mul_in, self.di = i_off, mul_in, di
        self.s_off, self.ds = s_off, ds
        self.o_off, self.mul_out, self.do = o_off, mul_out, do
        self.w_off, self.C = w_off, np.asarray(C, dtype=np.float64)


def fctp_path_table(in_irreps: str, sh_irreps: str, out_irreps: str):
    A, S, O = parse_irreps(in_irreps), parse_irreps(sh_irreps), parse_irreps(out_irreps)
    paths, numel = fctp_paths(in_irreps, sh_irreps, out_irreps)
    table = []
    for p in paths:
        a, s, o = A[p.i1], S[p.i2], O[p.io]
        assert s.mul == 1, "spherical-harmonic operands have multiplicity 1 on this path"
        C = p.coeff * wigner_3j(a.l, s.l, o.l)
        table.append(TPPath(a.offset, a.mul, 2 * a.l + 1, s.offset, 2 * s.l + 1, o.offset, o.mul, 2 * o.l + 1, p.w_off, C))
    return table, numel


def faster_path_table(in_irreps: str, out_irreps: str):
    """FasterTensorProduct (models/tensor_layers.py:71-122) as a path table: per output
    type a [fan_in, mul_out] matrix / sqrt(fan_in); terms use the hard-coded l<=1 couplings."""
    shapes, numel, im, om = faster_weight_shapes(in_irreps, out_irreps)
    blocks_in = {b.name: b for b in parse_irreps(in_irreps)}
    blocks_out = {b.name: b for b in parse_irreps(out_irreps)}
    eye3 = np.eye(3)
    eps = np.zeros((3, 3, 3))
    eps[0, 1, 2] = eps[1, 2, 0] = eps[2, 0, 1] = 1
    eps[0, 2, 1] = eps[2, 1, 0] = eps[1, 0, 2] = -1

    def coupling(tin, tout):
        li, lo = int(tin[0]), int(tout[0])
        if li == 0 and lo == 0:      # scalar * s0
            return 0, np.ones((1, 1, 1))
        if li == 1 and lo == 0:      # (v . s1)/sqrt3
            return 1, (eye3 / math.sqrt(3)).reshape(3, 3, 1)
        if li == 0 and lo == 1:      # scalar (x) s1
            return 1, eye3.reshape(1, 3, 3)
        if tin == tout:              # v * s0
            return 0, eye3.reshape(3, 1, 3)
        return 1, eps / math.sqrt(2)  # (v x s1)/sqrt2 : out_k = eps_{ijk} v_i s_j

    table, w_off = [], 0
    for t in FASTER_TYPES:
        fan, mo = shapes[t]
        row = 0
        for s in FASTER_TERMS[t]:
            if im[s] == 0:
                continue
            if mo > 0:
                ls, C = coupling(s, t)
                a, o = blocks_in[s], blocks_out[t]
                table.append(TPPath(a.offset, a.mul, 2 * a.l + 1, 0 if ls == 0 else 1, 1 if ls == 0 else 3,
                                    o.offset, o.mul, 2 * o.l + 1, w_off + row * mo, C / math.sqrt(fan)))
            row += im[s]
        w_off += fan * mo
    assert w_off == numel
    return table, numel


def full_tp_table(in1: str, in2: str):
    """Dense [dim1, dim2, dim_out] tensor of o3.FullTensorProduct(in1, in2) for multiplicity-1
    operands (sh (x) '2e', models/cg_model.py:240,412): coefficient sqrt(2l+1)*w3j, output
    blocks sorted by (l, p)."""
    A, Bk = parse_irreps(in1), parse_irreps(in2)
    blocks = []
    for a in A:
        for b in Bk:
            assert a.mul == 1 and b.mul == 1
            for l in range(abs(a.l - b.l), a.l + b.l + 1):
                blocks.append((a, b, l, a.p * b.p))
    order = sorted(range(len(blocks)), key=lambda i: ((blocks[i][2], blocks[i][3]), i))
    d1, d2 = sum(a.dim for a in A), sum(b.dim for b in Bk)
    dout = sum(2 * blocks[i][2] + 1 for i in order)
    T = np.zeros((d1, d2, dout))
    off = 0
    for i in order:
        a, b, l, _ = blocks[i]
        T[a.offset:a.offset + a.dim, b.offset:b.offset + b.dim, off:off + 2 * l + 1] = math.sqrt(2 * l + 1) * wigner_3j(a.l, b.l, l)
        off += 2 * l + 1
    return T
