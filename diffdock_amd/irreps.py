"""Irreps layout and tensor-product slot tables used to size and pre-pack weights.

Feature rows are the concatenation of blocks `mul x (l,p)`, each block mul-major
`[u, m]` (reference models/tensor_layers.py:73-75,107,112).  Two weight layouts exist
for the per-edge tensor-product weights `[E, weight_numel]`:

  * FasterTensorProduct (models/tensor_layers.py:63-69,92-98): per OUTPUT type in the
    fixed order 0e,1o,1e,0o one `[fan_in, mul_out]` matrix, fan_in rows in the order the
    contributing terms are appended at :77-90.
  * e3nn FullyConnectedTensorProduct: one `(mul1, mul2, mul_out)` slot per instruction,
    instructions enumerated `for i1 in in1: for i2 in in2: for io in out` (e3nn 0.5).
"""
from __future__ import annotations

import math
from typing import List, NamedTuple, Tuple


class Block(NamedTuple):
    mul: int
    l: int
    p: int  # +1 even, -1 odd
    offset: int

    @property
    def dim(self):
        return self.mul * (2 * self.l + 1)

    @property
    def name(self):
        return f"{self.l}{'e' if self.p == 1 else 'o'}"


def parse_irreps(s) -> List[Block]:
    if not isinstance(s, str):
        return list(s)
    out, off = [], 0
    for term in s.split("+"):
        term = term.strip()
        if not term:
            continue
        mul, ir = term.split("x") if "x" in term else ("1", term)
        l, p = int(ir.strip()[:-1]), (1 if ir.strip()[-1] == "e" else -1)
        out.append(Block(int(mul), l, p, off))
        off += int(mul) * (2 * l + 1)
    return out


def irreps_dim(s) -> int:
    return sum(b.dim for b in parse_irreps(s))


def irreps_num(s) -> int:
    return sum(b.mul for b in parse_irreps(s))


def sh_irreps(lmax) -> str:
    return " + ".join(f"1x{l}{'e' if l % 2 == 0 else 'o'}" for l in range(lmax + 1))


def _sort_key(b: Block):
    return (b.l, b.p)


def full_tp_irreps(in1: str, in2: str) -> str:
    """irreps_out of o3.FullTensorProduct(in1, in2): one block per (i1, i2, l_out),
    then stable-sorted by (l, p)."""
    blocks = []
    for a in parse_irreps(in1):
        for b in parse_irreps(in2):
            for l in range(abs(a.l - b.l), a.l + b.l + 1):
                blocks.append((a.mul * b.mul, l, a.p * b.p))
    order = sorted(range(len(blocks)), key=lambda i: ((blocks[i][1], blocks[i][2]), i))
    return " + ".join(f"{blocks[i][0]}x{blocks[i][1]}{'e' if blocks[i][2] == 1 else 'o'}" for i in order)


class Path(NamedTuple):
    i1: int
    i2: int
    io: int
    mul1: int
    mul2: int
    mulo: int
    l1: int
    l2: int
    lo: int
    w_off: int     # offset of this slot in the weight row
    coeff: float   # sqrt((2lo+1)/fan_in(io))


def fctp_paths(in1: str, in2: str, out: str) -> Tuple[List[Path], int]:
    A, B, C = parse_irreps(in1), parse_irreps(in2), parse_irreps(out)
    raw = []
    for i1, a in enumerate(A):
        for i2, b in enumerate(B):
            for io, c in enumerate(C):
                if c.p == a.p * b.p and abs(a.l - b.l) <= c.l <= a.l + b.l:
                    raw.append((i1, i2, io))
    fan = {}
    for i1, i2, io in raw:
        fan[io] = fan.get(io, 0) + A[i1].mul * B[i2].mul
    paths, off = [], 0
    for i1, i2, io in raw:
        a, b, c = A[i1], B[i2], C[io]
        paths.append(Path(i1, i2, io, a.mul, b.mul, c.mul, a.l, b.l, c.l, off,
                          math.sqrt((2 * c.l + 1) / fan[io])))
        off += a.mul * b.mul * c.mul
    return paths, off


FASTER_TYPES = ("0e", "1o", "1e", "0o")
# contributing input types per output type, in append order (tensor_layers.py:77-90)
FASTER_TERMS = {
    "0e": ("0e", "1o"),        # 0e*s0 ; (1o . s1)/sqrt3
    "1o": ("0e", "1o", "1e"),  # 0e (x) s1 ; 1o*s0 ; (1e x s1)/sqrt2
    "1e": ("1o", "1e", "0o"),  # (1o x s1)/sqrt2 ; 1e*s0 ; 0o (x) s1
    "0o": ("1e", "0o"),        # (1e . s1)/sqrt3 ; 0o*s0
}


def faster_weight_shapes(in_irreps: str, out_irreps: str):
    """{type: (fan_in, mul_out)} and weight_numel of FasterTensorProduct."""
    im = {t: 0 for t in FASTER_TYPES}
    om = {t: 0 for t in FASTER_TYPES}
    for b in parse_irreps(in_irreps):
        im[b.name] = b.mul
    for b in parse_irreps(out_irreps):
        om[b.name] = b.mul
    shapes = {t: (sum(im[s] for s in FASTER_TERMS[t]), om[t]) for t in FASTER_TYPES}
    return shapes, sum(a * b for a, b in shapes.values()), im, om


def tp_weight_numel(in_irreps, sh, out_irreps, faster) -> int:
    if faster:
        return faster_weight_shapes(in_irreps, out_irreps)[1]
    return fctp_paths(in_irreps, sh, out_irreps)[1]


def depthwise_numels(in_irreps: str, sh: str, out_irreps: str) -> Tuple[int, int]:
    """(weight_numel of the 'uvu' TensorProduct, weight_numel of linear_2) of a depthwise TensorProductConvLayer
    (models/tensor_layers.py:248-290): one weight per (instruction, u); linear_2 = o3.Linear(irreps_mid.sort().simplify(), out)
    = one [sum of mul over the instructions into an irrep, mul_out] slot per output irrep that is reached."""
    A, B, C = parse_irreps(in_irreps), parse_irreps(sh), parse_irreps(out_irreps)
    rows = {}
    n_tp = 0
    for a in A:
        for b in B:
            for l in range(abs(a.l - b.l), a.l + b.l + 1):
                key = (l, a.p * b.p)
                if any((c.l, c.p) == key for c in C):
                    rows[key] = rows.get(key, 0) + a.mul
                    n_tp += a.mul * b.mul
    n_lin = sum(r * c.mul for key, r in rows.items() for c in C if (c.l, c.p) == key)
    return n_tp, n_lin
