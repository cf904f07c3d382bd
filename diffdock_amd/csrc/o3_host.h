// Host-side O(3) bookkeeping for weight pre-packing: irreps layout, real-basis Wigner-3j
// (the e3nn 0.5 algorithm: SU(2) Clebsch-Gordan via Racah's formula, conjugated into the
// real spherical-harmonic basis, Frobenius-normalised) and the unified tensor-product
// "path table" every kernel consumes.
//
// Reference: FasterTensorProduct models/tensor_layers.py:44-122; e3nn
// FullyConnectedTensorProduct / FullTensorProduct call sites models/tensor_layers.py:299,
// models/cg_model.py:240 (e3nn itself is a third-party dependency, not in the reference tree).
#pragma once
#include <cmath>
#include <complex>
#include <map>
#include <string>
#include <tuple>
#include <vector>

namespace ddmi {

struct IrBlock {
  int mul, l, p, off;  // p = +1 even, -1 odd
  int d() const { return 2 * l + 1; }
  int dim() const { return mul * d(); }
};
typedef std::vector<IrBlock> Irreps;

inline Irreps make_irreps(std::initializer_list<std::tuple<int, int, int>> items) {
  Irreps r;
  int off = 0;
  for (auto& t : items) {
    if (std::get<0>(t) <= 0) continue;
    r.push_back({std::get<0>(t), std::get<1>(t), std::get<2>(t), off});
    off += std::get<0>(t) * (2 * std::get<1>(t) + 1);
  }
  return r;
}
inline int irreps_dim(const Irreps& r) { int s = 0; for (auto& b : r) s += b.dim(); return s; }
inline int irreps_num(const Irreps& r) { int s = 0; for (auto& b : r) s += b.mul; return s; }
inline Irreps sh_irreps(int lmax) {
  Irreps r;
  int off = 0;
  for (int l = 0; l <= lmax; ++l) { r.push_back({1, l, (l % 2) ? -1 : 1, off}); off += 2 * l + 1; }
  return r;
}

std::vector<double> wigner_3j(int l1, int l2, int l3);  // [(2l1+1),(2l2+1),(2l3+1)] row-major

// One path of a tensor product, with the coupling tensor C (normalisation folded in):
//   out[o_off + w*dout + k] += sum_{u,i,j} W[w_off + u*mul_out + w] * x[i_off + u*din + i] * sh[s_off + j] * C[i][j][k]
struct TPPath {
  int i_off, mul_in, din, s_off, ds, o_off, mul_out, dout, w_off;
  int out_block;           // index of the output irreps block
  std::vector<double> C;   // [din][ds][dout]
};
struct TPTable {
  std::vector<TPPath> paths;
  int weight_numel = 0, in_dim = 0, sh_dim = 0, out_dim = 0;
  Irreps out_irreps;
};

TPTable fctp_table(const Irreps& in, const Irreps& sh, const Irreps& out);    // e3nn FullyConnectedTensorProduct
TPTable faster_table(const Irreps& in, const Irreps& out);                    // FasterTensorProduct (sh = 0e+1o)
// o3.FullTensorProduct(sh1, sh2) for multiplicity-1 operands: dense [d1][d2][dout], irreps_out sorted by (l,p).
std::vector<double> full_tp_dense(const Irreps& a, const Irreps& b, Irreps* out_irreps);

}  // namespace ddmi
