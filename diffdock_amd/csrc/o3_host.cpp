#include "o3_host.h"

#include <algorithm>
#include <stdexcept>

namespace ddmi {
namespace {

double fact(int n) {
  double r = 1;
  for (int i = 2; i <= n; ++i) r *= i;
  return r;
}

// <j1 m1; j2 m2 | j3 m3>
double su2_cg(int j1, int m1, int j2, int m2, int j3, int m3) {
  if (m1 + m2 != m3 || j3 < std::abs(j1 - j2) || j3 > j1 + j2) return 0.0;
  double pref = (2.0 * j3 + 1.0) * fact(j3 + j1 - j2) * fact(j3 - j1 + j2) * fact(j1 + j2 - j3) * fact(j3 + m3) *
                fact(j3 - m3) / (fact(j1 + j2 + j3 + 1) * fact(j1 - m1) * fact(j1 + m1) * fact(j2 - m2) * fact(j2 + m2));
  int lo = std::max(std::max(-j1 + j2 + m3, -j1 + m1), 0);
  int hi = std::min(std::min(j2 + j3 + m1, j3 - j1 + j2), j3 + m3);
  double s = 0;
  for (int v = lo; v <= hi; ++v) {
    double term = fact(j2 + j3 + m1 - v) * fact(j1 - m1 + v) /
                  (fact(v) * fact(j3 - j1 + j2 - v) * fact(j3 + m3 - v) * fact(v + j1 - j2 - m3));
    s += (((v + j2 + m2) % 2 + 2) % 2 ? -1.0 : 1.0) * term;
  }
  return std::sqrt(pref) * s;
}

typedef std::complex<double> cd;

// real -> complex change of basis, rows m = -l..l, including the (-i)^l phase
std::vector<cd> q_matrix(int l) {
  int n = 2 * l + 1;
  std::vector<cd> q(n * n, cd(0, 0));
  const double r = 1.0 / std::sqrt(2.0);
  for (int m = -l; m < 0; ++m) {
    q[(l + m) * n + (l - m)] = cd(r, 0);
    q[(l + m) * n + (l + m)] = cd(0, -r);
  }
  q[l * n + l] = cd(1, 0);
  for (int m = 1; m <= l; ++m) {
    double s = (m % 2) ? -1.0 : 1.0;
    q[(l + m) * n + (l + m)] = cd(s * r, 0);
    q[(l + m) * n + (l - m)] = cd(0, s * r);
  }
  cd ph(1, 0);
  for (int i = 0; i < l; ++i) ph *= cd(0, -1);
  for (auto& x : q) x *= ph;
  return q;
}

}  // namespace

std::vector<double> wigner_3j(int l1, int l2, int l3) {
  int n1 = 2 * l1 + 1, n2 = 2 * l2 + 1, n3 = 2 * l3 + 1;
  std::vector<double> c(n1 * n2 * n3, 0.0);
  for (int m1 = -l1; m1 <= l1; ++m1)
    for (int m2 = -l2; m2 <= l2; ++m2)
      if (std::abs(m1 + m2) <= l3) c[((l1 + m1) * n2 + (l2 + m2)) * n3 + (l3 + m1 + m2)] = su2_cg(l1, m1, l2, m2, l3, m1 + m2);
  auto Q1 = q_matrix(l1), Q2 = q_matrix(l2), Q3 = q_matrix(l3);
  // out[j,l,m] = sum_{i,k,n} Q1[i,j] Q2[k,l] conj(Q3^T)[m,n] C[i,k,n],  conj(Q3^T)[m,n] = conj(Q3[n,m])
  std::vector<double> out(n1 * n2 * n3, 0.0);
  double norm = 0;
  for (int j = 0; j < n1; ++j)
    for (int l = 0; l < n2; ++l)
      for (int m = 0; m < n3; ++m) {
        cd acc(0, 0);
        for (int i = 0; i < n1; ++i)
          for (int k = 0; k < n2; ++k)
            for (int n = 0; n < n3; ++n) {
              double cv = c[(i * n2 + k) * n3 + n];
              if (cv != 0.0) acc += Q1[i * n1 + j] * Q2[k * n2 + l] * std::conj(Q3[n * n3 + m]) * cv;
            }
        if (std::abs(acc.imag()) > 1e-9) throw std::runtime_error("wigner_3j: non-real coupling");
        out[(j * n2 + l) * n3 + m] = acc.real();
        norm += acc.real() * acc.real();
      }
  norm = std::sqrt(norm);
  for (auto& x : out) x /= norm;
  return out;
}

TPTable fctp_table(const Irreps& in, const Irreps& sh, const Irreps& out) {
  TPTable t;
  t.in_dim = irreps_dim(in);
  t.sh_dim = irreps_dim(sh);
  t.out_dim = irreps_dim(out);
  t.out_irreps = out;
  struct Raw { int i1, i2, io; };
  std::vector<Raw> raw;
  std::map<int, int> fan;
  for (int i1 = 0; i1 < (int)in.size(); ++i1)
    for (int i2 = 0; i2 < (int)sh.size(); ++i2)
      for (int io = 0; io < (int)out.size(); ++io) {
        const IrBlock &a = in[i1], &b = sh[i2], &c = out[io];
        if (c.p == a.p * b.p && c.l >= std::abs(a.l - b.l) && c.l <= a.l + b.l) {
          raw.push_back({i1, i2, io});
          fan[io] += a.mul * b.mul;
        }
      }
  int off = 0;
  for (auto& r : raw) {
    const IrBlock &a = in[r.i1], &b = sh[r.i2], &c = out[r.io];
    if (b.mul != 1) throw std::runtime_error("fctp_table: second operand must have multiplicity 1");
    TPPath p;
    p.i_off = a.off; p.mul_in = a.mul; p.din = a.d();
    p.s_off = b.off; p.ds = b.d();
    p.o_off = c.off; p.mul_out = c.mul; p.dout = c.d();
    p.w_off = off; p.out_block = r.io;
    double coeff = std::sqrt((double)c.d() / (double)fan[r.io]);
    p.C = wigner_3j(a.l, b.l, c.l);
    for (auto& x : p.C) x *= coeff;
    off += a.mul * b.mul * c.mul;
    t.paths.push_back(std::move(p));
  }
  t.weight_numel = off;
  return t;
}

TPTable faster_table(const Irreps& in, const Irreps& out) {
  // type index: 0 = 0e, 1 = 1o, 2 = 1e, 3 = 0o
  auto type_of = [](const IrBlock& b) { return b.l == 0 ? (b.p == 1 ? 0 : 3) : (b.p == -1 ? 1 : 2); };
  const IrBlock* bin[4] = {nullptr, nullptr, nullptr, nullptr};
  const IrBlock* bout[4] = {nullptr, nullptr, nullptr, nullptr};
  int oblock[4] = {-1, -1, -1, -1};
  for (auto& b : in) { if (b.l > 1) throw std::runtime_error("faster_table: l>1 input"); bin[type_of(b)] = &b; }
  for (int i = 0; i < (int)out.size(); ++i) { if (out[i].l > 1) throw std::runtime_error("faster_table: l>1 output"); bout[type_of(out[i])] = &out[i]; oblock[type_of(out[i])] = i; }
  static const int terms[4][3] = {{0, 1, -1}, {0, 1, 2}, {1, 2, 3}, {2, 3, -1}};  // tensor_layers.py:77-90 append order
  TPTable t;
  t.in_dim = irreps_dim(in);
  t.sh_dim = 4;
  t.out_dim = irreps_dim(out);
  t.out_irreps = out;
  int w_off = 0;
  for (int ot = 0; ot < 4; ++ot) {
    int fan = 0;
    for (int k = 0; k < 3; ++k) if (terms[ot][k] >= 0 && bin[terms[ot][k]]) fan += bin[terms[ot][k]]->mul;
    int mo = bout[ot] ? bout[ot]->mul : 0;
    int row = 0;
    for (int k = 0; k < 3; ++k) {
      int it = terms[ot][k];
      if (it < 0 || !bin[it]) continue;
      if (mo > 0) {
        const IrBlock &a = *bin[it], &c = *bout[ot];
        TPPath p;
        p.i_off = a.off; p.mul_in = a.mul; p.din = a.d();
        p.o_off = c.off; p.mul_out = c.mul; p.dout = c.d();
        p.w_off = w_off + row * mo; p.out_block = oblock[ot];
        const double s = 1.0 / std::sqrt((double)fan);
        if (a.l == 0 && c.l == 0) { p.s_off = 0; p.ds = 1; p.C = {s}; }
        else if (a.l == 1 && c.l == 0) { p.s_off = 1; p.ds = 3; p.C.assign(9, 0.0); for (int i = 0; i < 3; ++i) p.C[i * 3 + i] = s / std::sqrt(3.0); }
        else if (a.l == 0 && c.l == 1) { p.s_off = 1; p.ds = 3; p.C.assign(9, 0.0); for (int j = 0; j < 3; ++j) p.C[j * 3 + j] = s; }
        else if (it == ot) { p.s_off = 0; p.ds = 1; p.C.assign(9, 0.0); for (int i = 0; i < 3; ++i) p.C[i * 3 + i] = s; }
        else {  // cross product / sqrt2 : out_k = eps_{ijk} v_i s_j
          p.s_off = 1; p.ds = 3; p.C.assign(27, 0.0);
          const double e = s / std::sqrt(2.0);
          auto at = [&](int i, int j, int k2) -> double& { return p.C[(i * 3 + j) * 3 + k2]; };
          at(0, 1, 2) = at(1, 2, 0) = at(2, 0, 1) = e;
          at(0, 2, 1) = at(2, 1, 0) = at(1, 0, 2) = -e;
        }
        t.paths.push_back(std::move(p));
      }
      row += bin[it]->mul;
    }
    w_off += fan * mo;
  }
  t.weight_numel = w_off;
  return t;
}

std::vector<double> full_tp_dense(const Irreps& a, const Irreps& b, Irreps* out_irreps) {
  struct Blk { int ia, ib, l, p, idx; };
  std::vector<Blk> blocks;
  for (int ia = 0; ia < (int)a.size(); ++ia)
    for (int ib = 0; ib < (int)b.size(); ++ib) {
      if (a[ia].mul != 1 || b[ib].mul != 1) throw std::runtime_error("full_tp_dense: multiplicity-1 operands only");
      for (int l = std::abs(a[ia].l - b[ib].l); l <= a[ia].l + b[ib].l; ++l)
        blocks.push_back({ia, ib, l, a[ia].p * b[ib].p, (int)blocks.size()});
    }
  std::stable_sort(blocks.begin(), blocks.end(), [](const Blk& x, const Blk& y) {
    return std::make_pair(x.l, x.p) < std::make_pair(y.l, y.p);
  });
  int d1 = irreps_dim(a), d2 = irreps_dim(b), dout = 0;
  for (auto& k : blocks) dout += 2 * k.l + 1;
  std::vector<double> T((size_t)d1 * d2 * dout, 0.0);
  Irreps oi;
  int off = 0;
  for (auto& k : blocks) {
    auto w = wigner_3j(a[k.ia].l, b[k.ib].l, k.l);
    int n1 = a[k.ia].d(), n2 = b[k.ib].d(), n3 = 2 * k.l + 1;
    double c = std::sqrt((double)n3);
    for (int i = 0; i < n1; ++i)
      for (int j = 0; j < n2; ++j)
        for (int m = 0; m < n3; ++m)
          T[((size_t)(a[k.ia].off + i) * d2 + (b[k.ib].off + j)) * dout + off + m] = c * w[(i * n2 + j) * n3 + m];
    oi.push_back({1, k.l, k.p, off});
    off += n3;
  }
  if (out_irreps) *out_irreps = oi;
  return T;
}

}  // namespace ddmi
