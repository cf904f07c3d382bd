// Reverse-diffusion step on the device (reference utils/sampling.py:117-191):
//   k_perturb           NaN guard (:117-131) + score/noise combination (:133-186) with the
//                       step's scalar coefficients computed on the host in float64 exactly as
//                       the reference's 0-dim float64 tensors are
//   k_modify_conformer  modify_conformer_batch (utils/diffusion_utils.py:60-78): rigid update
//                       (axis_angle_to_matrix, utils/geometry.py:39-86), sequential torsion
//                       updates (utils/torsion.py:75-90), Kabsch re-alignment
//                       (utils/geometry.py:246-276).  The optimal proper rotation is obtained
//                       from Horn's quaternion eigenproblem (float64 Jacobi) instead of an SVD
//                       with reflection fix: the same minimiser, no 3x3 SVD kernel needed.
#include "kernels.h"

namespace ddmi {

// ------------------------------------------------------------------ counter-based RNG
__device__ __forceinline__ void philox4x32(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1,
                                           unsigned* out) {
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0, p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0, n1 = (unsigned)p1;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1, n3 = (unsigned)p0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// standard normal keyed by (seed, sample, step, component)
__device__ __forceinline__ float normal_draw(unsigned long long seed, long long sample, int step, int comp) {
  unsigned o[4];
  philox4x32((unsigned)sample, (unsigned)((unsigned long long)sample >> 32), (unsigned)step, (unsigned)comp,
             (unsigned)seed, (unsigned)(seed >> 32), o);
  const float u1 = ((float)(o[0] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float u2 = ((float)(o[1] >> 8) + 0.5f) * (1.0f / 16777216.0f);
  return sqrtf(-2.0f * logf(u1)) * cosf(6.283185307179586f * u2);
}

// test entry points: the generator exactly as k_perturb uses it
__global__ void k_debug_philox(const unsigned* __restrict__ ctr, const unsigned* __restrict__ key, int n, unsigned* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned o[4];
  philox4x32(ctr[4 * i], ctr[4 * i + 1], ctr[4 * i + 2], ctr[4 * i + 3], key[2 * i], key[2 * i + 1], o);
  for (int j = 0; j < 4; ++j) out[4 * i + j] = o[j];
}
__global__ void k_debug_normal(unsigned long long seed, long long sample0, int n_samples, int step, int n_comp, float* __restrict__ out) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)n_samples * n_comp) return;
  const long long s = i / n_comp;
  out[i] = normal_draw(seed, sample0 + s, step, (int)(i - s * n_comp));
}
void launch_debug_philox(const unsigned* ctr, const unsigned* key, int n, unsigned* out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_philox, dim3(cdiv(n, 256)), dim3(256), 0, s, ctr, key, n, out);
  DDMI_CHECK_HIP(hipGetLastError());
}
void launch_debug_normal(unsigned long long seed, long long sample0, int n_samples, int step, int n_comp, float* out, hipStream_t s) {
  hipLaunchKernelGGL(k_debug_normal, dim3(cdiv((long)n_samples * n_comp, 256)), dim3(256), 0, s, seed, sample0, n_samples, step, n_comp, out);
  DDMI_CHECK_HIP(hipGetLastError());
}

__device__ __forceinline__ float mul_add_rn(float cs, float s, float cz, float z) {
#ifdef DDMI_HIPEMU
  volatile float a = cs * s, b = cz * z;
  return a + b;
#else
  return __fadd_rn(__fmul_rn(cs, s), __fmul_rn(cz, z));
#endif
}

__device__ void nan_fix(float* x, int n, float* red, int tid) {
  // eps = 0.01 * nanmean(|x|);  nan -> eps, +inf -> eps, -inf -> -eps
  float s = 0.f, c = 0.f;
  for (int i = tid; i < n; i += 256) { const float v = fabsf(x[i]); if (!(v != v)) { s += v; c += 1.f; } }
  red[tid] = s; red[256 + tid] = c;
  __syncthreads();
  for (int off = 128; off > 0; off >>= 1) {
    if (tid < off) { red[tid] += red[tid + off]; red[256 + tid] += red[256 + tid + off]; }
    __syncthreads();
  }
  const float eps = 0.01f * (red[0] / red[256]);
  __syncthreads();
  for (int i = tid; i < n; i += 256) {
    const float v = x[i];
    if (v != v) x[i] = eps;
    else if (isinf(v)) x[i] = v > 0.f ? eps : -eps;
  }
}

__global__ __launch_bounds__(256) void k_perturb(PerturbArgs a) {
  __shared__ float red[512];
  __shared__ int flag;
  const int tid = threadIdx.x;
  if (tid == 0) flag = 0;
  __syncthreads();
  for (int b = tid; b < a.B; b += 256) {
    const float m = (a.tr[3 * b] + a.tr[3 * b + 1] + a.tr[3 * b + 2]) / 3.f;
    if (m != m) flag = 1;
  }
  __syncthreads();
  if (flag) {
    nan_fix(a.tr, 3 * a.B, red, tid);
    __syncthreads();
    nan_fix(a.rot, 3 * a.B, red, tid);
    __syncthreads();
    if (a.R > 0) nan_fix(a.tor, a.B * a.R, red, tid);
    __syncthreads();
  }
  for (int i = tid; i < 3 * a.B; i += 256) {
    const int b = i / 3, k = i - 3 * b;
    const long long sid = a.sample_ids ? a.sample_ids[b] : b;
    const float zt = a.z_tr ? a.z_tr[i] : (a.use_rng ? normal_draw(a.seed, sid, a.step, k) : 0.f);
    const float zr = a.z_rot ? a.z_rot[i] : (a.use_rng ? normal_draw(a.seed, sid, a.step, 3 + k) : 0.f);
    a.tr[i] = mul_add_rn(a.c_tr_s, a.tr[i], a.c_tr_z, zt);
    a.rot[i] = mul_add_rn(a.c_rot_s, a.rot[i], a.c_rot_z, zr);
  }
  for (int i = tid; i < a.B * a.R; i += 256) {
    const int b = i / a.R, k = i - b * a.R;
    const long long sid = a.sample_ids ? a.sample_ids[b] : b;
    const float z = a.z_tor ? a.z_tor[i] : (a.use_rng ? normal_draw(a.seed, sid, a.step, 6 + k) : 0.f);
    a.tor[i] = mul_add_rn(a.c_tor_s, a.tor[i], a.c_tor_z, z);
  }
}
void launch_perturb(const PerturbArgs& a, hipStream_t s) {
  hipLaunchKernelGGL(k_perturb, dim3(1), dim3(256), 0, s, a);
  DDMI_CHECK_HIP(hipGetLastError());
}

__global__ void k_fill_times(float* __restrict__ t, int B, float t_tr, float t_rot, float t_tor) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B) { t[i] = t_tr; t[B + i] = t_rot; t[2 * B + i] = t_tor; }
}
// set_time of EVERY step of a sampling loop in one launch: t[k][3][B] (the times are host scalars: they ride in the kernel arguments)
__global__ void k_fill_times_all(float* __restrict__ t, int B, StepTimes st) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
  if (i < B) {
    float* __restrict__ tk = t + (size_t)k * 3 * B;
    tk[i] = st.t[3 * k]; tk[B + i] = st.t[3 * k + 1]; tk[2 * B + i] = st.t[3 * k + 2];
  }
}
void launch_fill_times_all(float* t, int B, const StepTimes& st, hipStream_t s) {
  hipLaunchKernelGGL(k_fill_times_all, dim3(cdiv(B, 256), st.steps), dim3(256), 0, s, t, B, st);
  DDMI_CHECK_HIP(hipGetLastError());
}
void launch_fill_times(float* t, int B, float t_tr, float t_rot, float t_tor, hipStream_t s) {
  hipLaunchKernelGGL(k_fill_times, dim3(cdiv(B, 256)), dim3(256), 0, s, t, B, t_tr, t_rot, t_tor);
  DDMI_CHECK_HIP(hipGetLastError());
}

// ------------------------------------------------------------------ conformer update
__device__ __forceinline__ void axis_angle_to_matrix(float ax, float ay, float az, float* R) {
  const float angle = sqrtf(ax * ax + ay * ay + az * az);
  const float half = 0.5f * angle;
  const float s = fabsf(angle) < 1e-6f ? 0.5f - (angle * angle) / 48.f : sinf(half) / angle;
  const float r = cosf(half), i = ax * s, j = ay * s, k = az * s;
  const float two_s = 2.0f / (r * r + i * i + j * j + k * k);
  R[0] = 1 - two_s * (j * j + k * k); R[1] = two_s * (i * j - k * r); R[2] = two_s * (i * k + j * r);
  R[3] = two_s * (i * j + k * r); R[4] = 1 - two_s * (i * i + k * k); R[5] = two_s * (j * k - i * r);
  R[6] = two_s * (i * k - j * r); R[7] = two_s * (j * k + i * r); R[8] = 1 - two_s * (i * i + j * j);
}

// dominant eigenvector of a symmetric 4x4 (cyclic Jacobi, float64)
__device__ void max_eigvec4(double* A, double* q) {
  double V[16];
  for (int i = 0; i < 16; ++i) V[i] = (i % 5 == 0) ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 30; ++sweep) {
    double off = 0;
    for (int p = 0; p < 4; ++p) for (int r = p + 1; r < 4; ++r) off += A[p * 4 + r] * A[p * 4 + r];
    if (off < 1e-30) break;
    for (int p = 0; p < 3; ++p)
      for (int r = p + 1; r < 4; ++r) {
        const double apr = A[p * 4 + r];
        if (fabs(apr) < 1e-300) continue;
        const double theta = (A[r * 4 + r] - A[p * 4 + p]) / (2.0 * apr);
        const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < 4; ++k) {
          const double akp = A[k * 4 + p], akr = A[k * 4 + r];
          A[k * 4 + p] = c * akp - s * akr;
          A[k * 4 + r] = s * akp + c * akr;
        }
        for (int k = 0; k < 4; ++k) {
          const double apk = A[p * 4 + k], ark = A[r * 4 + k];
          A[p * 4 + k] = c * apk - s * ark;
          A[r * 4 + k] = s * apk + c * ark;
        }
        for (int k = 0; k < 4; ++k) {
          const double vkp = V[k * 4 + p], vkr = V[k * 4 + r];
          V[k * 4 + p] = c * vkp - s * vkr;
          V[k * 4 + r] = s * vkp + c * vkr;
        }
      }
  }
  int best = 0;
  for (int i = 1; i < 4; ++i) if (A[i * 4 + i] > A[best * 4 + best]) best = i;
  for (int k = 0; k < 4; ++k) q[k] = V[k * 4 + best];
}

// block (64 threads) per sample
__global__ __launch_bounds__(64) void k_modify_conformer(float* __restrict__ pos, int Nl, int R,
                                                         const int* __restrict__ rot_u, const int* __restrict__ rot_v,
                                                         const unsigned char* __restrict__ mask_rotate,
                                                         const float* __restrict__ tr, const float* __restrict__ rot,
                                                         const float* __restrict__ tor) {
  DDMI_DYN_SMEM(float, smem);
  float* rigid = smem;            // [Nl][3]
  float* flex = smem + 3 * Nl;    // [Nl][3]
  float* sc = flex + 3 * Nl;      // scratch: centre(3), R(9), t(3)
  const int b = blockIdx.x, tid = threadIdx.x;
  float* p = pos + (size_t)b * Nl * 3;
  for (int i = tid; i < 3 * Nl; i += 64) flex[i] = p[i];
  __syncthreads();
  if (tid == 0) {
    float cx = 0.f, cy = 0.f, cz = 0.f;
    for (int a = 0; a < Nl; ++a) { cx += flex[3 * a]; cy += flex[3 * a + 1]; cz += flex[3 * a + 2]; }
    sc[0] = cx / Nl; sc[1] = cy / Nl; sc[2] = cz / Nl;
    axis_angle_to_matrix(rot[3 * b], rot[3 * b + 1], rot[3 * b + 2], sc + 3);
  }
  __syncthreads();
  for (int a = tid; a < Nl; a += 64) {
    const float x = flex[3 * a] - sc[0], y = flex[3 * a + 1] - sc[1], z = flex[3 * a + 2] - sc[2];
    for (int k = 0; k < 3; ++k)
      rigid[3 * a + k] = (sc[3 + 3 * k] * x + sc[4 + 3 * k] * y + sc[5 + 3 * k] * z) + tr[3 * b + k] + sc[k];
  }
  __syncthreads();
  if (tor == nullptr || R == 0) {
    for (int i = tid; i < 3 * Nl; i += 64) p[i] = rigid[i];
    return;
  }
  for (int i = tid; i < 3 * Nl; i += 64) flex[i] = rigid[i];
  __syncthreads();
  for (int idx = 0; idx < R; ++idx) {
    const int u = rot_u[idx], v = rot_v[idx];
    const float vx = flex[3 * u] - flex[3 * v], vy = flex[3 * u + 1] - flex[3 * v + 1], vz = flex[3 * u + 2] - flex[3 * v + 2];
    const float px = flex[3 * v], py = flex[3 * v + 1], pz = flex[3 * v + 2];
    __syncthreads();
    const float nrm = sqrtf(vx * vx + vy * vy + vz * vz);
    const float th = tor[(size_t)b * R + idx];
    float Rm[9];
    axis_angle_to_matrix(vx / nrm * th, vy / nrm * th, vz / nrm * th, Rm);
    for (int a = tid; a < Nl; a += 64) {
      if (!mask_rotate[(size_t)idx * Nl + a]) continue;
      const float x = flex[3 * a] - px, y = flex[3 * a + 1] - py, z = flex[3 * a + 2] - pz;
      flex[3 * a] = (Rm[0] * x + Rm[1] * y + Rm[2] * z) + px;
      flex[3 * a + 1] = (Rm[3] * x + Rm[4] * y + Rm[5] * z) + py;
      flex[3 * a + 2] = (Rm[6] * x + Rm[7] * y + Rm[8] * z) + pz;
    }
    __syncthreads();
  }
  if (tid == 0) {
    double cA[3] = {0, 0, 0}, cB[3] = {0, 0, 0};
    for (int a = 0; a < Nl; ++a)
      for (int k = 0; k < 3; ++k) { cA[k] += flex[3 * a + k]; cB[k] += rigid[3 * a + k]; }
    for (int k = 0; k < 3; ++k) { cA[k] /= Nl; cB[k] /= Nl; }
    double S[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int a = 0; a < Nl; ++a)
      for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) S[3 * i + j] += (flex[3 * a + i] - cA[i]) * (rigid[3 * a + j] - cB[j]);
    const double Sxx = S[0], Sxy = S[1], Sxz = S[2], Syx = S[3], Syy = S[4], Syz = S[5], Szx = S[6], Szy = S[7], Szz = S[8];
    double N[16] = {Sxx + Syy + Szz, Syz - Szy, Szx - Sxz, Sxy - Syx,
                    Syz - Szy, Sxx - Syy - Szz, Sxy + Syx, Szx + Sxz,
                    Szx - Sxz, Sxy + Syx, -Sxx + Syy - Szz, Syz + Szy,
                    Sxy - Syx, Szx + Sxz, Syz + Szy, -Sxx - Syy + Szz};
    double q[4];
    max_eigvec4(N, q);
    const double w = q[0], x = q[1], y = q[2], z = q[3];
    const double n2 = w * w + x * x + y * y + z * z;
    const double s2 = 2.0 / n2;
    double Rk[9] = {1 - s2 * (y * y + z * z), s2 * (x * y - z * w), s2 * (x * z + y * w),
                    s2 * (x * y + z * w), 1 - s2 * (x * x + z * z), s2 * (y * z - x * w),
                    s2 * (x * z - y * w), s2 * (y * z + x * w), 1 - s2 * (x * x + y * y)};
    for (int i = 0; i < 9; ++i) sc[3 + i] = (float)Rk[i];
    for (int i = 0; i < 3; ++i) sc[12 + i] = (float)(cB[i] - (Rk[3 * i] * cA[0] + Rk[3 * i + 1] * cA[1] + Rk[3 * i + 2] * cA[2]));
  }
  __syncthreads();
  for (int a = tid; a < Nl; a += 64) {
    const float x = flex[3 * a], y = flex[3 * a + 1], z = flex[3 * a + 2];
    for (int k = 0; k < 3; ++k) p[3 * a + k] = (sc[3 + 3 * k] * x + sc[4 + 3 * k] * y + sc[5 + 3 * k] * z) + sc[12 + k];
  }
}
void launch_modify_conformer(float* pos, int B, int Nl, int R, const int* rot_u, const int* rot_v,
                             const unsigned char* mask_rotate, const float* tr, const float* rot, const float* tor,
                             hipStream_t s) {
  if (B <= 0) return;
  const size_t smem = (size_t)(6 * Nl + 16) * sizeof(float);
  hipLaunchKernelGGL(k_modify_conformer, dim3(B), dim3(64), smem, s, pos, Nl, R, rot_u, rot_v, mask_rotate, tr, rot, tor);
  DDMI_CHECK_HIP(hipGetLastError());
}

}  // namespace ddmi
