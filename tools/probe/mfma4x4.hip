// Probe: register layout of v_mfma_f32_4x4x1_16b_f32 and of the gfx950 half / row swaps (v_permlane32_swap, v_permlane16_swap).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const float* a, const float* b, float* d, int* sw) {
  const int l = threadIdx.x;
  f32x4 c = {0.f, 0.f, 0.f, 0.f};
  c = __builtin_amdgcn_mfma_f32_4x4x1f32(a[l], b[l], c, 0, 0, 0);
  for (int i = 0; i < 4; ++i) d[i * 64 + l] = c[i];
  // swaps
  int v0 = 1000 + l, v1 = 2000 + l;
  auto r = __builtin_amdgcn_permlane32_swap(v0, v1, false, false);
  sw[l] = r[0]; sw[64 + l] = r[1];
  auto q = __builtin_amdgcn_permlane16_swap(v0, v1, false, false);
  sw[128 + l] = q[0]; sw[192 + l] = q[1];
}
int main() {
  float ha[64], hb[64], hd[256]; int hs[256];
  for (int l = 0; l < 64; ++l) { ha[l] = 1 + l; hb[l] = 100 * (1 + l); }
  float *a, *b, *d; int* s;
  hipMalloc(&a, 256); hipMalloc(&b, 256); hipMalloc(&d, 1024); hipMalloc(&s, 1024);
  hipMemcpy(a, ha, 256, hipMemcpyHostToDevice); hipMemcpy(b, hb, 256, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, a, b, d, s);
  hipMemcpy(hd, d, 1024, hipMemcpyDeviceToHost); hipMemcpy(hs, s, 1024, hipMemcpyDeviceToHost);
  // expected (assumed): D vgpr i, lane 4b+j = A_b[i] * B_b[j], A_b[i] = a[4b+i], B_b[j] = b[4b+j]
  int bad = 0;
  for (int i = 0; i < 4; ++i) for (int l = 0; l < 64; ++l) {
    const int bk = l / 4, j = l % 4;
    const float e = ha[4 * bk + i] * hb[4 * bk + j];
    if (hd[i * 64 + l] != e) { if (bad < 8) printf("mismatch vgpr %d lane %d: %g expected %g\n", i, l, hd[i * 64 + l], e); ++bad; }
  }
  printf("mfma 4x4x1 layout mismatches: %d\n", bad);
  printf("permlane32_swap r0:"); for (int l = 0; l < 64; l += 8) printf(" %d", hs[l]); printf("\n                r1:"); for (int l = 0; l < 64; l += 8) printf(" %d", hs[64 + l]);
  printf("\npermlane16_swap q0:"); for (int l = 0; l < 64; l += 8) printf(" %d", hs[128 + l]); printf("\n                q1:"); for (int l = 0; l < 64; l += 8) printf(" %d", hs[192 + l]);
  printf("\n");
  return 0;
}
