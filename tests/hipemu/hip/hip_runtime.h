// hipemu -- a tiny single-threaded HIP execution model for the CPU test suite.
//
// TEST INFRASTRUCTURE ONLY.  It lets `pytest -m "not gpu"` run the *unmodified* kernel
// sources of diffdock_amd/csrc on the host (there is no GPU in the build container) so
// that indexing, LDS hand-offs, MFMA fragment layouts and the host orchestration are
// debugged before GPU minutes are spent.  It is never linked into libddmi.so and the
// product never loads it (diffdock_amd/lib.py only opens the gfx950 build).
//
// Model: blocks run one after another; the threads of a block are fibers on one OS
// thread, resumed round-robin.  __syncthreads() and the wave-collective builtins
// (shuffles, MFMA) are rendezvous points.  Wave = 64 lanes.  MFMA fragment layouts follow
// /opt/skills/guides/cdna_hip_programming.md section 3:
//   16x16x4 f32 : A[i=l&15][k=l>>4], B[k=l>>4][j=l&15], D col=l&15,row=4*(l>>4)+r
//   32x32x2 f32 : A[i=l&31][k=l>>5], B[k=l>>5][j=l&31], D col=l&31,row=(r&3)+8*(r>>2)+4*(l>>5)
#pragma once
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>

#define DDMI_HIPEMU 1
#define __global__
#define __device__
#define __host__
#define __shared__ static
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __restrict__ __restrict

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct uint3_ { unsigned x, y, z; };
extern uint3_ threadIdx, blockIdx;
extern dim3 blockDim, gridDim;

struct float2 { float x, y; };
struct float3 { float x, y, z; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float3 make_float3(float x, float y, float z) { return float3{x, y, z}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline float2 make_float2(float x, float y) { return float2{x, y}; }

typedef float f32x4_emu __attribute__((ext_vector_type(4)));
typedef float f32x16_emu __attribute__((ext_vector_type(16)));
typedef unsigned u32x4_emu __attribute__((ext_vector_type(4)));

// ---- runtime API subset --------------------------------------------------------------
typedef int hipError_t;
typedef void* hipStream_t;
typedef struct hipEvent_s* hipEvent_t;
enum { hipSuccess = 0 };
enum hipMemcpyKind { hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
static inline const char* hipGetErrorString(hipError_t) { return "hipemu"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipDeviceAttribute_t { hipDeviceAttributeMultiprocessorCount = 0 };
static inline hipError_t hipDeviceGetAttribute(int* v, hipDeviceAttribute_t, int) { *v = 4; return hipSuccess; }
static inline hipError_t hipMalloc(void** p, size_t n) { *p = aligned_alloc(256, (n + 255) / 256 * 256 + 256); return *p ? hipSuccess : 1; }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { *p = malloc(n ? n : 1); return *p ? hipSuccess : 1; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { memcpy(d, s, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemset(void* d, int v, size_t n) { memset(d, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
enum { hipStreamNonBlocking = 1 };
// (distinct non-null handles: the multi-stream code paths of complex.cpp -- side stream, preparation streams -- run under the
// emulator too, in issue order)
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { static char handles[64]; static int n = 0; *s = (hipStream_t)(handles + (n++ & 63)); return hipSuccess; }
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
hipError_t hipEventCreate(hipEvent_t* e);
enum { hipEventDisableTiming = 2 };
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { return hipEventCreate(e); }
hipError_t hipEventDestroy(hipEvent_t e);
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s);
hipError_t hipEventSynchronize(hipEvent_t e);
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b);

enum hipFuncAttribute { hipFuncAttributeMaxDynamicSharedMemorySize = 8 };
static inline hipError_t hipFuncSetAttribute(const void*, hipFuncAttribute, int) { return hipSuccess; }

// ---- launch --------------------------------------------------------------------------
void hipemu_launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t dyn_smem);
void* hipemu_dyn_smem();
#define hipLaunchKernelGGL(kernel, grid, block, smem, stream, ...) \
  hipemu_launch([&]() { kernel(__VA_ARGS__); }, dim3(grid), dim3(block), (size_t)(smem))

// ---- device builtins -----------------------------------------------------------------
void __syncthreads();
void hipemu_wave_sync();   // rendezvous of the live lanes of the calling wave
float hipemu_shfl(float v, int src_lane);
static inline float __shfl(float v, int lane, int width = 64) { (void)width; return hipemu_shfl(v, lane); }
float __shfl_xor(float v, int mask, int width = 64);
float __shfl_down(float v, unsigned delta, int width = 64);
int __shfl_xor(int v, int mask, int width = 64);
int __shfl_down(int v, unsigned delta, int width = 64);
int __shfl(int v, int lane, int width = 64);
f32x4_emu __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4_emu c, int, int, int);
f32x4_emu __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, f32x4_emu c, int, int, int);
f32x4_emu hipemu_mfma_f32_16x16x32_bf16(u32x4_emu a, u32x4_emu b, f32x4_emu c);
unsigned long long __ballot(int pred);
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
f32x16_emu __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16_emu c, int, int, int);

template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
using std::isinf; using std::isnan;
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __fdividef(float a, float b) { return a / b; }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
template <typename T> static inline T min(T a, T b) { return a < b ? a : b; }
template <typename T> static inline T max(T a, T b) { return a > b ? a : b; }
static inline unsigned long long __umul64hi_emu(unsigned long long a, unsigned long long b) {
  return (unsigned long long)(((unsigned __int128)a * b) >> 64);
}
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }
