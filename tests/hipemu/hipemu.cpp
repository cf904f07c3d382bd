// hipemu runtime: fiber scheduler + wave collectives.  TEST INFRASTRUCTURE ONLY.
#include <hip/hip_runtime.h>

#include <chrono>
#include <vector>
#include <sys/mman.h>

uint3_ threadIdx, blockIdx;
dim3 blockDim, gridDim;

namespace {
constexpr size_t STACK = 256 * 1024;
struct Fiber {
  void* sp = nullptr;
  char* stack = nullptr;
  bool done = false;
  uint3_ tid;
};
std::vector<Fiber> fibers;
void* main_sp = nullptr;
int cur = -1, n_threads = 0, n_alive = 0;
const std::function<void()>* body_ptr = nullptr;
// block barrier
int bar_arrived = 0;
unsigned bar_gen = 0;
// wave state
struct Wave {
  int arrived = 0;
  unsigned gen = 0;
  int alive = 0;
  float fa[64], fb[64];
  unsigned ua[64][4], ub[64][4];   // 8 bf16 per lane (v_mfma_f32_16x16x32_bf16)
  int ia[64];
};
std::vector<Wave> waves;
std::vector<char> dyn_smem;

extern "C" void hipemu_switch(void** from_sp, void* to_sp);
asm(R"(
.text
.globl hipemu_switch
.type hipemu_switch,@function
hipemu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
)");

void yield_to_main() { hipemu_switch(&fibers[cur].sp, main_sp); }

void fiber_entry() {
  (*body_ptr)();
  Fiber& f = fibers[cur];
  f.done = true;
  n_alive--;
  waves[cur / 64].alive--;
  yield_to_main();
  abort();
}

void prepare(Fiber& f) {
  if (!f.stack) {
    f.stack = (char*)mmap(nullptr, STACK, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
    if (f.stack == (char*)MAP_FAILED) { perror("mmap"); abort(); }
  }
  uintptr_t top = ((uintptr_t)f.stack + STACK) & ~(uintptr_t)15;
  void** p = (void**)top;
  *--p = nullptr;               // fake return address for fiber_entry
  *--p = (void*)&fiber_entry;   // popped by 'ret'
  for (int i = 0; i < 6; ++i) *--p = nullptr;
  f.sp = (void*)p;
  f.done = false;
}

void wave_sync() {
  Wave& w = waves[cur / 64];
  unsigned g = w.gen;
  w.arrived++;
  while (w.gen == g) {
    if (w.arrived >= w.alive) { w.arrived = 0; w.gen++; break; }
    yield_to_main();
  }
}
inline int lane() { return cur % 64; }
}  // namespace

void* hipemu_dyn_smem() { return dyn_smem.data(); }
void hipemu_wave_sync() { wave_sync(); threadIdx = fibers[cur].tid; }

void hipemu_launch(const std::function<void()>& body, dim3 grid, dim3 block, size_t dyn) {
  n_threads = block.x * block.y * block.z;
  if (n_threads <= 0 || n_threads > 1024) { fprintf(stderr, "hipemu: bad block size %d\n", n_threads); abort(); }
  if ((int)fibers.size() < n_threads) fibers.resize(n_threads);
  dyn_smem.assign(dyn + 64, 0);
  blockDim = block;
  gridDim = grid;
  body_ptr = &body;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blockIdx = {bx, by, bz};
        waves.assign((n_threads + 63) / 64, Wave());
        bar_arrived = 0;
        n_alive = n_threads;
        for (int t = 0; t < n_threads; ++t) {
          prepare(fibers[t]);
          fibers[t].tid = {t % block.x, (t / block.x) % block.y, t / (block.x * block.y)};
          waves[t / 64].alive++;
        }
        while (n_alive > 0) {
          for (int t = 0; t < n_threads; ++t) {
            if (fibers[t].done) continue;
            cur = t;
            threadIdx = fibers[t].tid;
            hipemu_switch(&main_sp, fibers[t].sp);
          }
        }
      }
  cur = -1;
}

void __syncthreads() {
  unsigned g = bar_gen;
  bar_arrived++;
  while (bar_gen == g) {
    if (bar_arrived >= n_alive) { bar_arrived = 0; bar_gen++; break; }
    yield_to_main();
  }
  threadIdx = fibers[cur].tid;
}

float hipemu_shfl(float v, int src) {
  Wave& w = waves[cur / 64];
  w.fa[lane()] = v;
  wave_sync();
  float r = w.fa[src & 63];
  wave_sync();
  threadIdx = fibers[cur].tid;
  return r;
}
float __shfl_xor(float v, int mask, int) { return hipemu_shfl(v, lane() ^ mask); }
float __shfl_down(float v, unsigned d, int) { int s = lane() + (int)d; return hipemu_shfl(v, s < 64 ? s : lane()); }
static int shfl_i(int v, int src) {
  Wave& w = waves[cur / 64];
  w.ia[lane()] = v;
  wave_sync();
  int r = w.ia[src & 63];
  wave_sync();
  threadIdx = fibers[cur].tid;
  return r;
}
int __shfl_xor(int v, int mask, int) { return shfl_i(v, lane() ^ mask); }
int __shfl_down(int v, unsigned d, int) { int s = lane() + (int)d; return shfl_i(v, s < 64 ? s : lane()); }
int __shfl(int v, int l, int) { return shfl_i(v, l); }

f32x4_emu __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4_emu c, int, int, int) {
  Wave& w = waves[cur / 64];
  int l = lane();
  w.fa[l] = a;
  w.fb[l] = b;
  wave_sync();
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) acc = fmaf(w.fa[row + 16 * k], w.fb[col + 16 * k], acc);
    c[r] = acc;
  }
  wave_sync();
  threadIdx = fibers[cur].tid;
  return c;
}

// v_mfma_f32_4x4x1_16b_f32: 16 independent 4x4 outer products; block = lane / 4, A row / B column = lane % 4,
// D[i][j] of block b in register i of lane 4b + j (layout checked on the MI355X: tools/probe/mfma4x4.hip)
f32x4_emu __builtin_amdgcn_mfma_f32_4x4x1f32(float a, float b, f32x4_emu c, int, int, int) {
  Wave& w = waves[cur / 64];
  int l = lane();
  w.fa[l] = a;
  w.fb[l] = b;
  wave_sync();
  for (int i = 0; i < 4; ++i) c[i] = fmaf(w.fa[(l & ~3) + i], w.fb[l], c[i]);
  wave_sync();
  threadIdx = fibers[cur].tid;
  return c;
}

// v_mfma_f32_16x16x32_bf16: lane l holds row (A) / column (B) l & 15 and the eight K slots 8 (l >> 4) .. + 7 (two bf16 per
// register, element 2i in the low half of register i); products exact in f32, accumulated in f32 (the order of the hardware's
// internal sum is not modelled: the tests compare at a tolerance)
static inline float bf16_bits(unsigned h) { unsigned u = h << 16; float f; memcpy(&f, &u, 4); return f; }
f32x4_emu hipemu_mfma_f32_16x16x32_bf16(u32x4_emu a, u32x4_emu b, f32x4_emu c) {
  Wave& w = waves[cur / 64];
  int l = lane();
  for (int i = 0; i < 4; ++i) { w.ua[l][i] = a[i]; w.ub[l][i] = b[i]; }
  wave_sync();
  int col = l & 15;
  for (int r = 0; r < 4; ++r) {
    int row = 4 * (l >> 4) + r;
    float acc = c[r];
    for (int g = 0; g < 4; ++g)
      for (int i = 0; i < 4; ++i) {
        const unsigned ar = w.ua[row + 16 * g][i], br = w.ub[col + 16 * g][i];
        acc += bf16_bits(ar & 0xffffu) * bf16_bits(br & 0xffffu);
        acc += bf16_bits(ar >> 16) * bf16_bits(br >> 16);
      }
    c[r] = acc;
  }
  wave_sync();
  threadIdx = fibers[cur].tid;
  return c;
}

unsigned long long __ballot(int pred) {
  Wave& w = waves[cur / 64];
  w.ia[lane()] = pred != 0;
  wave_sync();
  unsigned long long m = 0;
  for (int i = 0; i < 64; ++i) m |= (unsigned long long)(w.ia[i] != 0) << i;
  wave_sync();
  threadIdx = fibers[cur].tid;
  return m;
}

f32x16_emu __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16_emu c, int, int, int) {
  Wave& w = waves[cur / 64];
  int l = lane();
  w.fa[l] = a;
  w.fb[l] = b;
  wave_sync();
  int col = l & 31;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) acc = fmaf(w.fa[row + 32 * k], w.fb[col + 32 * k], acc);
    c[r] = acc;
  }
  wave_sync();
  threadIdx = fibers[cur].tid;
  return c;
}

struct hipEvent_s { std::chrono::steady_clock::time_point t; };
hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipEvent_s; return hipSuccess; }
hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = std::chrono::steady_clock::now(); return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) {
  *ms = std::chrono::duration<float, std::milli>(b->t - a->t).count();
  return hipSuccess;
}
