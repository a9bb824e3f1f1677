// hostsim -- a tiny host-side stand-in for <hip/hip_runtime.h>.  TEST INFRASTRUCTURE ONLY.
//
// tests/hostsim/build.py compiles the product's kernel sources (sc-sfmlearner-release_amd/csrc)
// UNCHANGED with g++ against this header, so that the CPU-only CI (`pytest -m "not gpu"`) can run
// the kernels' tiling / halo / reduction / gradient logic against the oracle without a GPU.  It is
// never shipped, never loaded by the product (scsfm_hip/_lib.py only ever opens libscsfm_hip.so,
// which is built by hipcc for gfx950) and proves nothing about performance.
//
// Execution model: one OS thread.  Every HIP thread of a workgroup is a fiber; blocks run
// one after another; __syncthreads() and the wave-level shuffles are cooperative barriers among
// the fibers of the block / of the 64-lane wave.  Atomics are plain read-modify-writes.
// Fibers switch through a 20-instruction register swap on x86-64 (glibc's swapcontext makes two
// rt_sigprocmask system calls per switch, and a kernel of the product switches 10^7 times per launch:
// the simulated kernels were 70 % of the CPU suite's time); -DHOSTSIM_UCONTEXT, and every other
// architecture, keeps the portable ucontext fibers.  A fiber parked at a barrier whose generation has
// not moved is not switched to at all.
#pragma once
#if !defined(__x86_64__) && !defined(HOSTSIM_UCONTEXT)
#define HOSTSIM_UCONTEXT 1
#endif
#ifdef HOSTSIM_UCONTEXT
#include <ucontext.h>
#endif

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static
#define __launch_bounds__(...)
#define __HOSTSIM__ 1

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
static inline float2 make_float2(float x, float y) { return {x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return {x, y, z, w}; }

typedef void* hipStream_t;
typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1 };
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hostsim"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
typedef void* hipEvent_t;
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = nullptr; return hipSuccess; }
static inline hipError_t hipEventDestroy(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t, hipStream_t) { return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t, hipEvent_t) { *ms = 0.0f; return hipSuccess; }

namespace hostsim {
constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

#ifdef HOSTSIM_UCONTEXT
struct Ctx { ucontext_t uc; };
inline void ctx_switch(Ctx& from, Ctx& to) { swapcontext(&from.uc, &to.uc); }
inline void ctx_make(Ctx& c, Ctx& back, char* stack, size_t size, void (*fn)()) {
  getcontext(&c.uc);
  c.uc.uc_stack.ss_sp = stack;
  c.uc.uc_stack.ss_size = size;
  c.uc.uc_link = &back.uc;
  makecontext(&c.uc, fn, 0);
}
#else
// hostsim_switch(&save_sp, load_sp): push the callee-saved registers and the floating-point control words, park the stack
// pointer, adopt the other one, pop.  Weak + hidden: every translation unit of the library carries a copy, the linker
// keeps one.
extern "C" void hostsim_switch(void** save_sp, void* load_sp) __attribute__((visibility("hidden")));
asm(R"(
    .text
    .weak hostsim_switch
    .hidden hostsim_switch
    .type hostsim_switch,@function
hostsim_switch:
    pushq %rbp
    pushq %rbx
    pushq %r12
    pushq %r13
    pushq %r14
    pushq %r15
    subq $8, %rsp
    stmxcsr (%rsp)
    fnstcw 4(%rsp)
    movq %rsp, (%rdi)
    movq %rsi, %rsp
    ldmxcsr (%rsp)
    fldcw 4(%rsp)
    addq $8, %rsp
    popq %r15
    popq %r14
    popq %r13
    popq %r12
    popq %rbx
    popq %rbp
    ret
    .size hostsim_switch, .-hostsim_switch
)");
struct Ctx { void* sp = nullptr; };
inline void ctx_switch(Ctx& from, Ctx& to) { hostsim_switch(&from.sp, to.sp); }
inline void ctx_make(Ctx& c, Ctx&, char* stack, size_t size, void (*fn)()) {
  // the frame hostsim_switch pops: control words, r15 r14 r13 r12 rbx rbp, return address = fn (entered with the stack
  // pointer at 8 mod 16, as after a call); fn never returns (the trampoline switches away for good)
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  uint64_t* f = (uint64_t*)(top - 72);
  unsigned csr = 0; unsigned short cw = 0;
  asm volatile("stmxcsr %0" : "=m"(csr));
  asm volatile("fnstcw %0" : "=m"(cw));
  f[0] = (uint64_t)csr | ((uint64_t)cw << 32);
  for (int i = 1; i <= 6; ++i) f[i] = 0;
  f[7] = (uint64_t)(uintptr_t)fn;  // popped by `ret`: fn starts with the stack pointer at top - 8
  f[8] = 0;                        // (where a caller's return address would be)
  c.sp = (void*)f;
}
#endif

struct Fiber {
  Ctx ctx;
  bool done = false;
  int wait = 0;           // 0: runnable; 1: parked at the block barrier; 2: at its wave's barrier
  uint64_t wait_gen = 0;  // ... of this generation
};

struct State {
  dim3 grid, block;
  int nthreads = 0, cur = 0, live = 0;
  std::vector<Fiber> fibers;
  std::vector<char> stacks;
  Ctx main_ctx;
  const std::function<void()>* body = nullptr;
  // block barrier
  int arrived = 0;
  uint64_t gen = 0;
  // wave barriers + exchange buffers
  std::vector<int> w_arrived, w_live;
  std::vector<uint64_t> w_gen;
  std::vector<uint64_t> xbuf;
};
inline State& S() { static State s; return s; }

struct Idx { unsigned x, y, z; };
inline Idx& tidx() { static Idx v; return v; }
inline Idx& bidx() { static Idx v; return v; }
inline Idx& bdim() { static Idx v; return v; }
inline Idx& gdim() { static Idx v; return v; }

inline void set_thread(int t) {
  State& s = S();
  s.cur = t;
  tidx().x = t % s.block.x;
  tidx().y = (t / s.block.x) % s.block.y;
  tidx().z = t / (s.block.x * s.block.y);
}
inline void yield() {
  State& s = S();
  int me = s.cur;
  ctx_switch(s.fibers[me].ctx, s.main_ctx);
}
inline void trampoline() {
  State& s = S();
  (*s.body)();
  s.fibers[s.cur].done = true;
  ctx_switch(s.fibers[s.cur].ctx, s.main_ctx);
  abort();  // (a finished fiber is never resumed)
}
inline void block_barrier() {
  State& s = S();
  uint64_t g = s.gen;
  if (++s.arrived >= s.live) { s.arrived = 0; s.gen++; return; }
  Fiber& f = s.fibers[s.cur];
  f.wait = 1; f.wait_gen = g;
  while (s.gen == g) yield();
  f.wait = 0;
}
inline void wave_barrier() {
  State& s = S();
  int w = s.cur / kWave;
  uint64_t g = s.w_gen[w];
  if (++s.w_arrived[w] >= s.w_live[w]) { s.w_arrived[w] = 0; s.w_gen[w]++; return; }
  Fiber& f = s.fibers[s.cur];
  f.wait = 2; f.wait_gen = g;
  while (s.w_gen[w] == g) yield();
  f.wait = 0;
}
template <class T>
inline T shfl(T v, int src_lane) {
  static_assert(sizeof(T) <= 8, "shuffle payload");
  State& s = S();
  int w = s.cur / kWave, lane = s.cur % kWave;
  uint64_t bits = 0;
  memcpy(&bits, &v, sizeof(T));
  s.xbuf[w * kWave + lane] = bits;
  wave_barrier();
  T r = v;
  int nl = s.nthreads - w * kWave; if (nl > kWave) nl = kWave;
  if (src_lane >= 0 && src_lane < nl) memcpy(&r, &s.xbuf[w * kWave + src_lane], sizeof(T));
  wave_barrier();
  return r;
}

inline void run_block(const std::function<void()>& body) {
  State& s = S();
  int n = s.nthreads;
  s.body = &body;
  s.live = n; s.arrived = 0;
  int nw = (n + kWave - 1) / kWave;
  s.w_arrived.assign(nw, 0); s.w_gen.assign(nw, 0); s.w_live.assign(nw, 0);
  for (int t = 0; t < n; ++t) s.w_live[t / kWave]++;
  s.xbuf.assign((size_t)nw * kWave, 0);
  for (int t = 0; t < n; ++t) {
    Fiber& f = s.fibers[t];
    f.done = false; f.wait = 0;
    ctx_make(f.ctx, s.main_ctx, s.stacks.data() + (size_t)t * kStack, kStack, (void (*)())trampoline);
  }
  int remaining = n;
  long spins = 0;
  // HOSTSIM_ORDER=reverse runs the threads of a block (and, in launch(), the blocks of a grid) in descending
  // order: a result that depends on the order in which threads run between two barriers is a data race
  // (tests/test_hostsim_kernels.py runs the fused kernels under both orders).
  const char* order = getenv("HOSTSIM_ORDER");
  const bool reverse = order && order[0] == 'r';
  while (remaining > 0) {
    int progressed = 0;
    for (int k = 0; k < n; ++k) {
      const int t = reverse ? n - 1 - k : k;
      Fiber& f = s.fibers[t];
      if (f.done) continue;
      // parked at a barrier that has not opened: nothing to run (it would look at the generation and yield again)
      if (f.wait == 1 ? s.gen == f.wait_gen : (f.wait == 2 && s.w_gen[t / kWave] == f.wait_gen)) continue;
      set_thread(t);
      ctx_switch(s.main_ctx, f.ctx);
      if (f.done) {
        --remaining; ++progressed;
        // a finished thread no longer takes part in barriers (hardware counts live waves only)
        --s.live; --s.w_live[t / kWave];
        if (s.live > 0 && s.arrived >= s.live) { s.arrived = 0; s.gen++; }
        int w = t / kWave;
        if (s.w_live[w] > 0 && s.w_arrived[w] >= s.w_live[w]) { s.w_arrived[w] = 0; s.w_gen[w]++; }
      }
    }
    if (++spins > 100000000L) { fprintf(stderr, "hostsim: deadlock (divergent barrier?)\n"); abort(); }
  }
}

inline void launch(dim3 grid, dim3 block, const std::function<void()>& body) {
  State& s = S();
  s.grid = grid; s.block = block;
  s.nthreads = block.x * block.y * block.z;
  if ((int)s.fibers.size() < s.nthreads) {
    s.fibers.resize(s.nthreads);
    s.stacks.resize((size_t)s.nthreads * kStack);
  }
  bdim() = {block.x, block.y, block.z};
  gdim() = {grid.x, grid.y, grid.z};
  const char* order = getenv("HOSTSIM_ORDER");
  const bool reverse = order && order[0] == 'r';
  const unsigned long nb = (unsigned long)grid.x * grid.y * grid.z;
  for (unsigned long k = 0; k < nb; ++k) {
    const unsigned long l = reverse ? nb - 1 - k : k;
    bidx() = {(unsigned)(l % grid.x), (unsigned)((l / grid.x) % grid.y), (unsigned)(l / ((unsigned long)grid.x * grid.y))};
    run_block(body);
  }
}
}  // namespace hostsim

#define threadIdx (hostsim::tidx())
#define blockIdx (hostsim::bidx())
#define blockDim (hostsim::bdim())
#define gridDim (hostsim::gdim())
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...)                      \
  do {                                                                                   \
    (void)(shmem); (void)(stream);                                                       \
    hostsim::launch(dim3(grid), dim3(block), [&]() { kernel(__VA_ARGS__); });            \
  } while (0)

static inline void __syncthreads() { hostsim::block_barrier(); }
template <class T> static inline T __shfl_down(T v, unsigned d, int = 64) { return hostsim::shfl(v, hostsim::S().cur % 64 + (int)d); }
template <class T> static inline T __shfl_up(T v, unsigned d, int = 64) { return hostsim::shfl(v, hostsim::S().cur % 64 - (int)d); }
template <class T> static inline T __shfl_xor(T v, int m, int = 64) { return hostsim::shfl(v, (hostsim::S().cur % 64) ^ m); }
template <class T> static inline T __shfl(T v, int l, int = 64) { return hostsim::shfl(v, l); }

template <class T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
static inline float atomicAdd(float* p, double v) { float o = *p; *p = o + (float)v; return o; }
// scoped atomics (fibres never run concurrently, so a plain read-modify-write is atomic)
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_AGENT 3
template <class T> static inline T __hip_atomic_fetch_add(T* p, T v, int, int) { T o = *p; *p = o + v; return o; }
template <class T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
static inline void __threadfence() {}

// (__expf is declared by glibc itself)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline float __frcp_rn(float x) { return 1.0f / x; }
static inline float __saturatef(float x) { return x < 0 ? 0 : (x > 1 ? 1 : x); }
static inline int __float2int_rd(float x) { return (int)floorf(x); }
#define __builtin_amdgcn_readfirstlane(x) (x)
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_exp2f(float x) { return exp2f(x); }

// DPP wave shifts (v_*_dpp wave_shr:1 = 0x138: lane <- lane - 1; wave_shl:1 = 0x130: lane <- lane + 1); lanes without a
// source keep `old`, or read 0 with bound_ctrl
static inline int __builtin_amdgcn_update_dpp(int old, int src, int ctrl, int, int, bool bound_ctrl) {
  const int lane = hostsim::S().cur % 64;
  const int from = ctrl == 0x138 ? lane - 1 : (ctrl == 0x130 ? lane + 1 : -1000);
  if (from == -1000) { fprintf(stderr, "hostsim: unsupported dpp_ctrl %x\n", ctrl); abort(); }
  const int got = hostsim::shfl(src, from);
  hostsim::State& s = hostsim::S();
  const int w = s.cur / 64;
  int nl = s.nthreads - w * 64; if (nl > 64) nl = 64;
  if (from < 0 || from >= nl) return bound_ctrl ? 0 : old;
  return got;
}
static inline float __builtin_amdgcn_fmed3f(float x, float a, float b) {
  // median of three; NaN-safe the way the hardware is (a NaN operand yields the minimum of the others)
  if (x != x) return a < b ? a : b;
  const float lo = a < b ? a : b, hi = a < b ? b : a;
  return x < lo ? lo : (x > hi ? hi : x);
}
static inline void __builtin_amdgcn_wave_barrier() { hostsim::wave_barrier(); }
enum { hipDeviceAttributeMultiprocessorCount = 1 };
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipDeviceGetAttribute(int* v, int, int) { *v = 256; return hipSuccess; }
