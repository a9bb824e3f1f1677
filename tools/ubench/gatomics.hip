// Global atomic throughput on MI355X (gfx950) for the access pattern of the scatter window's flush: a wave adds to
// 64 consecutive cells of a row, workgroups walk their own rows of a plane that fits the L2s / Infinity Cache.
// Variants: element type (f32 hardware atomic, u32, u64 = two packed 32-bit cells, f64), the share of lanes that
// are active (the flush skips zero cells), and plain stores / load+store as the reference points.
//   hipcc -O3 --offload-arch=gfx950 -munsafe-fp-atomics -o gatomics gatomics.hip && ./gatomics
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// MODE 0: atomic add (no return), 1: store, 2: load + store.  KEEP: a lane is active if (lane * 7 + row) % 8 < KEEP.
template <typename T, int MODE, int KEEP>
__global__ __launch_bounds__(256) void k(T* plane, size_t elems, int rows_per_block, int pitch, T v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = ((size_t)blockIdx.x * rows_per_block) * pitch % elems;
  for (int r = wave; r < rows_per_block; r += 4) {
    if (((lane * 7 + r) & 7) >= KEEP) continue;
    T* q = plane + (base + (size_t)r * pitch + lane) % elems;
    if (MODE == 0) (void)__hip_atomic_fetch_add(q, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else if (MODE == 1) *q = v;
    else *q += v;
  }
}

template <typename T, int MODE, int KEEP>
static void run(const char* what, void* plane, size_t bytes) {
  const int blocks = 16384, rows = 32;
  const size_t elems = bytes / sizeof(T);
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  for (int pitch : {64, 832}) {
    auto go = [&] { hipLaunchKernelGGL((k<T, MODE, KEEP>), dim3(blocks), dim3(256), 0, 0, (T*)plane, elems, rows, pitch, T(1)); };
    go(); CHECK(hipDeviceSynchronize());
    CHECK(hipEventRecord(e0)); go(); go(); go(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    const double lanes = double(blocks) * rows * 64 * KEEP / 8;
    printf("%-28s active %d/8 pitch %4d : %6.1f M lane-ops in %.3f ms = %7.1f G lane-ops/s = %7.1f G 32-bit cells/s\n", what, KEEP,
           pitch, lanes * 1e-6, ms, lanes / ms * 1e-6, lanes * (sizeof(T) / 4) / ms * 1e-6);
    fflush(stdout);
  }
}

int main() {
  const size_t bytes = 40u << 20;
  void* plane; CHECK(hipMalloc(&plane, bytes)); CHECK(hipMemset(plane, 0, bytes));
  run<float, 0, 8>("atomic add f32", plane, bytes);
  run<unsigned, 0, 8>("atomic add u32", plane, bytes);
  run<unsigned long long, 0, 8>("atomic add u64 (2 cells)", plane, bytes);
  run<double, 0, 8>("atomic add f64", plane, bytes);
  run<float, 0, 3>("atomic add f32", plane, bytes);
  run<unsigned, 0, 3>("atomic add u32", plane, bytes);
  run<unsigned long long, 0, 3>("atomic add u64 (2 cells)", plane, bytes);
  run<float, 0, 1>("atomic add f32", plane, bytes);
  run<unsigned long long, 0, 1>("atomic add u64 (2 cells)", plane, bytes);
  run<float, 1, 8>("store f32", plane, bytes);
  run<unsigned long long, 1, 8>("store u64", plane, bytes);
  run<float, 2, 8>("load + store f32", plane, bytes);
  run<float, 1, 3>("store f32", plane, bytes);
  CHECK(hipFree(plane));
  return 0;
}
