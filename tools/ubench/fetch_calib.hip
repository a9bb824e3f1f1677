// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 against KNOWN byte counts, in the access widths the pair
// kernels use (MI355X_MICROARCH.md, HBM section: FETCH_SIZE reports exactly half the bytes of a wide coalesced streaming
// read; other widths and WRITE_SIZE are uncalibrated -- calibrate in your own access pattern).  Every kernel touches a
// buffer of N bytes (default 1 GiB, four times the 256 MiB Infinity Cache) exactly once:
//   read_b32 / read_b64 / read_b128 : coalesced loads of 4 / 8 / 16 bytes per lane (the result is reduced into one word)
//   gather_b64                      : 8-byte loads at a lane-permuted offset inside a 4 KiB window (the 2 x 2 tap rows)
//   write_b32 / write_b128          : coalesced stores of 4 / 16 bytes per lane
//   atomic_f32                      : one global_atomic_add_f32 per dword (the scatter window's flush)
//     hipcc -O3 --offload-arch=gfx950 -o tools/ubench/fetch_calib tools/ubench/fetch_calib.hip
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace -d out -- tools/ubench/fetch_calib     (and a second pass: WRITE_SIZE)
// tools/pmc_calib_summary.py turns the two passes into profiles/r06_fetch_calibration.json.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

template <typename V>
__global__ void read_kernel(const V* __restrict__ p, size_t n, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const V v = p[i];
    const unsigned* w = reinterpret_cast<const unsigned*>(&v);
#pragma unroll
    for (unsigned j = 0; j < sizeof(V) / 4; ++j) acc ^= w[j];
  }
  if (acc == 0x12345678u) sink[0] = acc;  // (never true for the fill pattern: keeps the loads alive)
}
__global__ void gather_b64_kernel(const uint2* __restrict__ p, size_t n, unsigned* __restrict__ sink) {
  unsigned acc = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const size_t win = i & ~(size_t)511, k = i & 511;           // 512 x 8 B = a 4 KiB window
    const uint2 v = p[win + ((k * 37 + 11) & 511)];               // a permutation of the window: every element once
    acc ^= v.x ^ v.y;
  }
  if (acc == 0x12345678u) sink[0] = acc;
}
template <typename V>
__global__ void write_kernel(V* __restrict__ p, size_t n, V v) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void atomic_f32_kernel(float* __restrict__ p, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) atomicAdd(p + i, 1.0f);
}

int main(int argc, char** argv) {
  const size_t bytes = argc > 1 ? strtoull(argv[1], nullptr, 10) : (size_t)1 << 30;
  void* buf; unsigned* sink;
  if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMemset(buf, 0x5a, bytes);
  const dim3 grid(256 * 16), block(256);
  for (int rep = 0; rep < 3; ++rep) {
    read_kernel<unsigned><<<grid, block>>>((const unsigned*)buf, bytes / 4, sink);
    read_kernel<uint2><<<grid, block>>>((const uint2*)buf, bytes / 8, sink);
    read_kernel<uint4><<<grid, block>>>((const uint4*)buf, bytes / 16, sink);
    gather_b64_kernel<<<grid, block>>>((const uint2*)buf, bytes / 8, sink);
    write_kernel<unsigned><<<grid, block>>>((unsigned*)buf, bytes / 4, 0x5a5a5a5au);
    write_kernel<uint4><<<grid, block>>>((uint4*)buf, bytes / 16, make_uint4(0x5a5a5a5au, 0x5a5a5a5au, 0x5a5a5a5au, 0x5a5a5a5au));
    atomic_f32_kernel<<<grid, block>>>((float*)buf, bytes / 4);
    hipMemset(buf, 0x5a, bytes);
  }
  hipDeviceSynchronize();
  printf("{\"bytes_per_kernel\": %zu, \"reps\": 3}\n", bytes);
  return 0;
}
