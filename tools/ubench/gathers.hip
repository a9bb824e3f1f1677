// What does a bilinear 2 x 2 gather of four planes cost on MI355X, by layout?  Every lane samples near its own pixel
// (neighbouring lanes -> neighbouring texels, as a coherent warp does); the planes fit the L2 / Infinity Cache.
//   planar : 4 planes x 2 rows x one 8-byte load  (what the pair kernels do on NCHW tensors)   8 loads, 64 B / lane
//   packed : texels of 4 floats, 2 rows x two 16-byte loads                                    4 loads, 64 B / lane
//   packed2: texels of 4 floats, 2 rows x one 32-byte access (the compiler splits it into two dwordx4)
//   dword  : 4 planes x 4 scalar loads                                                        16 loads, 64 B / lane
// Reported: ns per sampled pixel-lane-wave (one wave64 sampling one pixel per lane) per CU.
//   hipcc -O3 --offload-arch=gfx950 -o gathers gathers.hip && ./gathers
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int W = 832, H = 256, B = 12;
struct F2 { float a, b; };
struct alignas(16) F4 { float x, y, z, w; };

template <int MODE>
__global__ __launch_bounds__(256) void k(const float* __restrict__ planar, const F4* __restrict__ packed, float* out, int iters,
                                         int jitter) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x % B;
  int y = (blockIdx.x / B * 4 + wave) % (H - 1);
  int x0 = (blockIdx.y * 64) % (W - 64);
  float acc = 0.f;
  const size_t plane = (size_t)H * W;
  for (int it = 0; it < iters; ++it) {
    // a coherent warp: lane l samples around (x0 + l + shift, y + shift')
    const int x = min(max(x0 + lane + ((it * 7 + lane * jitter) & 3), 0), W - 2);
    const int yy = (y + it) % (H - 1);
    const size_t off = (size_t)yy * W + x;
    if (MODE == 0) {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* p = planar + ((size_t)b * 4 + c) * plane + off;
        const F2 n = *reinterpret_cast<const F2*>(p), s = *reinterpret_cast<const F2*>(p + W);
        acc += n.a + n.b + s.a + s.b;
      }
    } else if (MODE == 1 || MODE == 2) {
      const F4* p = packed + (size_t)b * plane + off;
      const F4 n0 = p[0], n1 = p[1], s0 = p[W], s1 = p[W + 1];
      acc += n0.x + n0.y + n0.z + n0.w + n1.x + n1.y + n1.z + n1.w + s0.x + s0.y + s0.z + s0.w + s1.x + s1.y + s1.z + s1.w;
    } else {
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const float* p = planar + ((size_t)b * 4 + c) * plane + off;
        acc += p[0] + p[1] + p[W] + p[W + 1];
      }
    }
  }
  if (acc == 12345.678f) out[0] = acc;
}

int main() {
  const size_t n = (size_t)B * 4 * H * W;
  float* planar; F4* packed; float* out;
  CHECK(hipMalloc(&planar, n * 4)); CHECK(hipMalloc(&packed, n * 4)); CHECK(hipMalloc(&out, 4));
  CHECK(hipMemset(planar, 0, n * 4)); CHECK(hipMemset(packed, 0, n * 4));
  hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  const int iters = 64;
  const char* names[] = {"planar 8 x dwordx2", "packed 4 x dwordx4", "packed (same)", "planar 16 x dword"};
  for (int jitter : {0, 1}) {
    for (int blocks_per_cu : {2, 4, 8}) {
      for (int mode : {0, 1, 3}) {
        const dim3 grid(256 * blocks_per_cu / 8, 8);
        auto go = [&] {
          if (mode == 0) hipLaunchKernelGGL((k<0>), grid, dim3(256), 0, 0, planar, packed, out, iters, jitter);
          else if (mode == 1) hipLaunchKernelGGL((k<1>), grid, dim3(256), 0, 0, planar, packed, out, iters, jitter);
          else hipLaunchKernelGGL((k<3>), grid, dim3(256), 0, 0, planar, packed, out, iters, jitter);
        };
        go(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); go(); go(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 2;
        const double wave_samples_per_cu = double(grid.x) * grid.y * 4 * iters / 256.0;
        printf("%-20s jitter %d  %d workgroups/CU : %.3f ms, %7.1f ns per wave-sample per CU, %6.1f GB/s per CU through L1\n", names[mode],
               jitter, blocks_per_cu, ms, ms * 1e6 / wave_samples_per_cu, 64.0 * 64 * wave_samples_per_cu / (ms * 1e6));
        fflush(stdout);
      }
    }
  }
  return 0;
}
