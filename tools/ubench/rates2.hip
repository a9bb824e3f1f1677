// Second micro-benchmark: wave-wide DPP shifts, LDS throughput (reads / writes / float and integer atomics with
// moving addresses), as the pair kernel would use them.   hipcc --offload-arch=gfx950 -O3 -o rates2 rates2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 1000;

template <int ID>
__global__ __launch_bounds__(256) void dpp_kernel(float* out, float seed_in) {
  float a[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) a[i] = seed_in + threadIdx.x + i;
  float x = seed_in * 0.5f;
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        if constexpr (ID == 0) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
        if constexpr (ID == 1) asm volatile("v_add_f32_dpp %0, %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
        if constexpr (ID == 2) asm volatile("v_add_f32_dpp %0, %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
        if constexpr (ID == 3) asm volatile("v_fmac_f32_dpp %0, %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x));
        if constexpr (ID == 4) asm volatile("v_mov_b32_dpp %0, %0 wave_ror:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
        if constexpr (ID == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x));
      }
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += a[i];
  if (s == 123.456f) out[threadIdx.x] = s;
}

// semantic check of the wave shifts: lane l receives lane l-1 (shr) / l+1 (shl); what do lanes 0 / 63 get?
__global__ void dpp_sem_kernel(float* out) {
  const float v = 100.0f + threadIdx.x;
  float r0 = -1.0f, r1 = -1.0f, r2 = -1.0f, r3 = -1.0f;
  asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(r0) : "v"(v));
  asm volatile("v_mov_b32_dpp %0, %1 wave_shl:1 row_mask:0xf bank_mask:0xf" : "+v"(r1) : "v"(v));
  asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "+v"(r2) : "v"(v));
  r3 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, -2.0f), __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, false));
  out[threadIdx.x] = r0; out[64 + threadIdx.x] = r1; out[128 + threadIdx.x] = r2; out[192 + threadIdx.x] = r3;
}

// LDS: every wave works on its own 16 rows x 64 dwords region; addresses move with the iteration so that no
// instruction hits the cell of the one before it.
template <int ID>
__global__ __launch_bounds__(256) void lds_kernel(float* out, float seed) {
  __shared__ __attribute__((aligned(16))) float buf[4 * 2048 + 64];
  for (int i = threadIdx.x; i < 4 * 2048 + 64; i += 256) buf[i] = 0.0f;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float* base = buf + wave * 2048;
  unsigned* ubase = reinterpret_cast<unsigned*>(base);
  unsigned long long* lbase = reinterpret_cast<unsigned long long*>(base);
  float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  f2 acc2[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) acc2[i] = f2{0, 0};
  for (int it = 0; it < kIters; ++it) {
    const int rot = (it * 7) & 15;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r + rot) & 15;
      if constexpr (ID == 0) acc[r & 7] += base[row * 64 + lane];                                        // ds_read_b32
      if constexpr (ID == 1) acc2[r & 7] += reinterpret_cast<f2*>(base)[row * 64 + lane];               // ds_read_b64 (1024 f2 per wave)
      if constexpr (ID == 2) base[row * 64 + lane] = seed + r;                                          // ds_write_b32
      if constexpr (ID == 3) reinterpret_cast<f2*>(base)[row * 64 + lane] = f2{seed, seed + r};          // ds_write_b64
      if constexpr (ID == 4) (void)__hip_atomic_fetch_add(&base[row * 64 + lane], seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if constexpr (ID == 5) (void)__hip_atomic_fetch_add(&ubase[row * 64 + lane], 3u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if constexpr (ID == 6) (void)__hip_atomic_fetch_add(&lbase[(row * 64 + lane) & 1023], 3ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      // float atomics with a stride of 2 dwords (half the banks), and with half of the lanes masked off
      if constexpr (ID == 7) (void)__hip_atomic_fetch_add(&base[(row * 64 + lane * 2) & 2047], seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      if constexpr (ID == 8) { if (lane & 1) (void)__hip_atomic_fetch_add(&base[row * 64 + lane], seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
      if constexpr (ID == 9) { if (lane < 8) (void)__hip_atomic_fetch_add(&base[row * 64 + lane], seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
    }
    if constexpr (ID == 2 || ID == 3) asm volatile("" ::: "memory");
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += acc[i] + acc2[i].x + acc2[i].y;
  __syncthreads();
  s += buf[threadIdx.x];
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <typename F>
static void run(const char* name, int wps, double insts_per_wave, F launch) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch(); CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0)); launch(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
  float ms = 0; CHECK(hipEventElapsedTime(&ms, e0, e1));
  // per CU: 4 * wps waves, each issuing insts_per_wave instructions
  printf("%-34s wps=%d : %7.3f ns per wave-instruction per SIMD, %7.3f ns per wave-instruction per CU (%.3f ms)\n", name, wps,
         1e6 * ms / (insts_per_wave * wps), 1e6 * ms / (insts_per_wave * wps * 4), ms);
  fflush(stdout);
}

int main() {
  float* d_out; CHECK(hipMalloc(&d_out, 4096));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int CU = prop.multiProcessorCount;
  {
    hipLaunchKernelGGL(dpp_sem_kernel, dim3(1), dim3(64), 0, 0, d_out);
    float h[256]; CHECK(hipMemcpy(h, d_out, sizeof(h), hipMemcpyDeviceToHost));
    const char* nm[4] = {"wave_shr:1 (old=-1)", "wave_shl:1 (old=-1)", "wave_shr:1 bound_ctrl:0", "update_dpp 0x130 (old=-2)"};
    for (int k = 0; k < 4; ++k) {
      printf("%-28s lanes 0,1,2,15,16,17,31,32,33,62,63 -> ", nm[k]);
      for (int l : {0, 1, 2, 15, 16, 17, 31, 32, 33, 62, 63}) printf("%6.1f ", h[64 * k + l]);
      printf("\n");
    }
  }
  const char* dn[] = {"v_add_f32_dpp wave_shr:1", "v_add_f32_dpp wave_shl:1", "v_add_f32_dpp row_shr:1", "v_fmac_f32_dpp wave_shr:1", "v_mov_dpp wave_ror:1", "v_add_f32 (plain)"};
#define RD(ID) for (int wps : {1, 2, 4}) run(dn[ID], wps, double(kIters) * 64, [&] { hipLaunchKernelGGL((dpp_kernel<ID>), dim3(CU * wps), dim3(256), 0, 0, d_out, 1.0f); });
  RD(0) RD(1) RD(2) RD(3) RD(4) RD(5)
  const char* ln[] = {"ds_read_b32", "ds_read_b64", "ds_write_b32", "ds_write_b64", "ds_add_f32", "ds_add_u32", "ds_add_u64",
                      "ds_add_f32 stride 2", "ds_add_f32 odd lanes only", "ds_add_f32 8 lanes only"};
#define RL(ID) for (int wps : {1, 2, 3}) run(ln[ID], wps, double(kIters) * 16, [&] { hipLaunchKernelGGL((lds_kernel<ID>), dim3(CU * wps), dim3(256), 0, 0, d_out, 1.0f); });
  RL(0) RL(1) RL(2) RL(3) RL(4) RL(5) RL(6) RL(7) RL(8) RL(9)
  return 0;
}
