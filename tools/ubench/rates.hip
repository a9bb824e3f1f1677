// Micro-benchmarks that price the instructions the pair kernel is made of (gfx950).
//   hipcc --offload-arch=gfx950 -O3 -o rates tools/ubench/rates.hip && ./rates
// Every VALU test issues 16 independent instructions per group, GROUPS groups per loop iteration; a launch puts
// exactly `wps` waves on every SIMD (256 CUs x wps workgroups of 256 threads) and the figure reported is
// shader cycles per wave-instruction per SIMD: elapsed cycles of a wave (s_memtime) / instructions it issued
// * waves per SIMD ... i.e. the issue interval the SIMD sustains.  Wall time is reported next to it.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

typedef float f2 __attribute__((ext_vector_type(2)));

constexpr int kIters = 2000;

#define R16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

// 16 float accumulators a[0..15], two float operands x, y (runtime, non-trivial)
#define DECL_F32 float a[16]; _Pragma("unroll") for (int i = 0; i < 16; ++i) a[i] = seed + i; float x = seed * 0.5f + 1.0f, y = seed * 0.25f;
#define SINK_F32 float s = 0; _Pragma("unroll") for (int i = 0; i < 16; ++i) s += a[i]; if (s == 123.456f) out[threadIdx.x] = s;

template <int ID>
__global__ __launch_bounds__(256) void valu_kernel(float* out, unsigned long long* cyc, float seed_in) {
  const float seed = seed_in + (threadIdx.x & 3);
  DECL_F32
  int ia[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) ia[i] = int(seed) + i * 7;
  int ix = int(seed) + 3, iy = 5;
  f2 p[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) p[i] = f2{seed + i, seed - i};
  f2 px = f2{x, y}, py = f2{y, x};
  double d[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) d[i] = seed + i;
  double dx = x;
  const float sx = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, seed_in * 1.5f)) ? seed_in * 1.5f : 0.f;
  unsigned long long mask = __builtin_amdgcn_read_exec() & 0x5555555555555555ull;
  unsigned long long sm[4] = {0, 0, 0, 0};
  int si[4] = {0, 0, 0, 0};
  asm volatile("v_cmp_lt_f32 vcc, %0, %1" : : "v"(a[0]), "v"(a[(threadIdx.x & 1) + 1]) : "vcc");
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
#define G4(B) B B B B
    if constexpr (ID == 0) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 1) {
#define OP(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 2) {
#define OP(i) asm volatile("v_subrev_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 3) {
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 4) {
#define OP(i) asm volatile("v_mul_f32 %0, 0.5, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 5) {
#define OP(i) asm volatile("v_mul_f32 %0, 0x3dcccccd, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 6) {
#define OP(i) asm volatile("v_mul_f32 %0, %1, %0" : "+v"(a[i]) : "s"(sx) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 7) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 8) {
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "s"(sx), "v"(y) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 9) {
#define OP(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 10) {
#define OP(i) asm volatile("v_fmaak_f32 %0, %0, %1, 0x3dcccccd" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 11) {
#define OP(i) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 12) {
#define OP(i) asm volatile("v_min_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 13) {
#define OP(i) asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 14) {
#define OP(i) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "v"(y) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 15) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1 clamp" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 16) {
#define OP(i) asm volatile("v_mul_f32 %0, |%0|, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 17) {
#define OP(i) asm volatile("v_mul_f32 %0, -%0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 18) {
#define OP(i) asm volatile("v_floor_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 19) {
#define OP(i) asm volatile("v_trunc_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 20) {
#define OP(i) asm volatile("v_rndne_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 21) {
#define OP(i) asm volatile("v_fract_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 22) {
#define OP(i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 23) {
#define OP(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 24) {
#define OP(i) asm volatile("v_sqrt_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 25) {
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 26) {
#define OP(i) asm volatile("v_cvt_i32_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 27) {
#define OP(i) asm volatile("v_cvt_f32_i32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 28) {
#define OP(i) asm volatile("v_cvt_u32_f32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 29) {
#define OP(i) asm volatile("v_cvt_f32_u32 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 30) {
#define OP(i) asm volatile("v_cvt_f32_ubyte0 %0, %0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 31) {
#define OP(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 32) {
#define OP(i) asm volatile("v_mov_b32 %0, 0" : "+v"(a[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 33) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 34) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 35) {
#define OP(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(x), "s"(mask) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 36) {
#define OP(i) asm volatile("v_cndmask_b32 %0, %1, %2, vcc" : "=v"(a[i]) : "v"(x), "v"(y) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 37) {
#define OP(i) asm volatile("v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]) :  : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 38) {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" :  : "v"(a[i]), "v"(x) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 39) {
#define OP(i) asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(sm[i & 3]) : "v"(a[i]), "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 40) {
#define OP(i) asm volatile("v_cmp_ge_i32 vcc, %0, %1" :  : "v"(ia[i]), "v"(ix) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 41) {
#define OP(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 42) {
#define OP(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 43) {
#define OP(i) asm volatile("v_sub_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 44) {
#define OP(i) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(ia[i]) : "v"(ix) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 45) {
#define OP(i) asm volatile("v_and_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 46) {
#define OP(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 47) {
#define OP(i) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 48) {
#define OP(i) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(ia[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 49) {
#define OP(i) asm volatile("v_lshrrev_b32 %0, 1, %0" : "+v"(ia[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 50) {
#define OP(i) asm volatile("v_ashrrev_i32 %0, 1, %0" : "+v"(ia[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 51) {
#define OP(i) asm volatile("v_lshl_add_u32 %0, %0, 2, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 52) {
#define OP(i) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ix), "v"(iy) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 53) {
#define OP(i) asm volatile("v_lshl_or_b32 %0, %0, 2, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 54) {
#define OP(i) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ix), "v"(iy) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 55) {
#define OP(i) asm volatile("v_bfe_u32 %0, %0, 3, 8" : "+v"(ia[i]) :  : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 56) {
#define OP(i) asm volatile("v_min_i32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 57) {
#define OP(i) asm volatile("v_max_i32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 58) {
#define OP(i) asm volatile("v_min_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 59) {
#define OP(i) asm volatile("v_med3_i32 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ix), "v"(iy) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 60) {
#define OP(i) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 61) {
#define OP(i) asm volatile("v_mul_i32_i24 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 62) {
#define OP(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ix), "v"(iy) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 63) {
#define OP(i) asm volatile("v_mad_i32_i24 %0, %0, %1, %2" : "+v"(ia[i]) : "v"(ix), "v"(iy) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 64) {
#define OP(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(ia[i]) : "v"(ix) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 65) {
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(px), "v"(py) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 66) {
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 67) {
#define OP(i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(px) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 68) {
#define OP(i) asm volatile("v_pk_mov_b32 %0, %1, %1" : "+v"(p[i]) : "v"(px) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 69) {
#define OP(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(d[i & 7]) : "v"(dx) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 70) {
#define OP(i) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 71) {
#define OP(i) asm volatile("v_add_f32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 72) {
#define OP(i) asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(si[i & 3]) : "v"(a[i]) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 73) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_max_f32 %0, %0, %1" : "+v"(a[i]) : "v"(x) : );
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 74) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      G4(R16(OP))
#undef OP
    }
    if constexpr (ID == 75) {
#define OP(i) asm volatile("v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_add_f32 %0, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i]) : "v"(x) : "vcc");
      G4(R16(OP))
#undef OP
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  SINK_F32
  int is = 0; f2 ps = f2{0, 0}; double ds = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { is += ia[i]; ps += p[i]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) ds += d[i];
  is += int(sm[0] + sm[1] + sm[2] + sm[3]) + si[0] + si[1] + si[2] + si[3];
  if (is == 0x7fffffff || ps.x == 123.456f || ds == 123.456) out[threadIdx.x] = float(is) + ps.y + float(ds);
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ------------------------------------------------------------------------------------------------------
// LDS instruction rates (conflict-free row access: lane l touches dword l of a row).
template <int ID>
__global__ __launch_bounds__(256) void lds_kernel(float* out, unsigned long long* cyc, float seed) {
  __shared__ float buf[256 * 2 * 20];
  for (int i = threadIdx.x; i < 256 * 2 * 20; i += 256) buf[i] = seed + i;
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0; f2 acc2 = f2{0, 0};
  const unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < kIters; ++it) {
    if constexpr (ID == 0) {  // ds_read_b32, 16 per iteration
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += ((volatile float*)buf)[(wave * 16 + r) * 64 + lane];
    } else if constexpr (ID == 1) {  // ds_read_b64
#pragma unroll
      for (int r = 0; r < 16; ++r) { f2 v = ((volatile f2*)buf)[(wave * 16 + r) * 64 + lane]; acc2 += v; }
    } else if constexpr (ID == 2) {  // ds_write_b32
#pragma unroll
      for (int r = 0; r < 16; ++r) ((volatile float*)buf)[(wave * 16 + r) * 64 + lane] = acc + r;
    } else if constexpr (ID == 3) {  // ds_write_b64
#pragma unroll
      for (int r = 0; r < 16; ++r) ((volatile f2*)buf)[(wave * 16 + r) * 64 + lane] = f2{acc + r, acc};
    } else if constexpr (ID == 4) {  // ds_add_f32 (no return), distinct addresses per lane
#pragma unroll
      for (int r = 0; r < 16; ++r)
        (void)__hip_atomic_fetch_add(&buf[(wave * 16 + r) * 64 + lane], seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if constexpr (ID == 5) {  // ds_add_f32, neighbouring lanes hit the same cell pairwise (lane/2)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        (void)__hip_atomic_fetch_add(&buf[(wave * 16 + r) * 64 + (lane >> 1)], seed, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    } else if constexpr (ID == 6) {  // ds_bpermute
#pragma unroll
      for (int r = 0; r < 16; ++r) acc += __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(((lane + r + 1) & 63) << 2, __builtin_bit_cast(int, acc)));
    }
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  if (acc == 123.456f || acc2.x == 123.456f) out[threadIdx.x] = acc + acc2.y;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

// ------------------------------------------------------------------------------------------------------
// Global fp32 atomics, row-coalesced (lane l -> consecutive dword l of a row), each workgroup on its own region
// of a plane that fits the L2s / infinity cache.  mode 0: atomicAdd no return; 1: plain store; 2: load+store RMW
template <int MODE>
__global__ __launch_bounds__(256) void gatomic_kernel(float* plane, size_t plane_elems, int rows_per_block, int row_pitch, float v) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const size_t base = ((size_t)blockIdx.x * rows_per_block) * row_pitch % plane_elems;
  for (int r = wave; r < rows_per_block; r += 4) {
    float* q = plane + (base + (size_t)r * row_pitch + lane) % plane_elems;
    if (MODE == 0) atomicAdd(q, v);
    else if (MODE == 1) *q = v;
    else *q += v;
  }
}

// ------------------------------------------------------------------------------------------------------
// Buffer-load bounds semantics: which (row, col) of a [H][W] fp32 plane read back as 0 through a descriptor with
// stride = W * 4 (index = row via idxen, byte offset = col * 4 via offen).
__global__ void bufsem_raw_kernel(const float* plane, int nbytes, const int* offs, float* res, float* res2, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)plane, 0, nbytes, 0x00020000);
  res[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, offs[i], 0, 0));
  typedef int i2 __attribute__((ext_vector_type(2)));
  i2 v = __builtin_amdgcn_raw_buffer_load_b64(rs, offs[i], 0, 0);
  res2[2 * i] = __builtin_bit_cast(float, v.x);
  res2[2 * i + 1] = __builtin_bit_cast(float, v.y);
}

// structured variant through inline asm (idxen + offen): vaddr = {index, offset}
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i2v __attribute__((ext_vector_type(2)));
__global__ void bufsem_struct_kernel(const float* plane, int H, int W, const int* ys, const int* xs, float* res, float* res2, int n) {
  const int i = threadIdx.x;
  if (i >= n) return;
  const unsigned long long p = (unsigned long long)plane;
  i4 rs;
  rs.x = __builtin_amdgcn_readfirstlane((int)(p & 0xffffffffu));
  rs.y = __builtin_amdgcn_readfirstlane((int)((p >> 32) & 0xffffu) | ((W * 4) << 16));
  rs.z = __builtin_amdgcn_readfirstlane(H);
  rs.w = __builtin_amdgcn_readfirstlane(0x00020000);
  i2v va; va.x = ys[i]; va.y = xs[i] * 4;
  int r; i2v r2v;
  asm volatile("buffer_load_dword %0, %1, %2, 0 idxen offen\n s_waitcnt vmcnt(0)" : "=v"(r) : "v"(va), "s"(rs) : "memory");
  asm volatile("buffer_load_dwordx2 %0, %1, %2, 0 idxen offen\n s_waitcnt vmcnt(0)" : "=v"(r2v) : "v"(va), "s"(rs) : "memory");
  res[i] = __builtin_bit_cast(float, r);
  res2[2 * i] = __builtin_bit_cast(float, r2v.x);
  res2[2 * i + 1] = __builtin_bit_cast(float, r2v.y);
}

template <typename F>
static void time_launch(const char* name, int wps, double insts_per_wave, F launch, unsigned long long* d_cyc) {
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
  launch();  // warm
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(e0));
  launch();
  CHECK(hipEventRecord(e1));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  unsigned long long cyc = 0;
  CHECK(hipMemcpy(&cyc, d_cyc, sizeof(cyc), hipMemcpyDeviceToHost));
  const double per_inst_wave = double(cyc) / insts_per_wave;            // cycles a wave spends per instruction
  const double issue = per_inst_wave / wps;                              // issue interval of the SIMD
  const double wall_ns_per_inst_simd = 1e6 * ms / (insts_per_wave * wps);
  printf("%-28s wps=%d  memtime/inst/wave=%7.2f  -> per SIMD %6.2f ticks ; wall %6.3f ns per inst per SIMD (%.3f ms)\n", name, wps,
         per_inst_wave, issue, wall_ns_per_inst_simd, ms);
  fflush(stdout);
}

int main(int argc, char** argv) {
  float* d_out; unsigned long long* d_cyc;
  CHECK(hipMalloc(&d_out, 4096)); CHECK(hipMalloc(&d_cyc, 8));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  printf("device %s CUs=%d clock=%d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate);
  const int CU = prop.multiProcessorCount;
  struct VT { const char* name; int n; };
  const VT vt[] = {{"v_add_f32", 1}, {"v_sub_f32", 1}, {"v_subrev_f32", 1}, {"v_mul_f32", 1}, {"v_mul_f32 const", 1}, {"v_mul_f32 literal", 1}, {"v_mul_f32 sgpr", 1}, {"v_fma_f32", 1}, {"v_fma_f32 sgpr", 1}, {"v_fmac_f32", 1}, {"v_mad_f32?fmaak", 1}, {"v_max_f32", 1}, {"v_min_f32", 1}, {"v_med3_f32", 1}, {"v_max3_f32", 1}, {"v_add_f32 clamp", 1}, {"v_mul_f32 abs", 1}, {"v_mul_f32 neg(vop3)", 1}, {"v_floor_f32", 1}, {"v_trunc_f32", 1}, {"v_rndne_f32", 1}, {"v_fract_f32", 1}, {"v_rcp_f32", 1}, {"v_rsq_f32", 1}, {"v_sqrt_f32", 1}, {"v_exp_f32", 1}, {"v_cvt_i32_f32", 1}, {"v_cvt_f32_i32", 1}, {"v_cvt_u32_f32", 1}, {"v_cvt_f32_u32", 1}, {"v_cvt_f32_ubyte0", 1}, {"v_mov_b32", 1}, {"v_mov_b32 const", 1}, {"v_cndmask vcc(undef)", 1}, {"v_cndmask vcc set", 1}, {"v_cndmask e64 sgpr", 1}, {"v_cndmask dst!=src", 1}, {"v_cndmask const,v", 1}, {"v_cmp_lt_f32 vcc", 1}, {"v_cmp_lt_f32 e64 sgpr", 1}, {"v_cmp_class?ge_i32", 1}, {"cmp+cndmask pair", 2}, {"v_add_u32", 1}, {"v_sub_u32", 1}, {"v_add_co_u32", 1}, {"v_and_b32", 1}, {"v_or_b32", 1}, {"v_xor_b32", 1}, {"v_lshlrev_b32", 1}, {"v_lshrrev_b32", 1}, {"v_ashrrev_i32", 1}, {"v_lshl_add_u32", 1}, {"v_add3_u32", 1}, {"v_lshl_or_b32", 1}, {"v_and_or_b32", 1}, {"v_bfe_u32", 1}, {"v_min_i32", 1}, {"v_max_i32", 1}, {"v_min_u32", 1}, {"v_med3_i32", 1}, {"v_mul_u32_u24", 1}, {"v_mul_i32_i24", 1}, {"v_mad_u32_u24", 1}, {"v_mad_i32_i24", 1}, {"v_mul_lo_u32", 1}, {"v_pk_fma_f32", 1}, {"v_pk_mul_f32", 1}, {"v_pk_add_f32", 1}, {"v_pk_mov_b32", 1}, {"v_add_f64", 1}, {"v_mov_dpp row_shr", 1}, {"v_add_f32_dpp", 1}, {"v_readlane->s", 1}, {"mix: 3 add + 1 max", 4}, {"mix: 3 add + 1 cndm", 4}, {"mix: 7 add + 1 cndm", 8}};
  const double ipw = double(kIters) * 64;
#define RUN(ID) for (int wps : {1, 2}) { time_launch(vt[ID].name, wps, ipw * vt[ID].n, [&] { hipLaunchKernelGGL((valu_kernel<ID>), dim3(CU * wps), dim3(256), 0, 0, d_out, d_cyc, 1.0f); }, d_cyc); }
  if (!(argc > 1 && !strcmp(argv[1], "novalu"))) {
  RUN(0)
  RUN(1)
  RUN(2)
  RUN(3)
  RUN(4)
  RUN(5)
  RUN(6)
  RUN(7)
  RUN(8)
  RUN(9)
  RUN(10)
  RUN(11)
  RUN(12)
  RUN(13)
  RUN(14)
  RUN(15)
  RUN(16)
  RUN(17)
  RUN(18)
  RUN(19)
  RUN(20)
  RUN(21)
  RUN(22)
  RUN(23)
  RUN(24)
  RUN(25)
  RUN(26)
  RUN(27)
  RUN(28)
  RUN(29)
  RUN(30)
  RUN(31)
  RUN(32)
  RUN(33)
  RUN(34)
  RUN(35)
  RUN(36)
  RUN(37)
  RUN(38)
  RUN(39)
  RUN(40)
  RUN(41)
  RUN(42)
  RUN(43)
  RUN(44)
  RUN(45)
  RUN(46)
  RUN(47)
  RUN(48)
  RUN(49)
  RUN(50)
  RUN(51)
  RUN(52)
  RUN(53)
  RUN(54)
  RUN(55)
  RUN(56)
  RUN(57)
  RUN(58)
  RUN(59)
  RUN(60)
  RUN(61)
  RUN(62)
  RUN(63)
  RUN(64)
  RUN(65)
  RUN(66)
  RUN(67)
  RUN(68)
  RUN(69)
  RUN(70)
  RUN(71)
  RUN(72)
  RUN(73)
  RUN(74)
  RUN(75)
  }
  const char* lnames[] = {"ds_read_b32", "ds_read_b64", "ds_write_b32", "ds_write_b64", "ds_add_f32", "ds_add_f32 2-way same", "ds_bpermute"};
#define RUNL(ID) for (int wps : {1, 2, 3}) { \
    time_launch(lnames[ID], wps, double(kIters) * 16, [&] { hipLaunchKernelGGL((lds_kernel<ID>), dim3(CU * wps), dim3(256), 0, 0, d_out, d_cyc, 1.0f); }, d_cyc); }
  RUNL(0) RUNL(1) RUNL(2) RUNL(3) RUNL(4) RUNL(5) RUNL(6)

  // global atomics: 40 MB plane (L2 / MALL resident), each block walks `rows` rows of 64 floats
  {
    const size_t elems = 10u << 20;
    float* plane; CHECK(hipMalloc(&plane, elems * 4)); CHECK(hipMemset(plane, 0, elems * 4));
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int pitch : {64, 832}) {
      for (int mode = 0; mode < 3; ++mode) {
        const int blocks = 16384, rows = 32;
        auto go = [&] {
          if (mode == 0) hipLaunchKernelGGL((gatomic_kernel<0>), dim3(blocks), dim3(256), 0, 0, plane, elems, rows, pitch, 1.0f);
          else if (mode == 1) hipLaunchKernelGGL((gatomic_kernel<1>), dim3(blocks), dim3(256), 0, 0, plane, elems, rows, pitch, 1.0f);
          else hipLaunchKernelGGL((gatomic_kernel<2>), dim3(blocks), dim3(256), 0, 0, plane, elems, rows, pitch, 1.0f);
        };
        go(); CHECK(hipDeviceSynchronize());
        CHECK(hipEventRecord(e0)); go(); CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double n = double(blocks) * rows * 64;
        printf("global %-10s pitch=%4d : %.1f M lane-ops in %.3f ms = %.2f G/s (%.1f GB/s of fp32)\n",
               mode == 0 ? "atomicAdd" : (mode == 1 ? "store" : "load+store"), pitch, n * 1e-6, ms, n / ms * 1e-6, 4 * n / ms * 1e-6);
        fflush(stdout);
      }
    }
    CHECK(hipFree(plane));
  }

  // buffer load bounds semantics
  {
    const int H = 6, W = 10;
    std::vector<float> h(H * W);
    for (int i = 0; i < H * W; ++i) h[i] = 100.f + i;
    float* plane; CHECK(hipMalloc(&plane, (H * W + 64) * 4));
    std::vector<float> big(H * W + 64, 7777.f);
    memcpy(big.data(), h.data(), H * W * 4);
    CHECK(hipMemcpy(plane, big.data(), big.size() * 4, hipMemcpyHostToDevice));
    std::vector<int> ys = {0, 0, 0, 0, 2, 2, 2, 5, 5, 6, -1, 3, 3}, xs = {0, 9, 10, -1, 8, 9, 10, 9, 10, 0, 0, -1, -2};
    const int n = (int)ys.size();
    int *dy, *dx; float *r1, *r2;
    CHECK(hipMalloc(&dy, n * 4)); CHECK(hipMalloc(&dx, n * 4)); CHECK(hipMalloc(&r1, n * 4)); CHECK(hipMalloc(&r2, n * 8));
    CHECK(hipMemcpy(dy, ys.data(), n * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dx, xs.data(), n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(bufsem_struct_kernel, dim3(1), dim3(64), 0, 0, plane, H, W, dy, dx, r1, r2, n);
    std::vector<float> o1(n), o2(2 * n);
    CHECK(hipMemcpy(o1.data(), r1, n * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o2.data(), r2, n * 8, hipMemcpyDeviceToHost));
    printf("struct buffer (stride=W*4, records=H=%d, W=%d); plane[y][x] = 100 + y*W + x; beyond the plane 7777\n", H, W);
    for (int i = 0; i < n; ++i) printf("  y=%2d x=%2d : b32 -> %7.1f   b64 -> (%7.1f, %7.1f)\n", ys[i], xs[i], o1[i], o2[2 * i], o2[2 * i + 1]);
    std::vector<int> offs = {0, 4, (H * W - 1) * 4, H * W * 4, H * W * 4 - 4, -4, -8, H * W * 4 + 4};
    const int m = (int)offs.size();
    int* doff; CHECK(hipMalloc(&doff, m * 4)); CHECK(hipMemcpy(doff, offs.data(), m * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(bufsem_raw_kernel, dim3(1), dim3(64), 0, 0, plane, H * W * 4, doff, r1, r2, m);
    CHECK(hipMemcpy(o1.data(), r1, m * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(o2.data(), r2, m * 8, hipMemcpyDeviceToHost));
    printf("raw buffer (num_records = %d bytes)\n", H * W * 4);
    for (int i = 0; i < m; ++i) printf("  byte off=%4d : b32 -> %7.1f   b64 -> (%7.1f, %7.1f)\n", offs[i], o1[i], o2[2 * i], o2[2 * i + 1]);
  }
  return 0;
}
