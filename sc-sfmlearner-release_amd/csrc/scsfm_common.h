// Shared device-side definitions for the SC-SfMLearner warp + loss kernels (gfx950).
//
// Math spec: SURVEY.md §9; reference lines are cited next to each function.  Everything is
// templated on the scalar type T: float is the product path, double exists for the
// gradient-check tests (exported as the *_f64 entry points).
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/scsfm_hip.h"

namespace scsfm {

constexpr int kWave = 64;        // CDNA wavefront
constexpr int kThreads = 256;    // 4 waves per workgroup, one per SIMD
constexpr int kTileW = 64;       // one wave covers one 64-pixel row segment: 256 B coalesced rows
constexpr int kTileH = 16;       // each thread owns a 4-row column strip
constexpr int kStrip = 4;
constexpr int kHaloW = kTileW + 2;
constexpr int kHaloH = kTileH + 2;
// backward: the 64x16 compute domain overlaps its neighbours by one pixel on every side, so the
// interior that a block writes is 62x14 (SSIM backward needs the forward statistics of the ring).
constexpr int kBwdOutW = kTileW - 2;
constexpr int kBwdOutH = kTileH - 2;

constexpr double kSsimC1 = 0.01 * 0.01;  // loss_functions.py:25
constexpr double kSsimC2 = 0.03 * 0.03;  // loss_functions.py:26
constexpr double kMaskGate = 10000.0;    // loss_functions.py:125
constexpr double kZMin = 1e-3;           // inverse_warp.py:211

// Per batch element constants written by prep_kernel: K^-1 (inverse_warp.py:253), A|c = K @ [R|t]
// (inverse_warp.py:258-260), and their product M = A K^-1: pixel2cam followed by the pose transform and the
// intrinsics is P = depth * M (u, v, 1) + c, and M (u, v, 1) splits into a column part a thread evaluates once for
// its strip and a row part of three FMAs per pixel.  32 scalars so that consecutive elements stay 32/64-byte
// aligned and a block can fetch its element with scalar loads (the address is workgroup-uniform).
template <typename T>
struct BatchConsts {
  T Kinv[9];
  T A[9];
  T c[3];
  T M[9];  // A K^-1: pixel (u, v, 1) * depth -> projective coordinates of the other view, minus c
  T pad[2];
};

// Workspace layout of one pair-direction call (scsfm_pair_ws_bytes).
//   [0]                consts   : B x BatchConsts<double>-sized slots (T = float uses the front)
//   [off_sums]         sums     : double[16] = {S_photo, S_geom, S_m, photo, geom, a, b, -, spec, w_photo, w_geom, ...}
//                                 a = d photo / d sum, b = d geom / d sum (0 when gated off); spec = 1 if the
//                                 forward was speculative for upstream weights (w_photo, w_geom)
//   [off_gP]           gPp      : double[B][blocks per image][12]  (per-block partial gradients of A|c, written by
//                                 the speculative forward's geometry tail or by the geometry pass;
//                                 reduced by pose_reduce_bwd_kernel -- same-address atomics from ~200 blocks per
//                                 image cost 60 us per launch, partials cost nothing)
//   [off_partials]     partials : double[nblocks][3]
//   [off_smooth]       smooth   : double[nblocks][3] = {sum D, Sx, Sy} of the target frame's smooth loss per tile (only
//                                 written by a speculative forward whose descriptor names a smooth workspace)
struct PairWs {
  size_t off_sums, off_gP, off_partials, off_smooth, total;
  int nbx, nby;
};

// hipGetLastError() reports the last error of ANY earlier runtime call of this thread (observed: a
// stale hipErrorNoDevice left behind before the first launch), so entry points clear it first and
// only report what their own launches produced.
inline void clear_status() { (void)hipGetLastError(); }
inline int launch_status() { return (int)hipGetLastError(); }

__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// What every entry point that takes (B, H, W) checks before it launches anything (SCSFM_ERR_ARG otherwise): the kernels
// address the three colour planes of ONE image of the batch with 32-bit byte offsets from a wave-uniform base, and the
// batched launches put (pair or frame, batch element) on the grid's z axis (<= 65535; up to kMaxPairs = 8 per launch).
template <typename T>
inline bool dims_ok(int B, int H, int W) {
  return B > 0 && B <= 8191 && H >= 1 && W >= 1 && (unsigned long long)3 * (unsigned long long)H * (unsigned long long)W * sizeof(T) < (1ull << 32);
}

inline PairWs pair_ws_layout(int B, int H, int W) {
  PairWs l;
  l.nbx = ceil_div(W, kTileW);
  l.nby = ceil_div(H, kTileH);
  size_t off = (size_t)B * sizeof(BatchConsts<double>);
  l.off_sums = off; off += 16 * sizeof(double);
  // (sized like the partials: the finest tiling that writes them is the fp64 speculative forward's)
  l.off_gP = off; off += (size_t)B * ceil_div(W, kTileW - 4) * ceil_div(H, 6) * 12 * sizeof(double);
  // partials: sized for the finest tiling that writes them (at most 62 x 6 outputs per block of the tiled kernels,
  // 60-column strips of the speculative forward)
  l.off_partials = off; off += (size_t)ceil_div(W, kTileW - 4) * ceil_div(H, 6) * B * 3 * sizeof(double);
  // the target frame's smooth-loss partials of a pair that carries them (scsfm_pair_desc::smooth_ws): one record per tile
  l.off_smooth = off; off += (size_t)ceil_div(W, kTileW - 4) * ceil_div(H, 6) * B * 3 * sizeof(double);
  l.total = (off + 255) & ~(size_t)255;
  return l;
}

// ------------------------------------------------------------------------------------------
// Wave / block reductions (wave = 64 lanes).
// ------------------------------------------------------------------------------------------
// Sum over the 64 lanes of a wave.  Within a wave the working type is kept (fp32 on the product
// path: 64 addends, ~1e-7 relative); across waves and blocks everything is carried in fp64.
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}

// Wave sums of N values per lane with ~N + log2(64) shuffles instead of 6 N: at every butterfly step the
// values are paired and each half of the exchanging lanes keeps one value of a pair (an odd value out
// does the plain butterfly), so the live values halve as the lane sets do.  Afterwards v[0] of a lane
// holds the complete sum of ONE of the original values: its index is returned, and `lead` says whether
// this lane is the one that should publish it.  Every value meets exactly the addends of wave_sum()
// in the same pairing, so the sums are bit-identical to wave_sum()'s.
template <int N, typename T>
__device__ __forceinline__ int wave_sum_packed(T (&v)[N], bool& lead) {
  const int lane = threadIdx.x & (kWave - 1);
  int n = N;
#pragma unroll
  for (int o = kWave / 2; o > 0; o >>= 1) {
    const bool up = (lane & o) != 0;
#pragma unroll
    for (int j = 0; j < n / 2; ++j) {
      const T send = up ? v[2 * j] : v[2 * j + 1], keep = up ? v[2 * j + 1] : v[2 * j];
      v[j] = keep + __shfl_xor(send, o);
    }
    if (n & 1) v[n / 2] = v[n - 1] + __shfl_xor(v[n - 1], o);
    n = (n + 1) / 2;
  }
  // walk the steps backwards from slot 0 to the original index; steps that did not split leave their
  // lane bit free (all those lanes hold the same sum)
  int cnt[7];
  cnt[0] = N;
#pragma unroll
  for (int s = 0; s < 6; ++s) cnt[s + 1] = (cnt[s] + 1) / 2;
  int slot = 0, free_bits = 0;
#pragma unroll
  for (int s = 5; s >= 0; --s) {
    const int o = (kWave / 2) >> s, half = cnt[s] / 2;
    if (slot < half) slot = 2 * slot + ((lane & o) ? 1 : 0);  // came from a pair: the lane bit says which
    else { slot = cnt[s] - 1; free_bits |= o; }                // the odd value out: both lane halves hold it
  }
  lead = (lane & free_bits) == 0;
  return slot;
}

// Sum N values per thread over the whole block; the result replaces v[] in thread 0 (other threads
// keep partial values).  `scratch` must hold N * (kThreads / kWave) doubles.
template <int N, typename T>
__device__ __forceinline__ void block_sum(T (&v)[N], double* scratch) {
  const int wave = threadIdx.x / kWave;
  bool lead;
  const int idx = wave_sum_packed<N>(v, lead);
  if (lead) scratch[wave * N + idx] = double(v[0]);
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < N; ++i) {
      double s = 0;
      for (int w = 0; w < kThreads / kWave; ++w) s += scratch[w * N + i];
      v[i] = T(s);
    }
  }
}

// The same sums, stored: out[i] (fp64) = sum over the block of v[i].  After the waves' packed sums meet in `scratch`, lane i
// of the first wave adds up value i and writes it -- N threads with kThreads / kWave additions each instead of one
// thread with N of them (the fp64 additions of a serial tail sat on wave 0 of every workgroup: 60 for N = 12).
template <int N, typename T>
__device__ __forceinline__ void block_sum_store(T (&v)[N], double* scratch, double* __restrict__ out) {
  const int wave = threadIdx.x / kWave;
  bool lead;
  const int idx = wave_sum_packed<N>(v, lead);
  if (lead) scratch[wave * N + idx] = double(v[0]);
  __syncthreads();
  if (threadIdx.x < N) {
    double s = 0;
#pragma unroll
    for (int w = 0; w < kThreads / kWave; ++w) s += scratch[w * N + threadIdx.x];
    out[threadIdx.x] = s;
  }
}

// XCD-aware block order.  The dispatcher hands consecutive workgroups to the 8 XCDs round-robin (observed,
// not contractual: used for speed only), so with the natural order horizontally adjacent tiles -- which
// share ring pixels and gather from the same neighbourhood -- sit on different, mutually non-coherent L2s.
// This bijection gives XCD k the k-th contiguous eighth of the logical (x fastest, then y, then z) order,
// i.e. whole images per XCD.
struct BlockId { int x, y, z; };
// p = linear id of a workgroup that the dispatcher placed on XCD p % 8 -> the logical tile it should process
// Chunked (round 4): the logical order is cut into chunks of C tiles and chunk j goes to XCD j % 8, so that an XCD works on
// every eighth chunk instead of on one contiguous eighth of the launch: with C = 266 -- one 256 x 832 image -- the images of
// a pair-direction, whose tiles cost alike, are spread over all XCDs and none of them is left working alone at the end
// (-1.5 % on the speculative forward; the locality that matters, between neighbouring tiles of an image, is kept; other
// image sizes get chunks of a few whole or fractional images, which is as good).  C is a compile-time constant on purpose:
// derived from the grid (nx * ny) it costs a scalar division and two live scalars, and the speculative forward sits at
// the limit of its 102 scalar registers -- the allocator then moves the images' buffer descriptors into vector
// registers and every load through them becomes a waterfall loop (+35 % vector instructions, measured in round 4).
// SCSFM_XCD_CHUNK (tuning knob): 0 = one contiguous eighth per XCD (rounds 1-3), n = n tiles per chunk.
#ifndef SCSFM_XCD_CHUNK
#define SCSFM_XCD_CHUNK 266
#endif
__device__ __forceinline__ BlockId xcd_tile_of(int p, int nx, int ny, int nz) {
  const int n = nx * ny * nz;
  const int xcd = p & 7, slot = p >> 3;
  constexpr int C = SCSFM_XCD_CHUNK;
  // tiles per XCD that whole chunks cover; the rest of the launch keeps the contiguous split
  const int full = C > 0 ? ((n >> 3) / C) * C : 0;
  int l;
  if (C > 0 && slot < full) {
    l = ((slot / C) * 8 + xcd) * C + slot % C;
  } else {
    const int m = n - 8 * full, q = m >> 3, r = m & 7;
    l = 8 * full + xcd * q + (xcd < r ? xcd : r) + (slot - full);
  }
  BlockId b;
  b.x = l % nx;
  const int t = l / nx;
  b.y = t % ny;
  b.z = t / ny;
  return b;
}
__device__ __forceinline__ BlockId xcd_block_id() {
  const int nx = gridDim.x, ny = gridDim.y;
  return xcd_tile_of(((int)blockIdx.z * ny + (int)blockIdx.y) * nx + (int)blockIdx.x, nx, ny, (int)gridDim.z);
}

__device__ __forceinline__ int reflect_index(int i, int n) {
  // ReflectionPad2d(1) index map (pad[-1] = x[1], pad[n] = x[n-2]); clamped so that positions
  // further out (partial tiles) stay addressable -- their values are never used.
  if (i < 0) i = -i;
  if (i >= n) i = 2 * (n - 1) - i;
  return i < 0 ? 0 : (i >= n ? n - 1 : i);
}

// fabs / fmin / fmax builtins: |x| is a free source modifier and min / max are single instructions on
// gfx950 (the ternary forms compile to compare + select because of their -0 / NaN corner cases)
__device__ __forceinline__ float t_abs(float x) { return __builtin_fabsf(x); }
__device__ __forceinline__ double t_abs(double x) { return __builtin_fabs(x); }
__device__ __forceinline__ float t_min(float a, float b) { return __builtin_fminf(a, b); }
__device__ __forceinline__ double t_min(double a, double b) { return __builtin_fmin(a, b); }
__device__ __forceinline__ float t_max(float a, float b) { return __builtin_fmaxf(a, b); }
__device__ __forceinline__ double t_max(double a, double b) { return __builtin_fmax(a, b); }
__device__ __forceinline__ float t_med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
__device__ __forceinline__ double t_med3(double x, double lo, double hi) { return t_min(t_max(x, lo), hi); }
template <typename T> __device__ __forceinline__ T t_sgn(T x) { return x > T(0) ? T(1) : (x < T(0) ? T(-1) : T(0)); }
__device__ __forceinline__ float t_floor(float x) { return floorf(x); }
__device__ __forceinline__ double t_floor(double x) { return floor(x); }
__device__ __forceinline__ float t_exp(float x) { return expf(x); }
__device__ __forceinline__ double t_exp(double x) { return exp(x); }
// exp of a small non-positive argument (edge weights): one v_exp_f32 after the log2(e) scaling, ~|x| 2^-24
// relative error from the scaling + 1 ulp, instead of expf's ~12-instruction range reduction
__device__ __forceinline__ float t_exp_weight(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }
__device__ __forceinline__ double t_exp_weight(double x) { return exp(x); }
__device__ __forceinline__ void t_sincos(float x, float* s, float* c) { *s = sinf(x); *c = cosf(x); }
__device__ __forceinline__ void t_sincos(double x, double* s, double* c) { *s = sin(x); *c = cos(x); }
// Load at a 32-bit byte offset from a wave-uniform base: compiles to the scalar-base addressing mode
// (global_load ... v_off, s[base:base+1]) with no 64-bit vector address arithmetic.  Every plane this
// library indexes that way is below 4 GiB: the entry points reject larger images (dims_ok below).
// A read that must stay an LDS access: behind an if / else whose other side reads global memory the optimiser would
// otherwise select between the two pointers and issue ONE flat load (through the texture path) for both.
template <typename T>
__device__ __forceinline__ T lds_ld(const T* p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return *(const __attribute__((address_space(3))) T*)p;
#else
  return *p;
#endif
}
template <typename T>
__device__ __forceinline__ T ld_at(const T* __restrict__ base, unsigned byte_off) {
  return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void st_at(T* __restrict__ base, unsigned byte_off, T v) {
  *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}

// reciprocal: v_rcp_f32 (1 ulp) on the fp32 product path, exact division for the fp64 check path
__device__ __forceinline__ float t_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
__device__ __forceinline__ double t_rcp(double x) { return 1.0 / x; }
__device__ __forceinline__ int t_clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }
__device__ __forceinline__ float t_sqrt(float x) { return sqrtf(x); }
__device__ __forceinline__ double t_sqrt(double x) { return sqrt(x); }

#ifdef PROBE_TIMING  // tuning builds only (tools/march_timing.py): wave 0's clock at the stage boundaries of every chunk / tile
constexpr int kProbeStamps = 24, kProbeChunks = 6, kProbeWgs = 4096;
__device__ unsigned long long g_probe[kProbeWgs * kProbeChunks * kProbeStamps];
#define STAMP(i)                                                                                             \
  do {                                                                                                       \
    if (threadIdx.x == 0 && probe_wg < kProbeWgs && probe_chunk < kProbeChunks)                              \
      g_probe[(probe_wg * kProbeChunks + probe_chunk) * kProbeStamps + (i)] = __builtin_readcyclecounter();   \
  } while (0)
#else
#define STAMP(i) do {} while (0)
#endif

// ------------------------------------------------------------------------------------------------------
// lane <- neighbouring lane (DPP wave shift; lanes without a neighbour read 0).  The compiler folds the shift into
// the consuming add / fmac (v_add_f32_dpp ... wave_shr:1); half-rate VALU, no LDS traffic.
// ------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_left(float v) {   // value of lane - 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_right(float v) {  // value of lane + 1
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ double lane_left(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x138, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x138, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double lane_right(double v) {
  const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
  const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x130, 0xf, 0xf, true);
  const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x130, 0xf, 0xf, true);
  return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
template <typename T> __device__ __forceinline__ T box3(T v) { return (v + lane_left(v)) + lane_right(v); }

// Sum over the 64 lanes of a wave, valid in lane 63 ONLY: an inclusive scan by DPP (row shifts by 1, 2, 4, 8, then the
// two row broadcasts -- the sequence of LLVM's atomic optimiser): six v_add_f32_dpp, no LDS traffic and no lane-index
// arithmetic (a butterfly of ds_bpermute costs ~10 vector instructions per step in a block that has not formed the
// lane's permute addresses yet).  The host simulation and fp64 take the butterfly (every lane then holds the sum).
template <int kCtrl, int kRowMask, bool kBound>
__device__ __forceinline__ float dpp_term(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), kCtrl, kRowMask, 0xf, kBound));
}
__device__ __forceinline__ float wave_sum_last(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
  v += dpp_term<0x111, 0xf, true>(v);  // row_shr:1
  v += dpp_term<0x112, 0xf, true>(v);  // row_shr:2
  v += dpp_term<0x114, 0xf, true>(v);  // row_shr:4
  v += dpp_term<0x118, 0xf, true>(v);  // row_shr:8
  v += dpp_term<0x142, 0xa, false>(v);  // row_bcast:15 into rows 1 and 3
  v += dpp_term<0x143, 0xc, false>(v);  // row_bcast:31 into rows 2 and 3
  return v;
#else
  return wave_sum(v);
#endif
}
__device__ __forceinline__ double wave_sum_last(double v) { return wave_sum(v); }


}  // namespace scsfm
