// The speculative forward of a pair-direction as a COLUMN MARCH (round 3; replaces the 64 x 16 tile with a 1-pixel
// apron on every side that computed 66 x 18 warps and 64 x 16 statistics for 62 x 14 outputs).
//
// A workgroup (4 waves) owns a band of 64 image columns -- lane l <-> column x0 + l, x0 = 60 band - 2 -- and a segment
// of rows [ys, ye).  It walks down the segment in chunks of CH = 4 STRIP rows (wave w takes rows 4w .. 4w + 3 of a
// chunk: the same strips as before, so the SSIM window sums still slide down a thread's column), and the three stages
// of the work trail each other by one row:
//
//     stage W (warp)        rows a     .. a + CH - 1   project, gather, masks, depth inconsistency     -> LDS rows 2 .. CH + 1
//     stage S (statistics)  rows a - 1 .. a + CH - 2   SSIM forward at the pixel, 1/9 (g_mu, g_E2, g_Exy) -> LDS rows 2 .. CH + 1
//     stage O (outputs)     rows a - 2 .. a + CH - 3   transposed box filter -> dL/d warped colour; then the geometry tail
//
// and the two rows a stage needs from the chunk before sit in rows 0, 1 of its LDS planes (carried over by the wave
// that produced them).  Vertically nothing is computed twice inside a segment (4 extra warped rows and 2 extra
// statistics rows per SEGMENT instead of per 14 rows); horizontally 64 lanes produce 60 outputs and no lane ever
// warps a second ("ring") pixel.  The block reductions (three forward sums, twelve pose partials) happen once per
// segment instead of once per tile.
//
// LDS (fp32, 52,416 B: three workgroups per CU, 168 VGPRs): the (target, warped) pairs of the three colours 3 x 18 x 64 x
// 8 B (once a colour's statistics are done its plane is dead but for the two rows the next chunk needs: dL/d warped
// colour of the outputs is parked there until the tail), the three gradient maps of ONE colour at a time 3 x 18 x 64 x
// 4 B (the scatter window of the tail lives there afterwards) with their carried rows per colour, the weight / mask
// plane, and one dL/d diff_depth row per wave (an output row is the statistics row of the same thread one step
// earlier, except for a wave's first).
//
// Reference lines: loss_functions.py:95-119 (compute_pairwise_loss), :11-42 (SSIM), inverse_warp.py:230-269.
#pragma once
#include "scsfm_geom.h"
#include "scsfm_ssim.h"

namespace scsfm {

#ifdef PROBE_TIMING  // tuning builds only (tools/march_timing.py): wave 0's clock at the stage boundaries of every chunk
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

constexpr int kBandOut = kWave - 4;  // columns a band writes (lanes 2 .. 61)
// Rows per thread and waves per workgroup of a chunk.  fp32: 8 waves x 2 rows.  What bounds this kernel is the length
// of a chunk's chain of dependent memory round trips, LDS round trips and barriers (tools/march_timing.py: 43,000
// cycles per 16-row chunk with 4 waves x 4 rows at three workgroups per CU, of which a wave issues vector
// instructions for 6,500), so the chunk is spread over twice the waves: half the pixels per thread means half the
// registers (16 waves per CU again, two workgroups' LDS instead of three or four) and both pixels' gathers in flight
// at once.  The price is window sums over 2 + 2 instead of 4 + 2 rows per strip (+5 % vector instructions).
template <typename T> struct March { static constexpr int kStrip = 2, kWaves = 8; };
template <> struct March<double> { static constexpr int kStrip = 2, kWaves = 4; };  // fp64 check path: half the planes

// Window sums of STRIP pixels down a column of a [rows][64] plane of (x, y) pairs: rows row0 .. row0 + STRIP + 1,
// columns cl / col / cr (the lane's neighbours, clamped at the band's ends: lanes 0 and 63 produce no statistics
// anyone uses).  cen[j] = the pair at (row0 + j, col).
template <typename T, int STRIP>
__device__ __forceinline__ void band_window_sums(const typename Vec2<T>::type (*tile)[kWave], int row0, int cl, int col,
                                                 int cr, WinSums<T>* out, typename Vec2<T>::type* cen) {
  typedef typename Vec2<T>::type V2;
  V2 h1[STRIP + 2], h2[STRIP + 2];
  T hxy[STRIP + 2];
#pragma unroll
  for (int r = 0; r < STRIP + 2; ++r) {
    const V2 a = tile[row0 + r][cl], b = tile[row0 + r][col], c = tile[row0 + r][cr];
    h1[r] = a + b + c;
    h2[r] = a * a + b * b + c * c;
    hxy[r] = a[0] * a[1] + b[0] * b[1] + c[0] * c[1];
    cen[r] = b;
  }
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    out[k].s1 = h1[k] + h1[k + 1] + h1[k + 2];
    out[k].s2 = h2[k] + h2[k + 1] + h2[k + 2];
    out[k].sxy = hxy[k] + hxy[k + 1] + hxy[k + 2];
  }
}

// Transpose of (ReflectionPad2d(1) o 3x3 box) down a column strip of NMAP [ROWS][64] maps: output k sits at map row
// row0 + k + 1 (image row py0 + k, image column px); an output next to the image border reaches the border pixel twice.
template <typename T, int STRIP, int ROWS, int NMAP>
__device__ __forceinline__ void band_box_transpose(const T (*g)[ROWS][kWave], int row0, int cl, int col, int cr, int px,
                                                   int py0, int H, int W, T (*out)[NMAP]) {
  const T wl = reflect_mult<T>(-1, px, W), wr = reflect_mult<T>(1, px, W);
  T h[NMAP][STRIP + 2];
#pragma unroll
  for (int j = 0; j < STRIP + 2; ++j)
#pragma unroll
    for (int m = 0; m < NMAP; ++m) h[m][j] = wl * g[m][row0 + j][cl] + g[m][row0 + j][col] + wr * g[m][row0 + j][cr];
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const T wt = reflect_mult<T>(-1, py0 + k, H), wb = reflect_mult<T>(1, py0 + k, H);
#pragma unroll
    for (int m = 0; m < NMAP; ++m) out[k][m] = wt * h[m][k] + h[m][k + 1] + wb * h[m][k + 2];
  }
}

// Nothing moves across this point when the compiler schedules the instructions (the loads issued before it stay before
// everything that consumes them).
__device__ __forceinline__ void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

// The weight / mask plane holds one number per pixel: -1 where the mask is 0, else the weight of the pixel's blend in
// the photometric sum (1 - diff_depth with the weight mask, 1 without; in [0, 1]).
template <typename T> __device__ __forceinline__ T mask_of(T val) { return clamp01(val * T(1e30) + T(1)); }
template <typename T> __device__ __forceinline__ T coef_of(T val) { return clamp01(val); }

// One segment of one band of one (pair, batch element).  blk.x = band, blk.y = segment, blk.z = pair * B + b.
template <typename T, bool kSsim, bool kScaled, unsigned kFlags>
__device__ __forceinline__ void march_segment(const BlockId blk, int nbands, int nsegs, int seg_rows, const PairBatch<T>& pb,
                                              int B, int H, int W, unsigned flags_arg, T r_hint) {
  const unsigned flags = kFlags == kRuntimeFlags ? flags_arg : kFlags;
  const int pair = blk.z / B, b = blk.z - pair * B;
  const PairArgs<T>& pa = pb.p[pair];
  const T* __restrict__ tgt_img = pa.tgt_img;
  const T* __restrict__ ref_img = pa.ref_img;
  typedef typename Vec2<T>::type V2;
  typedef typename WinCell<T>::type Cell;
  constexpr int STRIP = March<T>::kStrip, NW = March<T>::kWaves, NT = NW * kWave, CH = STRIP * NW;
  constexpr int LAG = kSsim ? 2 : 0;  // rows by which the outputs trail the warp
  constexpr int WW = kWinW, WH = kWinH * CH / kTileH;
  constexpr int RS = kSsim ? CH + 2 : 1, CS = kSsim ? kWave : 1;  // (planes that only exist with SSIM)
  __shared__ V2 sXY[3][RS][CS];       // (target, warped) per colour: rows 0, 1 = the chunk before's last two, 2 .. CH + 1 this chunk's warps
  __shared__ T sG[3][RS][CS];         // 1/9 (g_mu_y, g_E[y^2], g_E[xy]) of one colour: rows 0, 1 carried
  __shared__ T cG[3][3][2][CS];       // per colour and map: the last two statistics rows of the chunk before
  __shared__ T sC[RS][CS];            // weight / mask plane (mask_of, coef_of), rows as in sXY
  __shared__ T sGdd[kSsim ? NW + 1 : 1][CS];  // dL/d diff_depth of each wave's last statistics row (row w + 1; row 0: the chunk before's)
  __shared__ int sBox[NW][4];
  __shared__ double sAcc[NW][12];     // pose partials (pixel_geometry_bwd), summed per wave at the end of every chunk's tail
  // the scatter window of the tail: in sG once the chunk's last transposed box filter has read it
  constexpr bool kWinInG = kSsim && sizeof(Cell) * WW * WH <= sizeof(T) * 3 * RS * CS;
  __shared__ Cell win_own[kWinInG ? 1 : WH][kWinInG ? 1 : WW];
  Cell(*const win)[WW] = kWinInG ? reinterpret_cast<Cell(*)[WW]>(&sG[0][0][0]) : reinterpret_cast<Cell(*)[WW]>(&win_own[0][0]);
  // scratch of the block sum at the end of the segment: in sG or its own
  constexpr bool kRedInG = kSsim && sizeof(T) * 3 * RS * CS >= sizeof(double) * 3 * NW;
  __shared__ double red_own[kRedInG ? 1 : 3 * NW];
  double* const red = kRedInG ? reinterpret_cast<double*>(&sG[0][0][0]) : &red_own[0];
  // dL/d warped colour c of the outputs waits for the tail in the (by then dead) front of colour c's plane
  static_assert(!kSsim || sizeof(T) * CH * kWave <= sizeof(V2) * CH * kWave, "parking space");
  auto park = [&](int c, int slot, int lane_) -> T* { return reinterpret_cast<T*>(&sXY[c][0][0]) + slot * kWave + lane_; };

  if (threadIdx.x < NW * 12) (&sAcc[0][0])[threadIdx.x] = 0.0;
  if constexpr (kSsim) {
    // Every plane is read before all of it has been written (rows of waves that had nothing to do, the carried rows of
    // the first chunk): what is read there only reaches results nobody keeps, but it has to be finite -- 0 x NaN is
    // not 0 -- so the planes start from zeroes and only ever hold values computed from the inputs.
    auto zero = [](void* p, size_t bytes) {
      for (unsigned i = threadIdx.x; i < bytes / sizeof(int); i += NT) reinterpret_cast<int*>(p)[i] = 0;
    };
    zero(sXY, sizeof(sXY)); zero(sG, sizeof(sG)); zero(cG, sizeof(cG)); zero(sC, sizeof(sC)); zero(sGdd, sizeof(sGdd));
    __syncthreads();
  }
  // (the wave index as a scalar: every row index, row predicate and LDS row address below is then scalar arithmetic
  // and every `if (wave ...)` a scalar branch)
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave);
  // ... and as a vector register for LDS addresses: a DS instruction adds an immediate to ONE address register, so
  // (wave's first row, column) is formed once per thread and every row / plane of the planes is an immediate away
  // (with the scalar the compiler kept two dozen loop-invariant sums of the two in registers, and spilled them)
  const int wrow = (int(threadIdx.x) / kWave) * STRIP;
  static_assert(kWinH * CH % kTileH == 0, "window height");
  const bool with_mask = (flags & SCSFM_WITH_MASK) != 0, with_auto = (flags & SCSFM_WITH_AUTO_MASK) != 0;
  // of the per-element constants the march needs M and c only (K^-1 once, at the very end)
  BatchConsts<T> bc;
  {
    const BatchConsts<T>* __restrict__ src = pa.consts + b;
#pragma unroll
    for (int i = 0; i < 9; ++i) { bc.M[i] = src->M[i]; bc.Kinv[i] = T(0); bc.A[i] = T(0); }
#pragma unroll
    for (int i = 0; i < 3; ++i) bc.c[i] = src->c[i];
    bc.pad[0] = bc.pad[1] = T(0);
  }
  const unsigned plane = unsigned(H) * unsigned(W);
  const size_t gplane = (size_t)B * plane;
  tgt_img += (size_t)b * 3 * plane;
  ref_img += (size_t)b * 3 * plane;
  const DepthMap<T, kScaled> tgt_depth = depth_map<kScaled>(pa.tgt_depth, b, H, W, pa.ds);
  const DepthMap<T, kScaled> ref_depth = depth_map<kScaled>(pa.ref_depth, b, H, W, pa.ds);
  T* __restrict__ g_dense = pa.gbuf + kPlaneDense * gplane + (size_t)b * plane;
  T* __restrict__ g_scatter = pa.gbuf + kPlaneScatter * gplane + (size_t)b * plane;

  const int ys = blk.y * seg_rows, ye = ys + seg_rows < H ? ys + seg_rows : H;
  const int px = blk.x * kBandOut - 2 + lane;         // this lane's image column (may lie outside the image)
  const int u = reflect_index(px, W);                 // ... reflected into it (ReflectionPad2d(1); further out: clamped)
  const bool in_x = px >= 0 && px < W;
  const bool own_x = lane >= 2 && lane <= kWave - 3 && px < W;
  const int cl = lane > 0 ? lane - 1 : 0, cr = lane < kWave - 1 ? lane + 1 : kWave - 1;
  const T bg = r_hint;  // dL/d(geometry sum) in units of the photo coefficient (a = 1)

  // Bounding box of the north-west taps of the pixels that scatter (kept per wave, met in the tail).  The tail of a chunk
  // handles the rows warped in it except the last LAG, plus the last LAG rows of the chunk before: the last wave keeps
  // the box of those rows (`late`) from one chunk to the next.
  int late0 = 1 << 30, late1 = -(1 << 30), late2 = 1 << 30, late3 = -(1 << 30);
  T fsum[3] = {T(0), T(0), T(0)};  // the forward's three sums over the pixels this workgroup owns

#ifdef PROBE_TIMING
  const int probe_wg = (blk.z * nsegs + blk.y) * nbands + blk.x;
  int probe_chunk = -1;
#endif
  for (int a = ys - LAG; a < ye + LAG; a += CH) {
#ifdef PROBE_TIMING
    ++probe_chunk;
#endif
    STAMP(0);
    // ---------------- stage W: rows a + wave STRIP + k ------------------------------------------------------
    const int rw0 = a + wave * STRIP;
    T gI[kSsim ? 1 : STRIP][3];      // without SSIM: dL/d warped colour of the same rows (with: parked in LDS by stage O)
    T gdd[STRIP];                    // dL/d diff_depth of this thread's OUTPUT rows
    int bx0 = 1 << 30, bx1 = -(1 << 30), by0 = 1 << 30, by1 = -(1 << 30);
    int nx0 = 1 << 30, nx1 = -(1 << 30), ny0 = 1 << 30, ny1 = -(1 << 30);  // last wave: the rows the NEXT chunk's tail handles
    const bool w_on = rw0 < ye + LAG && rw0 <= H;  // (wave-uniform) some row of this wave is still needed
    if constexpr (kSsim) {
      // The last wave's last two rows of a chunk are rows 0, 1 of the next one: that wave moves them before it writes
      // its new ones (its own LDS accesses execute in order; the others read rows 0, 1 behind the next barrier only).
      if (wave == NW - 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
          for (int c = 0; c < 3; ++c) sXY[c][j][lane] = sXY[c][CH + j][lane];
          sC[j][lane] = sC[CH + j][lane];
        }
      }
    }
    if (w_on) {
      T in_d[STRIP], in_t[STRIP][3], in_r[STRIP][3];
#pragma unroll
      for (int k = 0; k < STRIP; ++k)
        load_pixel(u, reflect_index(rw0 + k, H), W, plane, tgt_img, ref_img, tgt_depth, with_auto, in_d[k], in_t[k], in_r[k]);
      // Every gather of the strip is in flight before the first one is consumed: measured with the clock at the stage
      // boundaries (tools/march_timing.py), a gather's round trip is ~2,500 cycles on the loaded chip, and pixel after
      // pixel the stage took 14,000 of a chunk's 43,000 cycles -- whether four waves or one had rows to warp.
#ifndef SCSFM_W_GROUP  // pixels of a strip whose gathers are in flight together in stage W
#define SCSFM_W_GROUP 4
#endif
      constexpr int WG_ = SCSFM_W_GROUP < STRIP ? SCSFM_W_GROUP : STRIP;
#pragma unroll
      for (int k0 = 0; k0 < STRIP; k0 += WG_) {
      Sample<T> sm[WG_];
      TapRows<T> tc[WG_][3], td[WG_];
      T ident[WG_];  // auto-mask: sum_c |It - Ir| of the un-warped pair (loss_functions.py:104), ready before the gathers return
#pragma unroll
      for (int j = 0; j < WG_; ++j) {
        const int k = k0 + j;
        ident[j] = t_abs(in_t[k][0] - in_r[k][0]) + t_abs(in_t[k][1] - in_r[k][1]) + t_abs(in_t[k][2] - in_r[k][2]);
        sm[j] = project_pixel(bc, u, reflect_index(rw0 + k, H), in_d[k], H, W, flags);
#pragma unroll
        for (int c = 0; c < 3; ++c) tc[j][c] = load_tap_rows(ref_img + c * plane, sm[j]);
        td[j] = ref_depth.taps(sm[j]);
      }
#ifndef PROBE_NO_WFENCE
      sched_fence();
#endif
#pragma unroll
      for (int j = 0; j < WG_; ++j) {
        const int k = k0 + j;
        const int rw = rw0 + k;
        const bool inimg = in_x && rw >= 0 && rw < H;
        const Sample<T>& s = sm[j];
        V2 xy[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) xy[c] = make2(in_t[k][c], bilerp_rows(tc[j][c], s));
        const T Dp = bilerp_rows(td[j], s);
        const T ddk = clamp01(t_abs(s.Z - Dp) * t_rcp(s.Z + Dp));  // loss_functions.py:101
        T m = (inimg && s.valid) ? T(1) : T(0);                    // inverse_warp.py:264
        if (with_auto) {  // loss_functions.py:103-105 (both means share the divisor 3: the sums are compared)
          const T warped = clamp01(t_abs(xy[0][0] - xy[0][1])) + clamp01(t_abs(xy[1][0] - xy[1][1])) +
                           clamp01(t_abs(xy[2][0] - xy[2][1]));
          m = (warped < ident[j]) ? m : T(0);
        }
        const T wgt = with_mask ? T(1) - ddk : T(1);               // loss_functions.py:111-113
        const bool own = own_x && rw >= ys && rw < ye;
        fsum[1] += own ? ddk * m : T(0);
        fsum[2] += own ? m : T(0);
        if (own_x && rw >= 0 && rw < H && m != T(0)) {  // a pixel that scatters (or, rows beyond ye, never does): where its taps lie
          if (LAG && wave == NW - 1 && k >= STRIP - LAG) {
            nx0 = s.xa < nx0 ? s.xa : nx0; nx1 = s.xa > nx1 ? s.xa : nx1;
            ny0 = s.ya < ny0 ? s.ya : ny0; ny1 = s.ya > ny1 ? s.ya : ny1;
          } else {
            bx0 = s.xa < bx0 ? s.xa : bx0; bx1 = s.xa > bx1 ? s.xa : bx1;
            by0 = s.ya < by0 ? s.ya : by0; by1 = s.ya > by1 ? s.ya : by1;
          }
        }
        if constexpr (kSsim) {
#pragma unroll
          for (int c = 0; c < 3; ++c) sXY[c][2 + wrow + k][lane] = xy[c];
          sC[2 + wrow + k][lane] = m != T(0) ? wgt : T(-1);
        } else {
          // no SSIM: the photometric term is the clamped L1 alone and everything is local to the pixel
          T bsum = T(0);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const T d = xy[c][0] - xy[c][1];
            bsum += clamp01(t_abs(d));
            gI[k][c] = (m * wgt) * ((t_abs(d) <= T(1)) ? -t_sgn(d) : T(0));
          }
          gdd[k] = bg * m - (with_mask ? m * bsum : T(0));
          fsum[0] += own ? bsum * (m * wgt) : T(0);
        }
      }
      }
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) {
        const int a0 = __shfl_xor(bx0, o), a1 = __shfl_xor(bx1, o), c0 = __shfl_xor(by0, o), c1 = __shfl_xor(by1, o);
        bx0 = a0 < bx0 ? a0 : bx0; bx1 = a1 > bx1 ? a1 : bx1; by0 = c0 < by0 ? c0 : by0; by1 = c1 > by1 ? c1 : by1;
      }
      if (LAG && wave == NW - 1) {
#pragma unroll
        for (int o = kWave / 2; o > 0; o >>= 1) {
          const int a0 = __shfl_xor(nx0, o), a1 = __shfl_xor(nx1, o), c0 = __shfl_xor(ny0, o), c1 = __shfl_xor(ny1, o);
          nx0 = a0 < nx0 ? a0 : nx0; nx1 = a1 > nx1 ? a1 : nx1; ny0 = c0 < ny0 ? c0 : ny0; ny1 = c1 > ny1 ? c1 : ny1;
        }
      }
    }
    if (LAG && wave == NW - 1) {  // this chunk's tail: the rows parked a chunk ago instead of this chunk's last rows
      bx0 = late0 < bx0 ? late0 : bx0; bx1 = late1 > bx1 ? late1 : bx1; by0 = late2 < by0 ? late2 : by0; by1 = late3 > by1 ? late3 : by1;
      // (wave-uniform after the butterfly: kept in scalar registers from one chunk to the next)
      late0 = __builtin_amdgcn_readfirstlane(nx0); late1 = __builtin_amdgcn_readfirstlane(nx1);
      late2 = __builtin_amdgcn_readfirstlane(ny0); late3 = __builtin_amdgcn_readfirstlane(ny1);
    }
    STAMP(1);
    if (lane == 0) { sBox[wave][0] = bx0; sBox[wave][1] = bx1; sBox[wave][2] = by0; sBox[wave][3] = by1; }
    const int ro0 = a - LAG + wave * STRIP;  // this thread's output rows
    if constexpr (kSsim) {
      // ---------------- stages S and O, one colour at a time --------------------------------------------------
      const int rs0 = a - 1 + wave * STRIP;
      const bool o_on = ro0 + STRIP - 1 >= ys && ro0 < ye;
      // (the outputs' centre pixels are read in stage S: it also runs for a wave whose statistics rows all lie below the image)
      const bool s_on = (rs0 + STRIP - 1 >= ys - 1 && rs0 < (ye + 1 < H ? ye + 1 : H)) || o_on;
      T bsum[STRIP];
#pragma unroll
      for (int k = 0; k < STRIP; ++k) { bsum[k] = T(0); gdd[k] = T(0); }
      STAMP(2);
      __syncthreads();  // the chunk's warps are in LDS
      STAMP(3);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        V2 cen[STRIP + 2];
        if (wave == NW - 1) {  // the carried rows of this colour's maps (see the move of sXY's rows above)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) sG[m][j][lane] = cG[c][m][j][lane];
          if (c == 2) sGdd[0][lane] = sGdd[NW][lane];
        }
        if (s_on) {
          WinSums<T> ws[STRIP];
          band_window_sums<T, STRIP>(sXY[c], wrow, cl, lane, cr, ws, cen);
          T g1[STRIP], g2[STRIP], g3[STRIP], vS[STRIP];
#pragma unroll
          for (int k = 0; k < STRIP; ++k) vS[k] = sC[1 + wrow + k][lane];
#pragma unroll
          for (int k = 0; k < STRIP; ++k) {
            const SsimStats<T> st = ssim_stats(ws[k]);
            bsum[k] += T(0.85) * clamp01(st.raw);
            const T d = cen[k + 1][0] - cen[k + 1][1];
            bsum[k] += T(0.15) * clamp01(t_abs(d));  // loss_functions.py:109
            // s = clamp((1 - S)/2, 0, 1): d s / d S = -1/2 inside the clamp (inclusive bounds)
            const T gS = clamp01(st.raw) == st.raw ? coef_of(vS[k]) * T(0.85) * T(-0.5) : T(0);
            ssim_grad_y(st, gS, g1[k], g2[k], g3[k]);
            const int r = 2 + wrow + k;
            sG[0][r][lane] = g1[k]; sG[1][r][lane] = g2[k]; sG[2][r][lane] = g3[k];
          }
          if (wave == NW - 1) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              cG[c][0][j][lane] = g1[STRIP - 2 + j]; cG[c][1][j][lane] = g2[STRIP - 2 + j]; cG[c][2][j][lane] = g3[STRIP - 2 + j];
            }
          }
          if (c == 2) {
            // dL/d diff_depth of the statistics rows: directly (geometry loss) and through the weight mask (no
            // detach, loss_functions.py:111-113); the photometric sum of the rows this workgroup owns.  Output row k
            // of a thread is its statistics row k - 1; a wave's first output row is the wave before's last statistics row.
#pragma unroll
            for (int k = 0; k < STRIP; ++k) {
              const int rs = rs0 + k;
              const T mS = mask_of(vS[k]);
              const T g = bg * mS - (with_mask ? mS * bsum[k] : T(0));
              if (k < STRIP - 1) gdd[k + 1] = g; else sGdd[wave + 1][lane] = g;
              fsum[0] += (own_x && rs >= ys && rs < ye) ? bsum[k] * coef_of(vS[k]) : T(0);
            }
          }
        }
        STAMP(4 + 5 * c);
        __syncthreads();  // the colour's maps are complete
        STAMP(5 + 5 * c);
        if (o_on) {
          T gt[STRIP][3];
          band_box_transpose<T, STRIP, RS, 3>(sG, wrow, cl, lane, cr, px, ro0, H, W, gt);
#pragma unroll
          for (int k = 0; k < STRIP; ++k) {
            const T x = cen[k][0], y = cen[k][1], d = x - y;
            // d clamp(|d|, 0, 1) / d Iw: the clamp passes gradient on [0, 1] inclusive, abs uses sgn
            const T l1g = (t_abs(d) <= T(1)) ? -t_sgn(d) : T(0);
            *park(c, wrow + k, lane) =
                gt[k][0] + T(2) * y * gt[k][1] + x * gt[k][2] + coef_of(sC[wrow + k][lane]) * T(0.15) * l1g;
          }
        }
        STAMP(6 + 5 * c);
        if (c < 2) {
          __syncthreads();  // the maps are rewritten by the next colour
          STAMP(7 + 5 * c);
        }
      }
    }
    STAMP(17);
    __syncthreads();  // the last transposed box filter has read sG: the window may go there; sBox is complete
    STAMP(18);
    // ---------------- geometry tail: rows a - LAG + wave STRIP + k ---------------------------------------------
    for (int i = threadIdx.x; i < WW * WH; i += NT) (&win[0][0])[i] = Cell(0);
    int wx0, wy0, cx0, cy0, cx1, cy1;  // window origin; cells of the window the taps can reach
    {
      int x0 = sBox[0][0], x1 = sBox[0][1], y0 = sBox[0][2], y1 = sBox[0][3];
#pragma unroll
      for (int w = 1; w < NW; ++w) {
        x0 = sBox[w][0] < x0 ? sBox[w][0] : x0; x1 = sBox[w][1] > x1 ? sBox[w][1] : x1;
        y0 = sBox[w][2] < y0 ? sBox[w][2] : y0; y1 = sBox[w][3] > y1 ? sBox[w][3] : y1;
      }
      if (x0 > x1) { x0 = x1 = 0; y0 = y1 = 0; }  // nothing scatters
      const int ex = x1 - x0 + 2, ey = y1 - y0 + 2;  // cells touched (each pixel reaches one past its tap)
      wx0 = ex <= WW ? x0 - (WW - ex) / 2 : (x0 + x1 + 1) / 2 - WW / 2;
      wy0 = ey <= WH ? y0 - (WH - ey) / 2 : (y0 + y1 + 1) / 2 - WH / 2;
      cx0 = x0 - wx0; cx1 = x1 + 1 - wx0; cy0 = y0 - wy0; cy1 = y1 + 1 - wy0;
    }
    const bool t_on = ro0 + STRIP - 1 >= ys && ro0 < ye && !(flags & SCSFM_DEBUG_X4);
    T d_own[STRIP], gd[STRIP];
    if (t_on) {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int cy = ro0 + k < H ? (ro0 + k < 0 ? 0 : ro0 + k) : H - 1;
        d_own[k] = tgt_depth.at(u, cy, (unsigned(cy) * unsigned(W) + unsigned(u)) * unsigned(sizeof(T)));
      }
      if constexpr (kSsim) gdd[0] = sGdd[wave][lane];
    }
    STAMP(19);
    __syncthreads();  // the window's zeroes
    STAMP(20);
    if (t_on) {
      T acc[12];  // pose partials of this chunk's owned pixels
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = T(0);
#ifdef PROBE_SIMPLE_TAIL
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int ro = ro0 + k;
        gd[k] = T(0);
        if (!(own_x && ro >= ys && ro < ye)) continue;
        T g3[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          if constexpr (kSsim) g3[c] = *park(c, wrow + k, lane); else g3[c] = gI[k][c];
        }
        gd[k] = geom_pixel<T, Cell, WW, WH>(bc, u, ro, d_own[k], g3, gdd[k], ref_img, ref_depth, plane, H, W, flags, win,
                                            wx0, wy0, g_scatter, acc);
      }
#else
      // two pixels' gathers in flight at a time (the whole strip's sampling state would not fit the register budget).
      // Branch-free: a pixel this thread does not own is sampled at a clamped position with zero upstream gradients
      // (no scatter, zero partials), so that no value is defined on one side of a branch only.
#pragma unroll
      for (int k0 = 0; k0 < STRIP; k0 += 2) {
        GeomTaps<T> f[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int ro = ro0 + k0 + j;
          const int cy = ro < H ? (ro < 0 ? 0 : ro) : H - 1;
          // (u == px for a pixel inside the image: the projection's column part is shared with stage W)
          f[j] = geom_fetch(bc, u, cy, d_own[k0 + j], ref_img, ref_depth, plane, H, W, flags);
        }
#ifndef PROBE_NO_TFENCE
        sched_fence();
#endif
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int k = k0 + j, ro = ro0 + k;
          const bool mine = own_x && ro >= ys && ro < ye;
          const int cy = ro < H ? (ro < 0 ? 0 : ro) : H - 1;
          T g3[3];
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            if constexpr (kSsim) g3[c] = *park(c, wrow + k, lane); else g3[c] = gI[k][c];
            g3[c] = mine ? g3[c] : T(0);
          }
          gd[k] = geom_consume<T, Cell, WW, WH>(bc, f[j], u, cy, d_own[k], g3, mine ? gdd[k] : T(0), H, W, flags, win, wx0, wy0,
                                                g_scatter, acc);
        }
      }
#endif
      // twelve registers that would otherwise live through every stage of every chunk: summed over the wave here
      // (N + 6 shuffles for the lot) and kept in LDS, in fp64, one row per wave (no atomics: a wave owns its row)
      bool lead;
      const int idx = wave_sum_packed<12>(acc, lead);
      if (lead) sAcc[wave][idx] += double(acc[0]);
    }
    STAMP(21);
    __syncthreads();  // the scatter's LDS atomics precede the flush
    STAMP(22);
    if (t_on) {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int ro = ro0 + k;
        if (own_x && ro >= ys && ro < ye) st_at(g_dense, (unsigned(ro) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), gd[k]);
      }
    }
    if (!(flags & (SCSFM_DEBUG_X1 | SCSFM_DEBUG_X5)))
      flush_scatter_region<T, Cell, WW, WH, NT>(win, wx0, wy0, cx0, cy0, cx1, cy1, g_scatter, W);
    STAMP(23);
    // (the next chunk writes sBox, the planes of stage W and -- behind its first barrier -- sG / the window: nothing the
    // flush reads is touched before every thread has passed that barrier)
  }
  // ---------------- the segment's sums ------------------------------------------------------------------------
  __syncthreads();
  {
    bool lead;
    const int idx = wave_sum_packed<3>(fsum, lead);
    if (lead) red[wave * 3 + idx] = double(fsum[0]);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double* o = pa.partials + 3 * ((size_t)(b * nsegs + blk.y) * nbands + blk.x);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      double v = 0.0;
      for (int w = 0; w < NW; ++w) v += red[w * 3 + i];
      o[i] = v;
    }
  }
  if (threadIdx.x == 0) {  // (block_sum's barrier orders the waves' last additions to sAcc before this)
    double* o = pa.gPp + 12 * ((size_t)(b * nsegs + blk.y) * nbands + blk.x);
    double g[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      g[i] = 0.0;
      for (int w = 0; w < NW; ++w) g[i] += sAcc[w][i];
    }
    pose_partials_to_A(pa.consts[b], g);
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = g[i];
  }
}

}  // namespace scsfm
