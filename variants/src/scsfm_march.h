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
// LDS (fp32, 40,512 B: four workgroups per CU): ONE colour's (target, warped) pairs at a time -- the warped colours
// wait in registers for their turn, dL/d(warped colour) of the finished ones likewise -- 18 x 64 x 8 B, the three
// gradient maps of that colour 3 x 18 x 64 x 4 B (the scatter window of the tail lives there afterwards), the
// weight / mask plane and the dL/d diff_depth plane, and the carried rows of every plane.
//
// Reference lines: loss_functions.py:95-119 (compute_pairwise_loss), :11-42 (SSIM), inverse_warp.py:230-269.
#pragma once
#include "scsfm_geom.h"
#include "scsfm_ssim.h"

namespace scsfm {

constexpr int kBandOut = kWave - 4;  // columns a band writes (lanes 2 .. 61)
#ifndef SCSFM_MARCH_STRIP
#define SCSFM_MARCH_STRIP 4
#endif
template <typename T> struct March { static constexpr int kStrip = SCSFM_MARCH_STRIP, kWaves = 4; };
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
  constexpr int STRIP = March<T>::kStrip, NW = kThreads / kWave, CH = STRIP * NW;
  constexpr int LAG = kSsim ? 2 : 0;  // rows by which the outputs trail the warp
  constexpr int WW = kWinW, WH = kWinH * CH / kTileH;
  constexpr int RS = kSsim ? CH + 2 : 1, CS = kSsim ? kWave : 1;  // (planes that only exist with SSIM)
  __shared__ V2 sXY[RS][CS];          // (target, warped) of ONE colour: rows 0, 1 carried, 2 .. CH + 1 this chunk's warps
  __shared__ V2 cXY[3][2][CS];        // per colour: the last two warped rows of the chunk before
  __shared__ T sG[3][RS][CS];         // 1/9 (g_mu_y, g_E[y^2], g_E[xy]) of one colour: rows 0, 1 carried
  __shared__ T cG[3][3][2][CS];       // per colour and map: the last two statistics rows of the chunk before
  __shared__ T sC[RS][CS];            // weight / mask plane (mask_of, coef_of), rows as in sXY
  __shared__ T cC[2][CS];
  __shared__ T sGdd[kSsim ? CH + 1 : 1][CS];  // dL/d diff_depth: row 0 carried, 1 .. CH this chunk's statistics rows
  __shared__ T cGdd[CS];
  __shared__ int sBox[NW][4];
  __shared__ double sAcc[NW][12];     // pose partials (pixel_geometry_bwd), summed per wave at the end of every chunk's tail
  // the scatter window of the tail: in sG once the chunk's last transposed box filter has read it
  constexpr bool kWinInG = kSsim && sizeof(Cell) * WW * WH <= sizeof(T) * 3 * RS * CS;
  __shared__ Cell win_own[kWinInG ? 1 : WH][kWinInG ? 1 : WW];
  Cell(*const win)[WW] = kWinInG ? reinterpret_cast<Cell(*)[WW]>(&sG[0][0][0]) : reinterpret_cast<Cell(*)[WW]>(&win_own[0][0]);
  // scratch of the block sum at the end of the segment: in sXY or its own
  constexpr bool kRedInXY = kSsim && sizeof(V2) * RS * CS >= sizeof(double) * 3 * NW;
  __shared__ double red_own[kRedInXY ? 1 : 3 * NW];
  double* const red = kRedInXY ? reinterpret_cast<double*>(&sXY[0][0]) : &red_own[0];

  if (threadIdx.x < NW * 12) (&sAcc[0][0])[threadIdx.x] = 0.0;
  if constexpr (kSsim) {
    // Every plane is read before all of it has been written (rows of waves that had nothing to do, the carried rows of
    // the first chunk): what is read there only reaches results nobody keeps, but it has to be finite -- 0 x NaN is
    // not 0 -- so the planes start from zeroes and only ever hold values computed from the inputs.
    auto zero = [](void* p, size_t bytes) {
      for (unsigned i = threadIdx.x; i < bytes / sizeof(int); i += kThreads) reinterpret_cast<int*>(p)[i] = 0;
    };
    zero(sXY, sizeof(sXY)); zero(cXY, sizeof(cXY)); zero(sG, sizeof(sG)); zero(cG, sizeof(cG));
    zero(sC, sizeof(sC)); zero(cC, sizeof(cC)); zero(sGdd, sizeof(sGdd)); zero(cGdd, sizeof(cGdd));
    __syncthreads();
  }
  // (the wave index as a scalar: every row index, row predicate and LDS row address below is then scalar arithmetic
  // and every `if (wave ...)` a scalar branch)
  const int lane = threadIdx.x & (kWave - 1), wave = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave);
  // ... and as a vector register for LDS addresses: a DS instruction adds an immediate to ONE address register, so
  // (wave's first row, column) is formed once per thread and every row / plane is an immediate away
  const int wrow = (int(threadIdx.x) / kWave) * STRIP;
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
  const int cxo = px < 0 ? 0 : (px < W ? px : W - 1);  // clamped column (addresses of rows nobody owns)
  const T bg = r_hint;  // dL/d(geometry sum) in units of the photo coefficient (a = 1)

  // Bounding box of the north-west taps of the pixels that scatter (kept per wave, met in the tail).  The tail of a chunk
  // handles the rows warped in it except the last LAG, plus the last LAG rows of the chunk before: the last wave keeps
  // the box of those rows (`late`) from one chunk to the next.
  int late0 = 1 << 30, late1 = -(1 << 30), late2 = 1 << 30, late3 = -(1 << 30);
  T fsum[3] = {T(0), T(0), T(0)};  // the forward's three sums over the pixels this workgroup owns

  for (int a = ys - LAG; a < ye + LAG; a += CH) {
    // ---------------- stage W: rows a + wave STRIP + k ------------------------------------------------------
    const int rw0 = a + wave * STRIP;
    T val[STRIP];                    // weight / mask of the warped pixel (see mask_of / coef_of)
    // the warped colours wait in registers for their turn in LDS; the target colours are fetched again when it comes
    // (L1 / L2 hits, requested a stage ahead) -- eight registers less through the statistics of the first colour
    T xt[kSsim ? STRIP : 1], yw[kSsim ? STRIP : 1][3];
    T gI[STRIP][3];                  // dL/d warped colour of this thread's OUTPUT rows
    T gdd_own[kSsim ? 1 : STRIP];    // without SSIM: dL/d diff_depth of the same rows
    int bx0 = 1 << 30, bx1 = -(1 << 30), by0 = 1 << 30, by1 = -(1 << 30);
    int nx0 = 1 << 30, nx1 = -(1 << 30), ny0 = 1 << 30, ny1 = -(1 << 30);  // last wave: the rows the NEXT chunk's tail handles
    const bool w_on = rw0 < ye + LAG && rw0 <= H;  // (wave-uniform) some row of this wave is still needed
    if (w_on) {
      T in_d[STRIP], in_t[STRIP][3], in_r[STRIP][3];
#pragma unroll
      for (int k = 0; k < STRIP; ++k)
        load_pixel(u, reflect_index(rw0 + k, H), W, plane, tgt_img, ref_img, tgt_depth, with_auto, in_d[k], in_t[k], in_r[k]);
      T ident[STRIP];  // auto-mask: sum_c |It - Ir| of the un-warped pair (loss_functions.py:104); frees the nine Ir registers
#pragma unroll
      for (int k = 0; k < STRIP; ++k)
        ident[k] = t_abs(in_t[k][0] - in_r[k][0]) + t_abs(in_t[k][1] - in_r[k][1]) + t_abs(in_t[k][2] - in_r[k][2]);
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        sched_fence();  // one pixel's sampling state at a time (register budget: four workgroups per CU)
        const int rw = rw0 + k;
        const bool inimg = in_x && rw >= 0 && rw < H;
        V2 xy[3];
        const Sample<T> s = warp_colours(bc, u, reflect_index(rw, H), in_d[k], in_t[k], H, W, flags, ref_img, xy);
        const T Dp = bilerp_rows(ref_depth.taps(s), s);
        const T ddk = clamp01(t_abs(s.Z - Dp) * t_rcp(s.Z + Dp));  // loss_functions.py:101
        T m = (inimg && s.valid) ? T(1) : T(0);                    // inverse_warp.py:264
        if (with_auto) {  // loss_functions.py:103-105 (both means share the divisor 3: the sums are compared)
          const T warped = clamp01(t_abs(xy[0][0] - xy[0][1])) + clamp01(t_abs(xy[1][0] - xy[1][1])) +
                           clamp01(t_abs(xy[2][0] - xy[2][1]));
          m = (warped < ident[k]) ? m : T(0);
        }
        const T wgt = with_mask ? T(1) - ddk : T(1);               // loss_functions.py:111-113
        val[k] = m != T(0) ? wgt : T(-1);
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
          for (int c = 0; c < 3; ++c) yw[k][c] = xy[c][1];
          xt[k] = xy[0][0];
        } else {
          // no SSIM: the photometric term is the clamped L1 alone and everything is local to the pixel
          T bsum = T(0);
#pragma unroll
          for (int c = 0; c < 3; ++c) {
            const T d = xy[c][0] - xy[c][1];
            bsum += clamp01(t_abs(d));
            gI[k][c] = (m * wgt) * ((t_abs(d) <= T(1)) ? -t_sgn(d) : T(0));
          }
          gdd_own[k] = bg * m - (with_mask ? m * bsum : T(0));
          fsum[0] += own ? bsum * (m * wgt) : T(0);
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
      if constexpr (kSsim) {
#pragma unroll
        for (int k = 0; k < STRIP; ++k) sC[2 + wrow + k][lane] = val[k];
      }
    }
    // The last wave's last two rows of a chunk are rows 0, 1 of the next one.  That wave moves them: rows 0, 1 <- what
    // it parked a chunk ago (whether or not it has rows of its own this time), then its new rows into the parking
    // space -- its own LDS accesses execute in order, and nobody else touches the parking space.
    if constexpr (kSsim) {
      if (wave == NW - 1) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          sC[j][lane] = cC[j][lane];
          if (w_on) cC[j][lane] = val[STRIP - 2 + j];
        }
      }
    }
    if (LAG && wave == NW - 1) {  // this chunk's tail: the rows parked a chunk ago instead of this chunk's last rows
      bx0 = late0 < bx0 ? late0 : bx0; bx1 = late1 > bx1 ? late1 : bx1; by0 = late2 < by0 ? late2 : by0; by1 = late3 > by1 ? late3 : by1;
      // (wave-uniform after the butterfly: kept in scalar registers from one chunk to the next)
      late0 = __builtin_amdgcn_readfirstlane(nx0); late1 = __builtin_amdgcn_readfirstlane(nx1);
      late2 = __builtin_amdgcn_readfirstlane(ny0); late3 = __builtin_amdgcn_readfirstlane(ny1);
    }
    if (lane == 0) { sBox[wave][0] = bx0; sBox[wave][1] = bx1; sBox[wave][2] = by0; sBox[wave][3] = by1; }
    if constexpr (kSsim) {
      // ---------------- stages S and O, one colour at a time --------------------------------------------------
      const int rs0 = a - 1 + wave * STRIP, ro0 = a - 2 + wave * STRIP;
      const bool o_on = ro0 + STRIP - 1 >= ys && ro0 < ye;
      // (the outputs' centre pixels are read in stage S: it also runs for a wave whose statistics rows all lie below the image)
      const bool s_on = (rs0 + STRIP - 1 >= ys - 1 && rs0 < (ye + 1 < H ? ye + 1 : H)) || o_on;
      T bsum[STRIP];
#pragma unroll
      for (int k = 0; k < STRIP; ++k) bsum[k] = T(0);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if (w_on) {
#pragma unroll
          for (int k = 0; k < STRIP; ++k) sXY[2 + wrow + k][lane] = make2(xt[k], yw[k][c]);
        }
        if (wave == NW - 1) {
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            sXY[j][lane] = cXY[c][j][lane];
            if (w_on) cXY[c][j][lane] = make2(xt[STRIP - 2 + j], yw[STRIP - 2 + j][c]);
          }
        }
        __syncthreads();
        V2 cen[STRIP];
        if (wave == NW - 1) {
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int m = 0; m < 3; ++m) sG[m][j][lane] = cG[c][m][j][lane];
          if (c == 2) sGdd[0][lane] = cGdd[lane];
        }
        if (s_on) {
          // The strip's statistics in groups of SG outputs (SG + 2 rows of window sums each), one after the other: the
          // rows two groups share are summed twice (+26 vector instructions per colour), and the registers of a group
          // are free before the next one starts (register budget: four workgroups per CU).
          constexpr int SG = STRIP % 2 == 0 ? 2 : STRIP;
#pragma unroll
          for (int k0 = 0; k0 < STRIP; k0 += SG) {
            WinSums<T> ws[SG];
            V2 cg[SG + 2];
            band_window_sums<T, SG>(sXY, wrow + k0, cl, lane, cr, ws, cg);
#pragma unroll
            for (int j = 0; j < SG; ++j) cen[k0 + j] = cg[j];  // the OUTPUT rows' centre pixels (used by stage O)
#pragma unroll
            for (int j = 0; j < SG; ++j) {
              const int k = k0 + j, rs = rs0 + k;
              // (the weight / mask of the statistics rows is read here, per colour, rather than held in registers)
              const T vS = sC[1 + wrow + k][lane];
              const SsimStats<T> st = ssim_stats(ws[j]);
              bsum[k] += T(0.85) * clamp01(st.raw);
              const T d = cg[j + 1][0] - cg[j + 1][1];
              bsum[k] += T(0.15) * clamp01(t_abs(d));  // loss_functions.py:109
              // s = clamp((1 - S)/2, 0, 1): d s / d S = -1/2 inside the clamp (inclusive bounds)
              const T gS = clamp01(st.raw) == st.raw ? coef_of(vS) * T(0.85) * T(-0.5) : T(0);
              T g1, g2, g3;
              ssim_grad_y(st, gS, g1, g2, g3);
              const int r = 2 + wrow + k;
              sG[0][r][lane] = g1; sG[1][r][lane] = g2; sG[2][r][lane] = g3;
              if (wave == NW - 1 && k >= STRIP - 2) {
                cG[c][0][k - (STRIP - 2)][lane] = g1; cG[c][1][k - (STRIP - 2)][lane] = g2; cG[c][2][k - (STRIP - 2)][lane] = g3;
              }
              if (c == 2) {
                // dL/d diff_depth of the statistics rows: directly (geometry loss) and through the weight mask (no
                // detach, loss_functions.py:111-113); the photometric sum of the rows this workgroup owns
                const T mS = mask_of(vS);
                const T g = bg * mS - (with_mask ? mS * bsum[k] : T(0));
                if (wave == NW - 1 && k == STRIP - 1) cGdd[lane] = g;
                sGdd[1 + wrow + k][lane] = g;
                fsum[0] += (own_x && rs >= ys && rs < ye) ? bsum[k] * coef_of(vS) : T(0);
              }
            }
            sched_fence();
          }
        }
        __syncthreads();
        if (c < 2 && w_on) {  // the next colour's target values (rows and column of stage W; the row offsets are scalars)
#pragma unroll
          for (int k = 0; k < STRIP; ++k)
            xt[k] = ld_at(tgt_img + (c + 1) * plane + unsigned(reflect_index(rw0 + k, H)) * unsigned(W), unsigned(u) * unsigned(sizeof(T)));
        }
        if (o_on) {
          T gt[STRIP][3];
          band_box_transpose<T, STRIP, RS, 3>(sG, wrow, cl, lane, cr, px, ro0, H, W, gt);
#pragma unroll
          for (int k = 0; k < STRIP; ++k) {
            const T x = cen[k][0], y = cen[k][1], d = x - y;
            // d clamp(|d|, 0, 1) / d Iw: the clamp passes gradient on [0, 1] inclusive, abs uses sgn
            const T l1g = (t_abs(d) <= T(1)) ? -t_sgn(d) : T(0);
            gI[k][c] = gt[k][0] + T(2) * y * gt[k][1] + x * gt[k][2] + coef_of(sC[wrow + k][lane]) * T(0.15) * l1g;
          }
        }
      }
    }
    __syncthreads();  // the last transposed box filter has read sG: the window may go there; sBox is complete
    // ---------------- geometry tail: rows a - LAG + wave STRIP + k ---------------------------------------------
    for (int i = threadIdx.x; i < WW * WH; i += kThreads) (&win[0][0])[i] = Cell(0);
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
    const int ro0 = a - LAG + wave * STRIP;
        const bool t_on = ro0 + STRIP - 1 >= ys && ro0 < ye && !(flags & SCSFM_DEBUG_X4);
    T d_own[STRIP], gdd[STRIP], gd[STRIP];
    if (t_on) {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int cy = ro0 + k < H ? (ro0 + k < 0 ? 0 : ro0 + k) : H - 1;
        d_own[k] = tgt_depth.at(cxo, cy, (unsigned(cy) * unsigned(W) + unsigned(cxo)) * unsigned(sizeof(T)));
        if constexpr (kSsim) gdd[k] = sGdd[wrow + k][lane]; else gdd[k] = gdd_own[k];
      }
    }
    __syncthreads();  // the window's zeroes
    if (t_on) {
      T acc[12];  // pose partials of this chunk's owned pixels
#pragma unroll
      for (int i = 0; i < 12; ++i) acc[i] = T(0);
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int ro = ro0 + k;
        gd[k] = T(0);
        sched_fence();  // one pixel's sampling state at a time (register budget: four workgroups per CU)
        if (!(own_x && ro >= ys && ro < ye)) continue;
        // (u == px for a pixel inside the image: the projection's column part is shared with stage W)
        gd[k] = geom_pixel<T, Cell, WW, WH>(bc, u, ro, d_own[k], gI[k], gdd[k], ref_img, ref_depth, plane, H, W, flags, win,
                                            wx0, wy0, g_scatter, acc);
      }
      // twelve registers that would otherwise live through every stage of every chunk: summed over the wave here
      // (N + 6 shuffles for the lot) and kept in LDS, in fp64, one row per wave (no atomics: a wave owns its row)
      bool lead;
      const int idx = wave_sum_packed<12>(acc, lead);
      if (lead) sAcc[wave][idx] += double(acc[0]);
    }
    __syncthreads();  // the scatter's LDS atomics precede the flush
    if (t_on) {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int ro = ro0 + k;
        if (own_x && ro >= ys && ro < ye) st_at(g_dense, (unsigned(ro) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), gd[k]);
      }
    }
    if (!(flags & (SCSFM_DEBUG_X1 | SCSFM_DEBUG_X5)))
      flush_scatter_region<T, Cell, WW, WH>(win, wx0, wy0, cx0, cy0, cx1, cy1, g_scatter, W);
    // (the next chunk writes sBox, the planes of stage W and -- behind its first barrier -- sG / the window: nothing the
    // flush reads is touched before every thread has passed that barrier)
  }
  // ---------------- the segment's sums ------------------------------------------------------------------------
  __syncthreads();
  block_sum<3>(fsum, red);
  if (threadIdx.x == 0) {
    double* o = pa.partials + 3 * ((size_t)(b * nsegs + blk.y) * nbands + blk.x);
    o[0] = double(fsum[0]); o[1] = double(fsum[1]); o[2] = double(fsum[2]);
  }
  if (threadIdx.x == 0) {  // (block_sum's barrier orders the waves' last additions to sAcc before this)
    double* o = pa.gPp + 12 * ((size_t)(b * nsegs + blk.y) * nbands + blk.x);
    double g[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      g[i] = 0.0;
      for (int w = 0; w < NW; ++w) g[i] += sAcc[w][i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = g[i];
  }
}

}  // namespace scsfm
