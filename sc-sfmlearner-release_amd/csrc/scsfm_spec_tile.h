// The speculative forward of a pair-direction as one 64 x 16 tile per workgroup (the product of rounds 1 and 2): warp in a
// 66 x 18 domain, forward statistics at every pixel of the 64 x 16 domain, transposed box filter and geometry tail for
// the 62 x 14 interior.  Four workgroups per CU (128 VGPRs, 40 KB of LDS: kLean).  The column march of variants/src/scsfm_march.h
// issues 15-20 % fewer vector instructions per output pixel but was measured slower on MI355X: what bounds
// both kernels is the chain of dependent memory / LDS round trips and barriers of a workgroup, and short-lived tiles at
// four per CU overlap those chains better than long-lived segments at two or three.
//
// Reference lines: loss_functions.py:95-119 (compute_pairwise_loss), :11-42 (SSIM), inverse_warp.py:230-269.
#pragma once
#include "scsfm_geom.h"
#include "scsfm_ssim.h"
#include "scsfm_smooth_math.h"
#ifdef SCSFM_WITH_MARCH  // tuning / test builds (-Ivariants/src): the staged-forward variant's tap reader
#include "scsfm_stagefwd_taps.h"
#endif

namespace scsfm {

// Nothing moves across this point when the compiler schedules the instructions (loads issued before it stay before
// everything that consumes them).
__device__ __forceinline__ void sched_fence() {
#if defined(__HIP_DEVICE_COMPILE__)
  __builtin_amdgcn_sched_barrier(0);
#endif
}

#ifndef SCSFM_LEAN_LDS  // tuning knob: 1 = the speculative forward in 40 KB of LDS (see kLean), 0 = 53.6 KB
#define SCSFM_LEAN_LDS 1
#endif
#ifndef SCSFM_STAGE_TAPS  // tuning knob: 0 = the tail gathers its taps from global memory
#define SCSFM_STAGE_TAPS 1
#endif

template <typename T, bool kSsim, bool kScaled, unsigned kFlags, bool kStageFwd = false>
// (kSpec: always true here -- the backward's own tiled pass is photo_tile in scsfm_pair.hip)
__device__ __forceinline__ void spec_tile(const BlockId blk, int nbx, int nby, const PairBatch<T>& pb, int B, int H,
                                           int W, unsigned flags_arg, const T* __restrict__ g_photo,
                                           const T* __restrict__ g_geom, T r_hint, T sm_icx = T(0), T sm_icy = T(0)) {
  constexpr bool kSpec = true;  // (the body is shared history with the backward's tiled pass: its !kSpec branches are dead here)
  const unsigned flags = kFlags == kRuntimeFlags ? flags_arg : kFlags;
  const int pair = blk.z / B, b = blk.z - pair * B;
  const PairArgs<T>& pa = pb.p[pair];
  const T* __restrict__ tgt_img = pa.tgt_img;
  const T* __restrict__ ref_img = pa.ref_img;
  const BatchConsts<T>* __restrict__ consts = pa.consts;
  const double* __restrict__ sums = pa.sums;
  T* __restrict__ gbuf = pa.gbuf;
  double* __restrict__ partials = pa.partials;
  typedef typename Vec2<T>::type V2;
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ V2 sXY[kSsim ? 3 : 1][kSsim ? TH + 2 : 1][kSsim ? kHaloW : 1];
  __shared__ T sG[kSsim ? 3 : 1][kSsim ? TH : 1][kSsim ? kTileW : 1];  // 1/9 (g_mu_y, g_E[y^2], g_E[xy]), one colour
  // kSpec: staging window of the geometry tail's scatter (its height follows the tile's)
  constexpr int WW = kWinW, WH = kWinH * TH / kTileH;
  typedef typename WinCell<T>::type Cell;
  // kLean (SCSFM_LEAN_LDS, fp32 + SSIM speculative forward): 40 KB of LDS instead of 53.6 KB, so that a CU holds four
  // workgroups: the window lives in sG (dead once the SSIM phases end; zeroed at the start of the tail), the
  // reduction scratch behind the parked gradients and the staged colours in tile 0
  constexpr bool kLean = SCSFM_LEAN_LDS && kSpec && kSsim && sizeof(T) == 4 && TH == kTileH;
  constexpr int kStageRows = kLean ? kStageH - 1 : kStageH;
  static_assert(!kLean || sizeof(Cell) * WW * WH <= sizeof(T) * 3 * TH * kTileW, "window in sG");
  __shared__ double red_own[(kSpec && !kLean) ? (3 + 12) * (kThreads / kWave) : 1];  // the two block sums use disjoint parts
  __shared__ Cell win_own[(kSpec && !kLean) ? WH : 1][(kSpec && !kLean) ? WW : 1];
  double* const red = kLean ? reinterpret_cast<double*>(reinterpret_cast<T*>(&sXY[0][0][0]) + TH * kTileW + kStageW * kStageRows)
                            : &red_own[0];
  Cell(*const win)[WW] = kLean ? reinterpret_cast<Cell(*)[WW]>(&sG[0][0][0]) : reinterpret_cast<Cell(*)[WW]>(&win_own[0][0]);
  if constexpr (kSpec && !kLean) {  // zeroed long before its first use (several barriers lie in between)
    for (int i = threadIdx.x; i < WW * WH; i += kThreads) (&win[0][0])[i] = Cell(0);
  }

  // upstream gradient x d(masked mean)/d(sum): zero when the 10000-pixel gate was closed
  T a = T(1), bg = r_hint;
  if constexpr (!kSpec) {
    a = T(sums[5]) * g_photo[0];
    bg = T(sums[6]) * g_geom[0];
    if (a == T(0) && bg == T(0)) return;        // workgroup-uniform: pass B skips as well
    if (spec_valid(sums, g_photo, g_geom)) return;  // the forward already left the planes in gbuf
  }

  // The wave index twice: as a scalar (row indices, row predicates and reflections become scalar arithmetic, `if (row
  // ...)` scalar branches) and as a vector register for LDS addresses (a DS instruction adds an immediate to ONE address
  // register: (wave's first row, column) is formed once and every row / plane is an immediate away).
  const int col = threadIdx.x & (kWave - 1), strip = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave);
  const int lrow = (int(threadIdx.x) / kWave) * STRIP;
  // the 64 x TH compute domain starts one pixel before the 62 x (TH-2) block of outputs
  const int ox = blk.x * (kTileW - 2) - 1, oy = blk.y * (TH - 2) - 1;
  const bool with_mask = (flags & SCSFM_WITH_MASK) != 0, with_auto = (flags & SCSFM_WITH_AUTO_MASK) != 0;
  const BatchConsts<T> bc = consts[b];
  const unsigned plane = unsigned(H) * unsigned(W);
  const size_t gplane = (size_t)B * plane;  // one gbuf plane spans the whole batch
  tgt_img += (size_t)b * 3 * plane;
  ref_img += (size_t)b * 3 * plane;
  const Planes3<T> tgtP = planes3(tgt_img, plane), refP = planes3(ref_img, plane);
  const DepthMap<T, kScaled> tgt_depth = depth_map<kScaled>(pa.tgt_depth, b, H, W, pa.ds);
  const DepthMap<T, kScaled> ref_depth = depth_map<kScaled>(pa.ref_depth, b, H, W, pa.ds);
  gbuf += (size_t)b * plane;

  const int px = ox + col, py0 = oy + strip * STRIP;
  const bool in_x = col >= 1 && col <= kTileW - 2 && px < W;
  T coef[STRIP];  // a * m * (1 - dd): weight of blend_c(q) in the loss
  T mq[STRIP];    // mask of the owned pixel
  T bsum[STRIP];  // sum_c blend_c of the owned pixel
  T acc_g = T(0), acc_m = T(0);  // kSpec: forward sums over the pixels this block owns
  int bx0 = 1 << 30, bx1 = -(1 << 30), by0 = 1 << 30, by1 = -(1 << 30);
  __shared__ int sBox[kSpec ? kThreads / kWave : 1][4];
  // pairs that carry their target frame's smooth loss (pa.sm_partials; uniform per pair): the waves' three sums
  __shared__ T sSm[kThreads / kWave][4];
  // fixed-point scatter cells: the bound of what this tile can add to one of them (scsfm_geom.h: win_units_of)
  constexpr bool kFixed = sizeof(typename WinCell<T>::type) == 4 && sizeof(T) == 4;
  __shared__ float sU[kThreads / kWave];
  float ub = 0.0f;
  // |dL/d diff_depth| <= |r| m + [mask] a m sum_c blend_c <= (|r| + 3) m; times 2 for d diff_depth / d D_p = 2 Z / (Z + D_p)^2
  const float ub_coef = 2.0f * (float(t_abs(r_hint)) + (with_mask ? 3.0f : 0.0f));
  V2 cen[kSsim ? 1 : STRIP][kSsim ? 1 : 3];
  // kSpec: dL/d(warped colour c) of the owned pixels waits for the geometry tail -- parked in the LDS tile of
  // colour c, which is dead by the time that gradient exists (every thread only touches its own slots); in
  // registers without SSIM
  static_assert(!kSsim || sizeof(V2) * (TH + 2) * kHaloW >= sizeof(T) * TH * kTileW, "parking space");
  T gI_reg[(kSpec && !kSsim) ? STRIP : 1][3];
#ifdef PROBE_TIMING
  const int probe_wg = ((blk.z * nby + blk.y) * nbx + blk.x) / 3, probe_chunk = ((blk.z * nby + blk.y) * nbx + blk.x) % 3 == 0 ? 0 : 99;
#endif
  STAMP(0);
  T in_d[STRIP];
#ifdef SCSFM_WITH_MARCH  // tuning / test builds only: the forward warp with LDS-staged taps (variants/src/; measured slower)
  constexpr bool kStageF = kStageFwd && kSpec && kSsim && sizeof(T) == 4 && TH == kTileH;
  if constexpr (kStageF) {
#include "scsfm_spec_stagefwd.inc"
  } else
#endif
  {
    // ---- phase 0: every streaming load of the strip and of this thread's ring pixel ---------------
    const int u = reflect_index(px, W);
    T in_t[STRIP][3], in_r[STRIP][3], rin_d = T(0), rin_t[3] = {T(0), T(0), T(0)};
  #pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      const int v = reflect_index(py0 + k, H);
      const unsigned off = (unsigned(v) * unsigned(W) + unsigned(u)) * unsigned(sizeof(T));
      in_d[k] = tgt_depth.at(u, v, off);
  #pragma unroll
      for (int c = 0; c < 3; ++c) in_t[k][c] = ld_plane(tgtP, c, off);
  #pragma unroll
      for (int c = 0; c < 3; ++c) in_r[k][c] = T(0);
      if (with_auto) {
  #pragma unroll
        for (int c = 0; c < 3; ++c) in_r[k][c] = ld_plane(refP, c, off);
      }
    }
    const bool has_ring = kSsim && threadIdx.x < 2 * kHaloW + 2 * TH;
    int ru = 0, rv = 0, rhy = 0, rhx = 0;
    if (has_ring) {
      ring_pos<TH>(threadIdx.x, rhy, rhx);
      ru = reflect_index(ox + rhx - 1, W); rv = reflect_index(oy + rhy - 1, H);
      const unsigned off = (unsigned(rv) * unsigned(W) + unsigned(ru)) * unsigned(sizeof(T));
      rin_d = tgt_depth.at(ru, rv, off);
  #pragma unroll
      for (int c = 0; c < 3; ++c) rin_t[c] = ld_plane(tgtP, c, off);
    }
    // ---- the target frame's smooth loss (loss_functions.py:133-152), for the pair that carries it (uniform per pair) ----
    // Depth and colours of the strip are in registers here; the rows above and below it are two more loads each.  One pass
    // yields the strip's part of sum D and of both edge sums, and the per-pixel edge terms the backward streams; the
    // waves' three sums meet in LDS and leave with the block sum's store further down (one record per tile).  Evaluating
    // it behind the flush instead (nothing live, everything re-read, a record per wave) measured 0.4315-0.4365 ms per
    // loss-path step against 0.4275-0.433 here and 0.441-0.443 with the stand-alone smooth forward
    // (profiles/r06_kernel_experiments.json; the variant: variants/src/patches/r06_smooth_behind_flush.patch).
    if (pa.sm_partials != nullptr) {
      Px<T> row[STRIP], up, dn;
      bool own_row[STRIP];
  #pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        row[k].d = in_d[k]; row[k].c0 = in_t[k][0]; row[k].c1 = in_t[k][1]; row[k].c2 = in_t[k][2];
        const int ly = strip * STRIP + k;
        own_row[k] = ly >= 1 && ly <= TH - 2;
      }
      {
        const int vu = t_clampi(py0 - 1, 0, H - 1), vd = t_clampi(py0 + STRIP, 0, H - 1);
        const unsigned ou = (unsigned(vu) * unsigned(W) + unsigned(u)) * unsigned(sizeof(T));
        const unsigned od = (unsigned(vd) * unsigned(W) + unsigned(u)) * unsigned(sizeof(T));
        up.d = tgt_depth.at(u, vu, ou); up.c0 = ld_plane(tgtP, 0, ou); up.c1 = ld_plane(tgtP, 1, ou); up.c2 = ld_plane(tgtP, 2, ou);
        dn.d = tgt_depth.at(u, vd, od); dn.c0 = ld_plane(tgtP, 0, od); dn.c1 = ld_plane(tgtP, 1, od); dn.c2 = ld_plane(tgtP, 2, od);
      }
      T sm[3] = {T(0), T(0), T(0)};
      smooth_strip<T, STRIP>(row, up, dn, px, py0, in_x, own_row, sm_icx, sm_icy, H, W,
                             pa.sm_edge ? pa.sm_edge + (size_t)b * plane : nullptr, sm);
  #pragma unroll
      for (int i = 0; i < 3; ++i) sm[i] = wave_sum_last(sm[i]);
      if (col == kWave - 1) { sSm[strip][0] = sm[0]; sSm[strip][1] = sm[1]; sSm[strip][2] = sm[2]; }
    }
    // ---- phase 1a ------------------------------------------------------------------------------
    // SCSFM_W_GROUP pixels' gathers are in flight together (tools/march_timing.py: with one pixel after the other this
    // phase took 12,400 of a tile's 48,000 cycles, a third of it vector instructions).  2 since the image loads are
    // buffer loads: -1.6 % (323 against 328 us per launch, three alternating runs each); 4 spills 12 registers.
  #ifndef SCSFM_W_GROUP
  #define SCSFM_W_GROUP 2
  #endif
    constexpr int WG_ = SCSFM_W_GROUP < STRIP ? SCSFM_W_GROUP : STRIP;
  #pragma unroll
    for (int k0 = 0; k0 < STRIP; k0 += WG_) {
      Sample<T> sm[WG_];
      TapRows<T> tc[WG_][3], td[WG_];
  #pragma unroll
      for (int j = 0; j < WG_; ++j) {
        const int k = k0 + j;
        sm[j] = project_pixel(bc, u, reflect_index(py0 + k, H), in_d[k], H, W, flags);
  #pragma unroll
        for (int c = 0; c < 3; ++c) tc[j][c] = load_tap_rows(refP, c, sm[j]);
        td[j] = ref_depth.taps(sm[j]);
      }
      if (WG_ > 1) sched_fence();
  #pragma unroll
      for (int j = 0; j < WG_; ++j) {
        const int k = k0 + j;
        const int ly = strip * STRIP + k, py = py0 + k;
        const bool inimg = px >= 0 && px < W && py >= 0 && py < H;
        const Sample<T>& s = sm[j];
        V2 xy[3];
  #pragma unroll
        for (int c = 0; c < 3; ++c) xy[c] = make2(in_t[k][c], bilerp_rows(tc[j][c], s));
  #pragma unroll
        for (int c = 0; c < 3; ++c) {
          if constexpr (kSsim) sXY[c][lrow + k + 1][col + 1] = xy[c]; else cen[k][c] = xy[c];
        }
        const T Dp = bilerp_rows(td[j], s);
        const T isum = t_rcp(s.Z + Dp);
        const T ddk = clamp01(t_abs(s.Z - Dp) * isum);
        mq[k] = inimg ? pixel_mask(s, with_auto, xy, in_r[k]) : T(0);
        coef[k] = a * mq[k] * (with_mask ? (T(1) - ddk) : T(1));
        bsum[k] = T(0);
        if constexpr (kSpec) {
          if (in_x && ly >= 1 && ly <= TH - 2 && py < H) {
            acc_g += ddk * mq[k]; acc_m += mq[k];
            if (mq[k] != T(0)) {  // this pixel will scatter: where its north-west tap lies
              bx0 = s.xa < bx0 ? s.xa : bx0; bx1 = s.xa > bx1 ? s.xa : bx1;
              by0 = s.ya < by0 ? s.ya : by0; by1 = s.ya > by1 ? s.ya : by1;
              // ... and at most how much (NaN-safe: min returns the cap)
              if constexpr (kFixed) ub += t_min(float(ub_coef * float(s.Z) * float(isum) * float(isum)), kFixCap);
            }
          }
        }
      }
    }
    if constexpr (kSpec) {  // bounding box of the block's scatter footprint: per wave here, met in the tail
  #pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) {
        const int a0 = __shfl_xor(bx0, o), a1 = __shfl_xor(bx1, o), c0 = __shfl_xor(by0, o), c1 = __shfl_xor(by1, o);
        bx0 = a0 < bx0 ? a0 : bx0; bx1 = a1 > bx1 ? a1 : bx1; by0 = c0 < by0 ? c0 : by0; by1 = c1 > by1 ? c1 : by1;
      }
      if (col == 0) { sBox[strip][0] = bx0; sBox[strip][1] = bx1; sBox[strip][2] = by0; sBox[strip][3] = by1; }
      if constexpr (kFixed) {
        ub = wave_sum_last(ub);
        if (col == kWave - 1) sU[strip] = ub;
      }
    }
    STAMP(1);
    // ---- phase 1b: ring ------------------------------------------------------------------------
    if constexpr (kSsim) {
      if (has_ring) {
        const Sample<T> rs = project_pixel(bc, ru, rv, rin_d, H, W, flags);
  #pragma unroll
        for (int c = 0; c < 3; ++c) sXY[c][rhy][rhx] = make2(rin_t[c], bilerp_rows(load_tap_rows(refP, c, rs), rs));
      }
      STAMP(2);
      __syncthreads();
      STAMP(3);
    }
  }
  // kSpec: the window goes where the block's pixels land: around the bounding box of their north-west taps (known
  // since the warp), centred on it when it is larger than the window (the rest falls back to global atomics)
  int wx0 = 0, wy0 = 0, cx0 = 0, cy0 = 0, cx1 = 0, cy1 = 0;  // window origin; cells of the window the taps can reach
#ifndef SCSFM_WIDE_WINDOW  // tuning knob: 0 = no wide window for incoherent footprints
#define SCSFM_WIDE_WINDOW 1
#endif
#ifndef SCSFM_WIDE_EY  // footprints taller than this many rows take the wide window.  Round 3: 2 * kWinH; round 4: kWinH -- a
#define SCSFM_WIDE_EY kWinH  // tile whose taps spread over more rows than the window has finds half of them outside it (a direct
#endif                       // global atomic each) and half of its texels outside the 17 staged rows anyway: -2 %
  // (the wide window needs the lean layout -- its extension rows are the staging regions -- and 32-bit cells)
  constexpr bool kWideOk = SCSFM_WIDE_WINDOW && SCSFM_LEAN_LDS && SCSFM_STAGE_TAPS && kSpec && kSsim && sizeof(T) == 4 && TH == kTileH;
  int wrows = WH;
  float fix_scale = kFixScale, fix_inv = kFixInv;  // (fixed-point cells only)
  auto scatter_box = [&]() {
    int x0 = sBox[0][0], x1 = sBox[0][1], y0 = sBox[0][2], y1 = sBox[0][3];
#pragma unroll
    for (int w = 1; w < kThreads / kWave; ++w) {
      x0 = sBox[w][0] < x0 ? sBox[w][0] : x0; x1 = sBox[w][1] > x1 ? sBox[w][1] : x1;
      y0 = sBox[w][2] < y0 ? sBox[w][2] : y0; y1 = sBox[w][3] > y1 ? sBox[w][3] : y1;
    }
    if (x0 > x1) { x0 = x1 = 0; y0 = y1 = 0; }  // nothing scatters
    const int ex = x1 - x0 + 2, ey = y1 - y0 + 2;  // cells touched (each pixel reaches one past its tap)
    // a footprint taller than SCSFM_WIDE_EY rows: the wide window (scsfm_geom.h: WideWin), no staged taps
    wrows = (kWideOk && ey > SCSFM_WIDE_EY) ? WH + 3 * kWideRows : WH;
    wx0 = ex <= WW ? x0 - (WW - ex) / 2 : (x0 + x1 + 1) / 2 - WW / 2;
    wy0 = ey <= wrows ? y0 - (wrows - ey) / 2 : (y0 + y1 + 1) / 2 - wrows / 2;
    cx0 = x0 - wx0; cx1 = x1 + 1 - wx0; cy0 = y0 - wy0; cy1 = y1 + 1 - wy0;
    // the unit of the window's cells follows from the tile's bound (every thread evaluates the same four values)
    if constexpr (kFixed) {
      float U = sU[0];
#pragma unroll
      for (int w = 1; w < kThreads / kWave; ++w) U += sU[w];
      win_units_of(U, fix_scale, fix_inv);
    }
  };
  // kStage (fp32 + SSIM): the texels the geometry tail samples -- the reference view's colours and depth around
  // where the tile lands -- are staged in LDS at the start of the tail: the colour planes behind the parked
  // gradients in the (then dead) tiles, the depth plane in sG.  (Requesting them here, so that the round trip hides
  // under the SSIM phases, was measured: the 24 registers held across those phases cost more than the latency.)
  constexpr bool kStage = SCSFM_STAGE_TAPS && kSpec && kSsim && sizeof(T) == 4 && TH == kTileH;
  constexpr int NR = (kStageRows + kThreads / kWave - 1) / (kThreads / kWave), XW = kStageW - kWave;
  constexpr int kTileFloats = int(sizeof(V2) / sizeof(T)) * (TH + 2) * kHaloW;  // one colour's tile
  static_assert(!kStage || kTileFloats >= TH * kTileW + kStageW * kStageRows + (kLean ? 2 * (3 + 12) * (kThreads / kWave) : 0),
                "staging space (colours, + the reduction scratch in lean mode)");
  static_assert(!kStage || 3 * TH * kTileW >= kStageW * kStageRows, "staging space (depth)");
  T* const sp_colour = reinterpret_cast<T*>(&sXY[0][0][0]) + TH * kTileW;
  T* const sp_depth = &sG[0][0][0];
  StagedTaps<T> staged;
  T stage_v[kStage ? 4 : 1][kStage ? NR + 1 : 1];
  const int er = threadIdx.x / XW, ec = kWave + threadIdx.x - er * XW;  // the columns beyond 64: (row, column) of this thread
  if constexpr (kSpec && kSsim) scatter_box();
  // The staging loads of the tail (used below).  SCSFM_STAGE_EARLY issues them before the last colour's transposed box
  // filter, so that their round trip (~3,300 cycles of a tile's 47,000 by the stage timeline) runs under that phase
  // instead of in front of the tail; the price is their ~20 registers across it.  (A macro, not a lambda: capturing
  // the staging arrays by reference kept them in scratch memory.)
#ifndef SCSFM_STAGE_EARLY
#define SCSFM_STAGE_EARLY 0
#endif
#define SCSFM_ISSUE_STAGE_LOADS() \
  do { \
      const int bx = wx0 + cx0, by = wy0 + cy0, ex = cx1 - cx0 + 1, ey = cy1 - cy0 + 1; \
      int sx0 = bx - (kStageW - ex) / 2, sy0 = by - (kStageRows - ey) / 2; \
      sx0 = sx0 > W - kStageW ? W - kStageW : sx0; sx0 = sx0 < 0 ? 0 : sx0; \
      sy0 = sy0 > H - kStageRows ? H - kStageRows : sy0; sy0 = sy0 < 0 ? 0 : sy0; \
      staged.x0 = sx0; staged.y0 = sy0; \
      staged.colour = sp_colour; staged.depth = sp_depth; staged.stride = kTileFloats; \
      const int gx = sx0 + col < W ? sx0 + col : W - 1; \
      const int egx = sx0 + ec < W ? sx0 + ec : W - 1, egy = sy0 + er < H ? sy0 + er : H - 1; \
  _Pragma("unroll") \
      for (int i = 0; i <= NR; ++i) { \
        const int r = strip + i * (kThreads / kWave); \
        const int gy = sy0 + r < H ? sy0 + r : H - 1; \
        const int x = i < NR ? gx : egx, y = i < NR ? gy : egy; \
        const unsigned off = (unsigned(y) * unsigned(W) + unsigned(x)) * unsigned(sizeof(T)); \
        const bool on = i < NR ? r < kStageRows : threadIdx.x < XW * kStageRows; \
  _Pragma("unroll") \
        for (int c = 0; c < 4; ++c) stage_v[c][i] = T(0); \
        if (on) { \
  _Pragma("unroll") \
          for (int c = 0; c < 3; ++c) stage_v[c][i] = ld_plane(refP, c, off); \
          if (!kLean) stage_v[3][i] = ref_depth.at(x, y, off); \
        } \
      } \
  } while (0)
  // ---- phases 2/3, one colour channel at a time ------------------------------------------------
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    T gI[STRIP];
    if constexpr (kSsim) {
      // phase 2: forward statistics at every owned pixel q; publish 1/9 (g_mu_y, g_E[y^2], g_E[xy])(q)
      WinSums<T> ws[STRIP];
      V2 centre[STRIP];
      strip_window_sums<T, STRIP>(sXY[c], lrow, col, ws, centre);
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const SsimStats<T> st = ssim_stats(ws[k]);
        bsum[k] += T(0.85) * clamp01(st.raw);
        // s = clamp((1 - S)/2, 0, 1): d s / d S = -1/2 inside the clamp (inclusive bounds)
        const T gS = clamp01(st.raw) == st.raw ? coef[k] * T(0.85) * T(-0.5) : T(0);  // (i.e. 0 <= raw <= 1)
        T g1, g2, g3;
        ssim_grad_y(st, gS, g1, g2, g3);
        sG[0][lrow + k][col] = g1; sG[1][lrow + k][col] = g2; sG[2][lrow + k][col] = g3;
      }
      __syncthreads();
      if constexpr (kStage && SCSFM_STAGE_EARLY == 1) {
        if (c == 2) SCSFM_ISSUE_STAGE_LOADS();
      }
      // phase 3: transpose of (reflect-pad + 3x3 box) as a separable 3x3 gather
      T gt[STRIP][3];
      strip_box_transpose<T, STRIP, TH, 3>(sG, lrow, col, px, py0, H, W, gt);
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const T x = centre[k][0], y = centre[k][1], d = x - y;
        bsum[k] += T(0.15) * clamp01(t_abs(d));
        // d clamp(|d|, 0, 1) / d Iw: the clamp passes gradient on [0, 1] inclusive, abs uses sgn
        const T l1g = (t_abs(d) <= T(1)) ? -t_sgn(d) : T(0);
        gI[k] = gt[k][0] + T(2) * y * gt[k][1] + x * gt[k][2] + coef[k] * T(0.15) * l1g;
      }
      if (c < 2) __syncthreads();  // sG is rewritten by the next colour
    } else {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const T d = cen[k][c][0] - cen[k][c][1];
        bsum[k] += clamp01(t_abs(d));
        gI[k] = coef[k] * ((t_abs(d) <= T(1)) ? -t_sgn(d) : T(0));
      }
    }
#pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      const int ly = strip * STRIP + k, py = py0 + k;
      // note: m(p) = 0 still receives SSIM gradient through its neighbours' windows
      if constexpr (kSpec) {
        if constexpr (kSsim) reinterpret_cast<T*>(&sXY[c][0][0])[(lrow + k) * kTileW + col] = gI[k]; else gI_reg[k][c] = gI[k];
      } else {
        if (in_x && ly >= 1 && ly <= TH - 2 && py < H)
          st_at(gbuf + (kPlaneGI + c) * gplane, (unsigned(py) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), gI[k]);
      }
    }
  }
  // dL/d diff_depth: directly (geometry loss) and through the weight mask (no detach, loss_functions.py:111-113)
  T gdd[STRIP];
#pragma unroll
  for (int k = 0; k < STRIP; ++k) gdd[k] = bg * mq[k] - (with_mask ? a * mq[k] * bsum[k] : T(0));
  if constexpr (!kSpec) {
    // hand over to pass B; this tile's part of the pair's scatter plane is cleared on the way (pass B only
    // runs when this pass did)
#pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      const int ly = strip * STRIP + k, py = py0 + k;
      if (in_x && ly >= 1 && ly <= TH - 2 && py < H) {
        const unsigned off = (unsigned(py) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T));
        st_at(gbuf + kPlaneGdd * gplane, off, gdd[k]);
        st_at(gbuf + kPlaneScatter * gplane, off, T(0));
      }
    }
  } else {
    if constexpr (kStage && SCSFM_STAGE_EARLY == 2) SCSFM_ISSUE_STAGE_LOADS();  // (in flight under the block sum below)
    // ---- the forward's three sums over the pixels this block owns ---------------------------------
    T v[3] = {T(0), acc_g, acc_m};
#pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      const int ly = strip * STRIP + k, py = py0 + k;
      // with a = 1, coef = m * (1 - dd) (or m): exactly the weight of blend in the photo sum
      if (in_x && ly >= 1 && ly <= TH - 2 && py < H) v[0] += bsum[k] * coef[k];
    }
    STAMP(4);
    // (contains a barrier: the window's zeroes are visible below even without SSIM)
    block_sum_store<3>(v, red, partials + 3 * ((size_t)(b * nby + blk.y) * nbx + blk.x));
    if (pa.sm_partials != nullptr && threadIdx.x < 3) {  // (sSm was written in phase 0: several barriers ago)
      double t = 0;
#pragma unroll
      for (int w = 0; w < kThreads / kWave; ++w) t += double(sSm[w][threadIdx.x]);
      pa.sm_partials[3 * ((size_t)(b * nby + blk.y) * nbx + blk.x) + threadIdx.x] = t;
    }
    STAMP(5);
    // ---- geometry tail: pass B for the owned pixels, up to the factor the reduction will supply ----
    // (everything downstream of dL/d(warped colour), dL/d diff_depth is linear in them: the dense plane, the
    // scatter plane and the pose partials are all scaled by a = g_photo / (3 S_m) when they are combined)
    T* __restrict__ g_dense = pa.gbuf + kPlaneDense * gplane + (size_t)b * plane;
    T* __restrict__ g_scatter = pa.gbuf + kPlaneScatter * gplane + (size_t)b * plane;
    if constexpr (!kSsim) scatter_box();  // (with SSIM: done after the warp phase's barrier)
    const bool wide = kWideOk && __builtin_amdgcn_readfirstlane(wrows) > WH;  // (uniform: every thread read the same boxes)
    if constexpr (kStage && !SCSFM_STAGE_EARLY) {
      if (!wide) SCSFM_ISSUE_STAGE_LOADS();
    }
    T d_own[STRIP];  // depth of the owned pixels: kept since phase 0, or (kLean: 4 registers less across the SSIM phases) re-read
#pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      if constexpr (kLean) {
        const int cy = py0 + k < H ? (py0 + k < 0 ? 0 : py0 + k) : H - 1, cx = px < W ? (px < 0 ? 0 : px) : W - 1;
        d_own[k] = tgt_depth.at(cx, cy, (unsigned(cy) * unsigned(W) + unsigned(cx)) * unsigned(sizeof(T)));
      } else {
        d_own[k] = in_d[k];
      }
    }
    if constexpr (kLean) {  // the window (in sG, dead since the barrier of the block sum above)
      for (int i = threadIdx.x; i < WW * WH; i += kThreads) (&win[0][0])[i] = Cell(0);
    }
    WideWin ww{WH, 0, 0};
    if constexpr (kStage) {
      if (wide) {
        // the staging regions become window rows: cleared, and no block of the tail lies "inside" a staged window
        static_assert(!kWideOk || (kStageW * kStageRows >= kWideRows * WW && sizeof(Cell) == sizeof(T)), "wide window rows");
        for (int i = threadIdx.x; i < 3 * kWideRows * WW; i += kThreads) {
          const int r = i / (kWideRows * WW);
          reinterpret_cast<Cell*>(sp_colour)[r * kTileFloats + (i - r * kWideRows * WW)] = Cell(0);
        }
        staged.x0 = 1 << 28; staged.y0 = 1 << 28;
        staged.colour = sp_colour; staged.depth = sp_depth; staged.stride = kTileFloats;
        ww.rows = WH + 3 * kWideRows;
        ww.off = int(reinterpret_cast<Cell*>(sp_colour) - &win[0][0]) - WH * WW;
        ww.step = kTileFloats - kWideRows * WW;
      } else {
      // ... and go to LDS: the tiles and sG are dead by now
#pragma unroll
      for (int i = 0; i <= NR; ++i) {
        const int r = i < NR ? strip + i * (kThreads / kWave) : er, cc = i < NR ? col : ec;
        const bool on = i < NR ? r < kStageRows : threadIdx.x < XW * kStageRows;
        if (on) {
#pragma unroll
          for (int c = 0; c < 3; ++c) sp_colour[c * kTileFloats + r * kStageW + cc] = stage_v[c][i];
          if (!kLean) sp_depth[r * kStageW + cc] = stage_v[3][i];
        }
      }
      }
      __syncthreads();
    }
    STAMP(6);
    T acc[12], gd[STRIP];
#pragma unroll
    for (int i = 0; i < 12; ++i) acc[i] = T(0);
    // debugging launches count the wraps of the window's fixed-point cells (scsfm_geom.h: win_add); flags is a
    // compile-time constant without that bit in the product instantiation, so this is a constant nullptr there
    unsigned* const ovf = (flags & SCSFM_DEBUG_CHECK_WINDOW) ? window_overflow_counter(pa) : nullptr;
#pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      const int ly = strip * STRIP + k, py = py0 + k;
      gd[k] = T(0);
      if (!(in_x && ly >= 1 && ly <= TH - 2 && py < H) || (flags & SCSFM_DEBUG_X4)) continue;
      T gI[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        if constexpr (kSsim) gI[c] = reinterpret_cast<const T*>(&sXY[c][0][0])[(lrow + k) * kTileW + col]; else gI[c] = gI_reg[k][c];
      }
      if constexpr (kStage) {
        const GeomTaps<T> f = geom_fetch<kStageRows, !kLean>(bc, px, py, d_own[k], refP, ref_depth, H, W, flags, staged);
        gd[k] = geom_consume<T, Cell, WW, WH>(bc, f, px, py, d_own[k], gI, gdd[k], H, W, flags, win, wx0, wy0, g_scatter, acc, ww,
                                              kFixed ? T(fix_scale) : T(1), ovf);
      } else {
        gd[k] = geom_pixel<T, Cell, WW, WH>(bc, px, py, d_own[k], gI, gdd[k], ref_img, ref_depth, plane, H, W, flags, win, wx0,
                                      wy0, g_scatter, acc, kFixed ? T(fix_scale) : T(1), ovf);
      }
    }
    // A barrier waits for every outstanding global store / atomic of the wave, so everything that writes to
    // global memory comes after the last barrier: the round trips of the dense stores and of the window's
    // atomics then overlap with the next workgroup instead of stalling this one.
    STAMP(7);
    // (its barrier also orders the scatter's LDS atomics before the flush.  Raw sums against the pixel-frame point:
    // the pose reducer applies K^-1 once per image -- pose_partials_to_A is linear -- instead of wave 0 of every tile
    // doing 36 fp64 multiply-adds)
    block_sum_store<12>(acc, red + 3 * (kThreads / kWave), pa.gPp + 12 * ((size_t)(b * nby + blk.y) * nbx + blk.x));
#pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      const int ly = strip * STRIP + k, py = py0 + k;
      if (in_x && ly >= 1 && ly <= TH - 2 && py < H && !(flags & SCSFM_DEBUG_X4))
        st_at(g_dense, (unsigned(py) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), gd[k]);
    }
    if (!(flags & (SCSFM_DEBUG_X1 | SCSFM_DEBUG_X5)))
      flush_scatter_region<T, Cell, WW, WH>(win, wx0, wy0, cx0, cy0, cx1, cy1, g_scatter, W, ww, kFixed ? T(fix_inv) : T(1));
    STAMP(8);
  }
}


#undef SCSFM_ISSUE_STAGE_LOADS
}  // namespace scsfm
