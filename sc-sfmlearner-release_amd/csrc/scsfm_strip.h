// The speculative forward of a pair-direction as a REGISTER PIPELINE per wave (gfx950), replacing the LDS-tiled
// kernel of round 1 (photo_tile<kSpec = true>, kept for the backward's fallback passes).
//
// What it computes is unchanged: compute_pairwise_loss (loss_functions.py:95-119) forward -- three partial sums per
// work unit -- and the complete backward of the pair-direction up to the scalar the reduction supplies later:
// the dense dL/d tgt_depth plane, the scattered dL/d ref_depth plane and per-unit partials of dL/d(A|c).
//
// How (measured on MI355X, tools/ubench: a wave64 VALU instruction issues in ~1.05 ns per SIMD for
// add/mul/fma/mov/and/or, ~1.9 ns for min/max/cmp/cndmask/cvt/floor/DPP/packed/fp64, 3.5 ns for rcp;
// ds_add_f32 executes at 1.25 ns PER LANE per CU whereas ds_add_u32 takes 2 ns per wave-instruction):
//   * one wave owns a 64-pixel-wide column strip of an image segment and marches down it one row per step;
//     lane = column.  Nothing is exchanged through LDS and there is no barrier: the 3x3 SSIM windows and their
//     transpose take their horizontal neighbours from the adjacent lanes (DPP wave shifts fused into the adds) and
//     their vertical neighbours from the rows the lane itself processed one and two steps earlier (registers).
//     A strip yields 60 output columns (lanes 2..61: the transpose of the box filter at p needs the statistics at
//     p +- 1, which need the warped colours at p +- 2); vertically only the segment ends are redone (RS + 4 warped
//     rows for RS output rows), where the tiles of round 1 warped 66 x 18 pixels for 62 x 14 outputs.
//   * everything the geometry tail needs of a pixel (d value / d(ix, iy) of the three colours and of the sampled
//     depth, tap weights, X, Y, 1/Z) is kept from its warp two steps earlier: no second projection, no second
//     gather.
//   * bilinear taps: one 8-byte load per row of taps from a column clamped into [0, W-2], weights from the hat
//     function max(0, 1 - |ix - column|) -- exactly the bilinear weight of an in-image column, and 0 for a column
//     that is not a tap -- so the zeros padding costs no compare / select.
//   * the scatter into dL/d ref_depth is staged in a per-wave LDS window of 32-bit FIXED-POINT cells (ds_add_u32;
//     2^-20 per unit, +-2048: the float LDS atomic is 40x slower) and flushed once per unit with row-coalesced
//     global fp32 atomics; taps outside the window go to global atomics directly.  (fp64 instantiation: plain
//     fp64 cells, for the gradient-check tests.)
#pragma once
#include "scsfm_geom.h"
#include "scsfm_ssim.h"

namespace scsfm {

constexpr int kStripOut = kWave - 4;  // output columns per strip (lanes 2 .. 61)
#ifndef SCSFM_STRIP_ROWS  // tuning knob: output rows per work unit
#define SCSFM_STRIP_ROWS 32
#endif
constexpr int kStripRows = SCSFM_STRIP_ROWS;
constexpr int kStripWaves = kThreads / kWave;  // independent work units per workgroup
constexpr int kSWinW = 96;                     // scatter window of a unit (columns x rows)
constexpr int kSWinH = kStripRows + 16;
__host__ __device__ inline int strip_nbx(int W) { return ceil_div(W, kStripOut); }
__host__ __device__ inline int strip_nby(int H) { return ceil_div(H, kStripRows); }

template <typename T> struct StripCell { typedef typename WinCell<T>::type type; };  // fixed point / fp64 (scsfm_geom.h)

// ------------------------------------------------------------------------------------------------------
// lane <- neighbouring lane (DPP wave shift; lanes without a neighbour read 0).  The compiler folds the shift into
// the consuming add / fmac (v_add_f32_dpp ... wave_shr:1).
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

// ------------------------------------------------------------------------------------------------------
// What a lane keeps of one row: the pixel it warped there.
// ------------------------------------------------------------------------------------------------------
template <typename T>
struct StripRow {
  T x[3], y[3];      // target / warped colours
  T coef, m, l1;     // m (1 - diff_depth) [or m], mask, sum_c clamp(|x_c - y_c|)
  T dIx[3], dIy[3];  // d warped colour / d (X/Z, Y/Z): the sampler's d / d(ix, iy) times d ix / d (X/Z) = (W/2)(2/(W-1)),
                     // or 0 where the coordinate was overwritten / clipped
  T dDx, dDy;        // likewise for the sampled depth
  T gZc, gDpc;       // d diff_depth / d Z (times the Z >= 1e-3 gate), d diff_depth / d D_p (0 outside the clamp)
  T wxa, wxb, wya, wyb;  // bilinear weights of the tap pair's columns / of the two tap rows
  unsigned tap;      // (row << 16) | column of the first tap (clamped into the image)
  T Xz, Yz, iz, d;   // gate * X / Z^2, gate * Y / Z^2 (what d Z' takes from the x / y gradients), 1 / Z, target depth
};

template <typename T>
struct StripHSums {  // horizontal 3-sums of one row: x, y, x^2, y^2, x y per colour
  T sx[3], sy[3], sxx[3], syy[3], sxy[3];
};

// One work unit.  `win`: this wave's private scatter window (kSWinH x kSWinW cells).
template <typename T, unsigned kFlags>
struct StripUnit {
  typedef typename StripCell<T>::type Cell;
  // ---- wave-uniform / per-lane constants ----
  unsigned flags;
  bool with_ssim, with_mask, with_auto, border;
  int lane, H, W, r0, r1, px, u;
  unsigned plane;
  bool col_in, own_x, border_cols;
  const T* __restrict__ tgt_img; const T* __restrict__ ref_img; const T* __restrict__ tgt_depth; const T* __restrict__ ref_depth;
  T* __restrict__ g_dense; T* __restrict__ g_scatter;
  // q = (A K^-1)(u, v, 1) = qc + qv * v: X = q_x d + c_x, ... (the column part qc per lane, the rest wave-uniform)
  T qcx, qcy, qcz, qvx, qvy, qvz, c0, c1, c2;
  T Kinv[9];  // only for the unit's epilogue (dL/dA from the sums accumulated against d (u, v, 1))
  T uf, inv_w, inv_h, wl, wr, r_hint;
  Cell* __restrict__ win;
  int wx0, wy0;
  // ---- pipeline state ----
  StripRow<T> rows[3];
  StripHSums<T> hs[3];
  T ht[3][3][3];  // [slot][colour][map]: horizontally transposed 1/9 (g_mu_y, g_E[y^2], g_E[xy]) of a row
  T bsum_q[3];    // sum_c blend_c of the row a slot holds
  T acc_p, acc_g, acc_m, acc[12];
  T nd, nt[3], nr[3];  // streaming inputs of the next row to warp

  __device__ __forceinline__ void prefetch(int t) {
    const int v = reflect_index(t, H);
    const unsigned off = (unsigned(v) * unsigned(W) + unsigned(u)) * unsigned(sizeof(T));
    nd = ld_at(tgt_depth, off);
#pragma unroll
    for (int c = 0; c < 3; ++c) nt[c] = ld_at(tgt_img + c * plane, off);
#pragma unroll
    for (int c = 0; c < 3; ++c) nr[c] = with_auto ? ld_at(ref_img + c * plane, off) : T(0);
  }

  // ---- warp row t into `rw`; on the first call also place the scatter window --------------------------
  __device__ __forceinline__ void warp_row(StripRow<T>& rw, int t, bool place_window) {
    const int v = reflect_index(t, H);
    const bool inimg = col_in && t >= 0 && t < H;
    const T d = nd;
    T ct[3], cr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { ct[c] = nt[c]; cr[c] = nr[c]; }
    const T vf = T(v);
    const T X = (qcx + qvx * vf) * d + c0;         // A (K^-1 (u, v, 1) d) + c  (inverse_warp.py:253-260)
    const T Y = (qcy + qvy * vf) * d + c1;
    const T Zraw = (qcz + qvz * vf) * d + c2;
    const T Z = t_max(Zraw, T(kZMin));            // inverse_warp.py:211
    const T iz = t_rcp(Z);
    const T xn = (X * iz) * inv_w - T(1);         // inverse_warp.py:217-218
    const T yn = (Y * iz) * inv_h - T(1);
    T ix = ((xn + T(1)) * T(W) - T(1)) * T(0.5);  // grid_sampler_unnormalize, align_corners = False
    T iy = ((yn + T(1)) * T(H) - T(1)) * T(0.5);
    T kx = T(0.5) * T(W) * inv_w, ky = T(0.5) * T(H) * inv_h;
    const bool vx = t_abs(xn) <= T(1), vy = t_abs(yn) <= T(1);  // (false for NaN)
    const bool valid = vx && vy;                                  // inverse_warp.py:264
    if (!border) {
      // inverse_warp.py:219-224: an out-of-range coordinate becomes the constant 2, i.e. a sampling position all of
      // whose taps lie outside the image; -1 is such a position too (weight 1 on column -1, weight 0 on column 0)
      if (!vx) { ix = T(-1); kx = T(0); }
      if (!vy) { iy = T(-1); ky = T(0); }
    } else {  // clip_coordinates_set_grad: zero gradient on and outside the bounds
      if (!(ix > T(0))) { ix = T(0); kx = T(0); } else if (!(ix < T(W - 1))) { ix = T(W - 1); kx = T(0); }
      if (!(iy > T(0))) { iy = T(0); ky = T(0); } else if (!(iy < T(H - 1))) { iy = T(H - 1); ky = T(0); }
    }
    // tap block (xa, xa + 1) x (ya, ya + 1), always inside the image; hat-function weights and their slopes
    // (scsfm_geom.h: hat_axis), the slopes times d ix / d (X/Z)
    int xa, ya;
    T wxa, wxb, wya, wyb, dxa, dxb, dya, dyb;
    hat_axis(ix, W, xa, wxa, wxb, dxa, dxb);
    hat_axis(iy, H, ya, wya, wyb, dya, dyb);
    dxa *= kx; dxb *= kx; dya *= ky; dyb *= ky;
    const unsigned off = (unsigned(ya) * unsigned(W) + unsigned(xa)) * unsigned(sizeof(T));
    const unsigned off_s = off + unsigned(W) * unsigned(sizeof(T));
    TapRows<T> tc[3], td;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tc[c].n = ld_at(reinterpret_cast<const TapPair<T>*>(ref_img + c * plane), off);
      tc[c].s = ld_at(reinterpret_cast<const TapPair<T>*>(ref_img + c * plane), off_s);
    }
    td.n = ld_at(reinterpret_cast<const TapPair<T>*>(ref_depth), off);
    td.s = ld_at(reinterpret_cast<const TapPair<T>*>(ref_depth), off_s);
    prefetch(t + 1);  // the next row's streaming loads go out behind this row's gathers
    if (place_window) {
      // where this unit's pixels land: the smallest tap column / row among the valid lanes of its first row
      int mx = valid ? xa : (1 << 30), my = valid ? ya : (1 << 30);
#pragma unroll
      for (int o = kWave / 2; o > 0; o >>= 1) {
        const int ox = __shfl_xor(mx, o), oy = __shfl_xor(my, o);
        mx = ox < mx ? ox : mx; my = oy < my ? oy : my;
      }
      const bool any = mx != (1 << 30);
      // (first warped row = r0 - 2: the outputs start two rows further down)
      wx0 = (any ? mx : px - lane) - (kSWinW - kWave) / 2 + 1;
      wy0 = (any ? my : r0 - 2) - 5;
    }
    T l1 = T(0), ident = T(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T rn = wxa * tc[c].n.a + wxb * tc[c].n.b, rs = wxa * tc[c].s.a + wxb * tc[c].s.b;  // tap rows at ix
      const T qn = dxa * tc[c].n.a + dxb * tc[c].n.b, qs = dxa * tc[c].s.a + dxb * tc[c].s.b;  // their d / d ix
      rw.x[c] = ct[c];
      rw.y[c] = wya * rn + wyb * rs;
      rw.dIx[c] = wya * qn + wyb * qs;
      rw.dIy[c] = dya * rn + dyb * rs;
      l1 += clamp01(t_abs(rw.x[c] - rw.y[c]));  // loss_functions.py:99
      ident += t_abs(rw.x[c] - cr[c]);
    }
    const T rn = wxa * td.n.a + wxb * td.n.b, rs = wxa * td.s.a + wxb * td.s.b;
    const T qn = dxa * td.n.a + dxb * td.n.b, qs = dxa * td.s.a + dxb * td.s.b;
    const T Dp = wya * rn + wyb * rs;
    rw.dDx = wya * qn + wyb * qs;
    rw.dDy = dya * rn + dyb * rs;
    const T diff = Z - Dp, isum = t_rcp(Z + Dp);
    const T raw = t_abs(diff) * isum;
    const T dd = clamp01(raw);                     // loss_functions.py:101
    const bool ddpass = raw >= T(0) && raw <= T(1);
    const T g2 = ddpass ? t_sgn(diff) * T(2) * isum * isum : T(0);
    const T zg = Zraw >= T(kZMin) ? T(1) : T(0);   // Z = clamp(Zraw, min = 1e-3) passes gradient where Zraw >= 1e-3
    rw.gZc = g2 * Dp * zg;
    rw.gDpc = -g2 * Z;
    T m = (valid && inimg) ? T(1) : T(0);
    if (with_auto) m = (l1 < ident) ? m : T(0);    // loss_functions.py:103-105 (both means share the divisor 3)
    rw.m = m;
    rw.l1 = l1;
    rw.coef = with_mask ? m * (T(1) - dd) : m;     // loss_functions.py:111-113
    rw.wxa = wxa; rw.wxb = wxb; rw.wya = wya; rw.wyb = wyb;
    rw.tap = (unsigned(ya) << 16) | unsigned(xa);
    const T zq = zg * iz * iz;
    rw.Xz = X * zq; rw.Yz = Y * zq; rw.iz = iz; rw.d = d;
    if (own_x && t >= r0 && t < r1) { acc_g += dd * m; acc_m += m; }
  }

  __device__ __forceinline__ void hsums(const StripRow<T>& rw, StripHSums<T>& h) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T x = rw.x[c], y = rw.y[c];
      h.sx[c] = box3(x); h.sy[c] = box3(y);
      h.sxx[c] = box3(x * x); h.syy[c] = box3(y * y); h.sxy[c] = box3(x * y);
    }
  }

  // statistics of row q (slot `rq`; the rows above / below it in `ha`, `hb`), its blend sum, the forward photo sum,
  // and the horizontally transposed gradient maps of the row
  __device__ __forceinline__ void stats_row(int q, const StripRow<T>& rq, const StripHSums<T>& ha, const StripHSums<T>& hq,
                                            const StripHSums<T>& hb, T (&hto)[3][3], T& bsum) {
    T bs = with_ssim ? T(0.15) * rq.l1 : rq.l1;  // loss_functions.py:109
    if (with_ssim) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        WinSums<T> w;
        w.s1 = make2((ha.sx[c] + hq.sx[c]) + hb.sx[c], (ha.sy[c] + hq.sy[c]) + hb.sy[c]);
        w.s2 = make2((ha.sxx[c] + hq.sxx[c]) + hb.sxx[c], (ha.syy[c] + hq.syy[c]) + hb.syy[c]);
        w.sxy = (ha.sxy[c] + hq.sxy[c]) + hb.sxy[c];
        const SsimStats<T> st = ssim_stats(w);
        bs += T(0.85) * clamp01(st.raw);
        // s = clamp((1 - S)/2, 0, 1): d s / d S = -1/2 inside the clamp (inclusive bounds)
        const T gS = (st.raw >= T(0) && st.raw <= T(1)) ? rq.coef * T(0.85) * T(-0.5) : T(0);
        T g[3];
        ssim_grad_y(st, gS, g[0], g[1], g[2]);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          T h = box3(g[m]);
          if (border_cols) h += (wl - T(1)) * lane_left(g[m]) + (wr - T(1)) * lane_right(g[m]);
          hto[c][m] = h;
        }
      }
    }
    bsum = bs;
    if (own_x && q >= r0 && q < r1) acc_p += bs * rq.coef;
  }

  __device__ __forceinline__ void scatter(const StripRow<T>& rp, T gDp) {
    if (gDp == T(0)) return;
    const int xa = int(rp.tap & 0xffffu), ya = int(rp.tap >> 16);
    const int lx = xa - wx0, ly = ya - wy0;
    const T v0 = gDp * (rp.wya * rp.wxa), v1 = gDp * (rp.wya * rp.wxb), v2 = gDp * (rp.wyb * rp.wxa), v3 = gDp * (rp.wyb * rp.wxb);
    if (unsigned(lx) < unsigned(kSWinW - 1) && unsigned(ly) < unsigned(kSWinH - 1)) {
      Cell* c = win + ly * kSWinW + lx;
      win_add(c, v0); win_add(c + 1, v1); win_add(c + kSWinW, v2); win_add(c + kSWinW + 1, v3);
    } else {
      T* g = g_scatter + unsigned(ya) * unsigned(W) + unsigned(xa);
      if (v0 != T(0)) atomicAdd(g, v0);
      if (v1 != T(0)) atomicAdd(g + 1, v1);
      if (v2 != T(0)) atomicAdd(g + W, v2);
      if (v3 != T(0)) atomicAdd(g + W + 1, v3);
    }
  }

  // output row p (slot `rp`): finish dL/d warped colours from the transposed maps of rows p-1, p, p+1, then the
  // geometry tail
  __device__ __forceinline__ void output_row(int p, const StripRow<T>& rp, const T (&h0)[3][3], const T (&h1)[3][3],
                                             const T (&h2)[3][3], T bsum) {
    const T wt = reflect_mult<T>(-1, p, H), wb = reflect_mult<T>(1, p, H);  // wave-uniform
    T gI[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T x = rp.x[c], y = rp.y[c], d = x - y;
      // d clamp(|d|, 0, 1) / d Iw: the clamp passes gradient on [0, 1] inclusive, abs uses sgn
      const T l1g = (t_abs(d) <= T(1)) ? -t_sgn(d) : T(0);
      if (with_ssim) {
        const T g0 = wt * h0[c][0] + h1[c][0] + wb * h2[c][0];
        const T g1 = wt * h0[c][1] + h1[c][1] + wb * h2[c][1];
        const T g2 = wt * h0[c][2] + h1[c][2] + wb * h2[c][2];
        gI[c] = g0 + T(2) * y * g1 + x * g2 + rp.coef * T(0.15) * l1g;
      } else {
        gI[c] = rp.coef * l1g;
      }
    }
    if (!own_x) return;
    // dL/d diff_depth: directly (geometry loss) and through the weight mask (no detach, loss_functions.py:111-113)
    const T gdd = r_hint * rp.m - (with_mask ? rp.m * bsum : T(0));
    const T gDp = gdd * rp.gDpc, gZ = gdd * rp.gZc;
    const T gix = gI[0] * rp.dIx[0] + gI[1] * rp.dIx[1] + gI[2] * rp.dIx[2] + gDp * rp.dDx;
    const T giy = gI[0] * rp.dIy[0] + gI[1] * rp.dIy[1] + gI[2] * rp.dIy[2] + gDp * rp.dDy;
    scatter(rp, gDp);
    // gix, giy are already gradients with respect to X/Z, Y/Z (the factors of ix = ((xn+1) W - 1)/2,
    // xn = 2 (X/Z)/(W-1) - 1 and the overwrite / clip gates were folded into the stored derivatives)
    const T dX = gix * rp.iz, dY = giy * rp.iz;
    const T dZ = gZ - (gix * rp.Xz + giy * rp.Yz);
    // dL/d(A K^-1) accumulated against d (u, v, 1) (turned into dL/dA in the unit's epilogue); dL/dc
    const T vf = T(p);
    const T tX = dX * rp.d, tY = dY * rp.d, tZ = dZ * rp.d;
    acc[0] += tX * uf; acc[1] += tX * vf; acc[2] += tX;
    acc[3] += tY * uf; acc[4] += tY * vf; acc[5] += tY;
    acc[6] += tZ * uf; acc[7] += tZ * vf; acc[8] += tZ;
    acc[9] += dX; acc[10] += dY; acc[11] += dZ;
    // dL/d depth = <d(X, Y, Z')/d depth, (dX, dY, dZ)> = <q, .>
    const T gd = (qcx + qvx * vf) * dX + (qcy + qvy * vf) * dY + (qcz + qvz * vf) * dZ;
    st_at(g_dense, (unsigned(p) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), gd);
  }

  // step k: warp row t = r0 - 2 + k (slot S), statistics of row t - 1, output of row t - 2
  template <int S>
  __device__ __forceinline__ void step(int k) {
    constexpr int S1 = (S + 2) % 3, S2 = (S + 1) % 3;  // slots of rows t - 1, t - 2 (and of row t - 3's maps: S)
    const int t = r0 - 2 + k;
    warp_row(rows[S], t, k == 0);
    if (with_ssim) hsums(rows[S], hs[S]);
    if (k >= 2) stats_row(t - 1, rows[S1], hs[S2], hs[S1], hs[S], ht[S1], bsum_q[S1]);
    if (k >= 4) output_row(t - 2, rows[S2], ht[S], ht[S2], ht[S1], bsum_q[S2]);
  }

  __device__ __forceinline__ void flush() {
    // every lane's last scatter precedes any lane's read of the window: a wave runs in lockstep and its LDS
    // operations complete in order, so on the GPU this is a scheduling fence only (the CPU simulation of the
    // tests, whose lanes are fibres, synchronises here)
    __builtin_amdgcn_wave_barrier();
    for (int ly = 0; ly < kSWinH; ++ly) {
#pragma unroll
      for (int j = 0; j < (kSWinW + kWave - 1) / kWave; ++j) {
        const int lx = lane + j * kWave;
        if (lx < kSWinW) {
          const Cell v = win[ly * kSWinW + lx];
          // only cells that received an in-image tap are non-zero, so every flushed cell is a valid pixel
          if (v != Cell(0)) {
            atomicAdd(g_scatter + unsigned(wy0 + ly) * unsigned(W) + unsigned(wx0 + lx), T(win_value(v)));
          }
        }
      }
    }
  }
};

template <typename T, unsigned kFlags>
__device__ __forceinline__ void strip_unit(const PairArgs<T>& pa, int b, int seg, int strip, int nbx, int nby, int B, int H,
                                           int W, unsigned flags_arg, T r_hint, typename StripCell<T>::type* __restrict__ win) {
  StripUnit<T, kFlags> s;
  s.flags = kFlags == kRuntimeFlags ? flags_arg : kFlags;
  s.with_ssim = (s.flags & SCSFM_WITH_SSIM) != 0; s.with_mask = (s.flags & SCSFM_WITH_MASK) != 0;
  s.with_auto = (s.flags & SCSFM_WITH_AUTO_MASK) != 0; s.border = (s.flags & SCSFM_PAD_BORDER) != 0;
  s.lane = threadIdx.x & (kWave - 1);
  s.H = H; s.W = W;
  s.plane = unsigned(H) * unsigned(W);
  const size_t gplane = (size_t)B * s.plane;
  s.tgt_img = pa.tgt_img + (size_t)b * 3 * s.plane;
  s.ref_img = pa.ref_img + (size_t)b * 3 * s.plane;
  s.tgt_depth = pa.tgt_depth + (size_t)b * s.plane;
  s.ref_depth = pa.ref_depth + (size_t)b * s.plane;
  s.g_dense = pa.gbuf + kPlaneDense * gplane + (size_t)b * s.plane;
  s.g_scatter = pa.gbuf + kPlaneScatter * gplane + (size_t)b * s.plane;
  s.r0 = seg * kStripRows;
  s.r1 = s.r0 + kStripRows < H ? s.r0 + kStripRows : H;
  s.px = strip * kStripOut - 2 + s.lane;
  s.u = reflect_index(s.px, W);
  s.col_in = s.px >= 0 && s.px < W;
  s.own_x = s.lane >= 2 && s.lane <= kWave - 3 && s.px < W;
  s.uf = T(s.u);
  {
    // M = A K^-1 (wave-uniform; evaluated once per unit): X = (M (u, v, 1)) d + c
    const BatchConsts<T>& bc = pa.consts[b];
    T M[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) M[3 * i + j] = bc.A[3 * i] * bc.Kinv[j] + bc.A[3 * i + 1] * bc.Kinv[3 + j] + bc.A[3 * i + 2] * bc.Kinv[6 + j];
    s.qcx = M[0] * s.uf + M[2]; s.qcy = M[3] * s.uf + M[5]; s.qcz = M[6] * s.uf + M[8];
    s.qvx = M[1]; s.qvy = M[4]; s.qvz = M[7];
    s.c0 = bc.c[0]; s.c1 = bc.c[1]; s.c2 = bc.c[2];
#pragma unroll
    for (int i = 0; i < 9; ++i) s.Kinv[i] = bc.Kinv[i];
  }
  s.inv_w = T(2) / T(W - 1); s.inv_h = T(2) / T(H - 1);
  // transpose of (reflect pad o box): an output next to the image border is reached twice from the border pixel
  s.wl = reflect_mult<T>(-1, s.px, W); s.wr = reflect_mult<T>(1, s.px, W);
  s.border_cols = strip == 0 || (strip + 1) * kStripOut + 2 >= W;  // wave-uniform: wl / wr may differ from 1 here
  s.r_hint = r_hint;
  s.win = win;
  s.wx0 = 0; s.wy0 = 0;
  for (int i = s.lane; i < kSWinW * kSWinH; i += kWave) win[i] = typename StripCell<T>::type(0);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    s.bsum_q[q] = T(0);
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
      for (int m = 0; m < 3; ++m) s.ht[q][c][m] = T(0);
  }
  s.acc_p = s.acc_g = s.acc_m = T(0);
#pragma unroll
  for (int i = 0; i < 12; ++i) s.acc[i] = T(0);

  s.prefetch(s.r0 - 2);
  const int K = (s.r1 - s.r0) + 4;
  for (int k = 0; k < K; k += 3) {
    s.template step<0>(k);
    if (k + 1 < K) s.template step<1>(k + 1);
    if (k + 2 < K) s.template step<2>(k + 2);
  }
  s.flush();
  // the unit's three forward sums and its partials of dL/d(A|c)
  const size_t unit = (size_t)(b * nby + seg) * nbx + strip;
  {
    T v[3] = {s.acc_p, s.acc_g, s.acc_m};
    bool lead;
    const int idx = wave_sum_packed<3>(v, lead);
    if (lead) pa.partials[3 * unit + idx] = double(v[0]);
  }
  {
    // dL/dA[i][j] = sum_k G[i][k] K^-1[j][k]  (cam = K^-1 (u, v, 1) d, G accumulated against d (u, v, 1))
    T gA[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        gA[3 * i + j] = s.acc[3 * i] * s.Kinv[3 * j] + s.acc[3 * i + 1] * s.Kinv[3 * j + 1] + s.acc[3 * i + 2] * s.Kinv[3 * j + 2];
    gA[9] = s.acc[9]; gA[10] = s.acc[10]; gA[11] = s.acc[11];
    bool lead;
    const int idx = wave_sum_packed<12>(gA, lead);
    if (lead) pa.gPp[12 * unit + idx] = double(gA[0]);
  }
}

// The speculative forward: kStripWaves independent work units per workgroup, XCD-aware order.
template <typename T, unsigned kFlags = kRuntimeFlags>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? 2 : 1) void pair_strip_kernel(PairBatch<T> pb, int B, int H, int W,
                                                                                      int nbx, int nby, int nunits,
                                                                                      unsigned flags, T r_hint) {
  typedef typename StripCell<T>::type Cell;
  __shared__ Cell win[kStripWaves][kSWinH * kSWinW];
  const int nblocks = (int)gridDim.x;
  // consecutive workgroups go to different XCDs: give every XCD a contiguous eighth of the logical order, so that the
  // strips that share halo columns and gather from the same neighbourhood meet in one L2
  const BlockId blk = xcd_tile_of((int)blockIdx.x, nblocks, 1, 1);
  // (readfirstlane: the wave index is uniform by construction, but only this tells the compiler -- everything derived
  // from it, the unit's pointers and its 24 per-batch constants included, then lives in scalar registers)
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x / kWave));
  const int unit = blk.x * kStripWaves + wave;
  if (unit >= nunits) return;
  const int strip = unit % nbx;
  const int rest = unit / nbx;
  const int seg = rest % nby, z = rest / nby;
  const int pair = z / B, b = z - pair * B;
  strip_unit<T, kFlags>(pb.p[pair], b, seg, strip, nbx, nby, B, H, W, flags, r_hint, win[wave]);
}

}  // namespace scsfm
