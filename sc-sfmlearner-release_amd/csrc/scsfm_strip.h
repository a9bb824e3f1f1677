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
constexpr int kStripRowsMin = 16, kStripRowsMax = 48;  // output rows per work unit: chosen per launch (strip_rows)
constexpr int kStripWaves = kThreads / kWave;          // independent work units per workgroup
constexpr int kSWinW = 80;                             // scatter window of a unit (columns x rows): 17 KB per wave
constexpr int kSWinH = kStripRowsMax + 6;

__host__ __device__ inline int strip_nbx(int W) { return ceil_div(W, kStripOut); }

// Output rows per work unit.  Every unit costs about (rows + 4 + a flush) row-steps and all of them take the same
// time, so a launch is as long as ceil(units / resident waves) units: pick the row count that minimises that
// (at BASELINE.json configs[1]: 43 rows -> 4032 units = 1.97 x the 2048 resident waves; with 32 rows the 5376 units
// ran as three rounds of which the last was 62 % full).  `slots` = waves the device holds (CUs x 4 SIMDs x 2).
inline int strip_rows(int H, int units_per_row_block, int slots) {
  int best = kStripRowsMin;
  double best_cost = 1e300;
  for (int rs = kStripRowsMin; rs <= kStripRowsMax; ++rs) {
    const long units = (long)units_per_row_block * ceil_div(H, rs);
    const long rounds = (units + slots - 1) / slots;
    const int last = H - (ceil_div(H, rs) - 1) * rs;  // the last segment may be short; the round lasts as long as a full one
    (void)last;
    const double cost = double(rounds) * (rs + 4 + 3);
    if (cost < best_cost - 1e-9) { best_cost = cost; best = rs; }
  }
  return best;
}

template <typename T> struct StripCell { typedef typename WinCell<T>::type type; };  // fixed point / fp64 (scsfm_geom.h)

// ------------------------------------------------------------------------------------------------------
// What a lane keeps of one row until the row's gradients are finished two steps later: the colours, the mask terms
// and where the pixel landed.  Everything else the geometry tail needs (tap values, weights, their slopes, X, Y,
// 1/Z) is re-derived there from `d`, `ix`, `iy` and a second gather that is issued at the top of the step and
// consumed at its end: 13 registers per row instead of 28, which is what keeps the kernel free of scratch spills --
// a spilled value's reload waits for every older load of the wave (the prefetched next row included), i.e. it turns
// the prefetch into a synchronous round trip per row (measured: 46 % of a wave's time in the tail).
// ------------------------------------------------------------------------------------------------------
template <typename T>
struct StripRow {
  T x[3], y[3];   // target / warped colours
  T coef, m, l1;  // m (1 - diff_depth) [or m], mask, sum_c clamp(|x_c - y_c|)
  T ix, iy;       // sampling position (after the zeros-mode overwrite / border clip)
  T d;            // target depth
  unsigned tap;   // (row << 16) | column of the 2 x 2 tap block
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
  T uf, inv_w, inv_h, r_hint;
  Cell* __restrict__ win;
  int wx0, wy0;
  // ---- pipeline state ----
  StripRow<T> rows[3];
  T hs[3][15];    // horizontal 3-sums of a row: (x, y, x^2, y^2, x y) per colour
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

  // where the pixel (column of this lane, row v) of depth d lands; returns validity, leaves (X, Y, Zraw, Z, 1/Z) and
  // the sampling position with its gradient factors d ix / d (X/Z) (0 where overwritten / clipped)
  __device__ __forceinline__ bool project(T vf, T d, T& X, T& Y, T& Zraw, T& Z, T& iz, T& ix, T& iy, T& kx, T& ky) {
    X = (qcx + qvx * vf) * d + c0;                 // A (K^-1 (u, v, 1) d) + c  (inverse_warp.py:253-260)
    Y = (qcy + qvy * vf) * d + c1;
    Zraw = (qcz + qvz * vf) * d + c2;
    Z = t_max(Zraw, T(kZMin));                     // inverse_warp.py:211
    iz = t_rcp(Z);
    const T xn = (X * iz) * inv_w - T(1);          // inverse_warp.py:217-218
    const T yn = (Y * iz) * inv_h - T(1);
    ix = ((xn + T(1)) * T(W) - T(1)) * T(0.5);     // grid_sampler_unnormalize, align_corners = False
    iy = ((yn + T(1)) * T(H) - T(1)) * T(0.5);
    kx = T(0.5) * T(W) * inv_w; ky = T(0.5) * T(H) * inv_h;
    const bool vx = t_abs(xn) <= T(1), vy = t_abs(yn) <= T(1);  // (false for NaN)
    if (!border) {
      // inverse_warp.py:219-224: an out-of-range coordinate becomes the constant 2, i.e. a sampling position all of
      // whose taps lie outside the image; -1 is such a position too (weight 1 on column -1, weight 0 on column 0)
      if (!vx) { ix = T(-1); kx = T(0); }
      if (!vy) { iy = T(-1); ky = T(0); }
    } else {  // clip_coordinates_set_grad: zero gradient on and outside the bounds
      if (!(ix > T(0))) { ix = T(0); kx = T(0); } else if (!(ix < T(W - 1))) { ix = T(W - 1); kx = T(0); }
      if (!(iy > T(0))) { iy = T(0); ky = T(0); } else if (!(iy < T(H - 1))) { iy = T(H - 1); ky = T(0); }
    }
    return vx && vy;                               // inverse_warp.py:264
  }

  // ---- warp row t into `rw`; on the first call also place the scatter window --------------------------
  __device__ __forceinline__ void warp_row(StripRow<T>& rw, int t, bool place_window) {
    const int v = reflect_index(t, H);
    const bool inimg = col_in && t >= 0 && t < H;
    const T d = nd;
    T ct[3], cr[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { ct[c] = nt[c]; cr[c] = nr[c]; }
    T X, Y, Zraw, Z, iz, ix, iy, kx, ky;
    const bool valid = project(T(v), d, X, Y, Zraw, Z, iz, ix, iy, kx, ky);
    // tap block (xa, xa + 1) x (ya, ya + 1), always inside the image; hat-function weights (scsfm_geom.h: hat_axis)
    int xa, ya;
    T wxa, wxb, wya, wyb, sa, sb;
    hat_axis(ix, W, xa, wxa, wxb, sa, sb);
    hat_axis(iy, H, ya, wya, wyb, sa, sb);
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
    const T w00 = wya * wxa, w01 = wya * wxb, w10 = wyb * wxa, w11 = wyb * wxb;
    T l1 = T(0), ident = T(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      rw.x[c] = ct[c];
      rw.y[c] = tc[c].n.a * w00 + tc[c].n.b * w01 + tc[c].s.a * w10 + tc[c].s.b * w11;
      l1 += clamp01(t_abs(rw.x[c] - rw.y[c]));  // loss_functions.py:99
      ident += t_abs(rw.x[c] - cr[c]);
    }
    const T Dp = td.n.a * w00 + td.n.b * w01 + td.s.a * w10 + td.s.b * w11;
    const T dd = clamp01(t_abs(Z - Dp) * t_rcp(Z + Dp));  // loss_functions.py:101
    T m = (valid && inimg) ? T(1) : T(0);
    if (with_auto) m = (l1 < ident) ? m : T(0);    // loss_functions.py:103-105 (both means share the divisor 3)
    rw.m = m;
    rw.l1 = l1;
    rw.coef = with_mask ? m * (T(1) - dd) : m;     // loss_functions.py:111-113
    rw.ix = ix; rw.iy = iy; rw.d = d;
    rw.tap = (unsigned(ya) << 16) | unsigned(xa);
    if (own_x && t >= r0 && t < r1) { acc_g += dd * m; acc_m += m; }
  }

  __device__ __forceinline__ void hsums(const StripRow<T>& rw, T (&h)[15]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T x = rw.x[c], y = rw.y[c];
      h[5 * c] = box3(x); h[5 * c + 1] = box3(y);
      h[5 * c + 2] = box3(x * x); h[5 * c + 3] = box3(y * y); h[5 * c + 4] = box3(x * y);
    }
  }

  // statistics of row q (slot `rq`; the rows above / below it in `ha`, `hb`), its blend sum, the forward photo sum,
  // and the horizontally transposed gradient maps of the row
  __device__ __forceinline__ void stats_row(int q, const StripRow<T>& rq, const T (&ha)[15], const T (&hq)[15],
                                            const T (&hb)[15], T (&hto)[3][3], T& bsum) {
    T bs = with_ssim ? T(0.15) * rq.l1 : rq.l1;  // loss_functions.py:109
    if (with_ssim) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        WinSums<T> w;
        w.s1 = make2((ha[5 * c] + hq[5 * c]) + hb[5 * c], (ha[5 * c + 1] + hq[5 * c + 1]) + hb[5 * c + 1]);
        w.s2 = make2((ha[5 * c + 2] + hq[5 * c + 2]) + hb[5 * c + 2], (ha[5 * c + 3] + hq[5 * c + 3]) + hb[5 * c + 3]);
        w.sxy = (ha[5 * c + 4] + hq[5 * c + 4]) + hb[5 * c + 4];
        const SsimStats<T> st = ssim_stats(w);
        bs += T(0.85) * clamp01(st.raw);
        // s = clamp((1 - S)/2, 0, 1): d s / d S = -1/2 inside the clamp (inclusive bounds)
        const T gS = clamp01(st.raw) == st.raw ? rq.coef * T(0.85) * T(-0.5) : T(0);  // (i.e. 0 <= raw <= 1)
        T g[3];
        ssim_grad_y(st, gS, g[0], g[1], g[2]);
#pragma unroll
        for (int m = 0; m < 3; ++m) {
          T h = box3(g[m]);
          if (border_cols) {
            // transpose of (reflect pad o box): an output next to the image border is reached twice from the
            // border pixel
            h += (reflect_mult<T>(-1, px, W) - T(1)) * lane_left(g[m]) + (reflect_mult<T>(1, px, W) - T(1)) * lane_right(g[m]);
          }
          hto[c][m] = h;
        }
      }
    }
    bsum = bs;
    if (own_x && q >= r0 && q < r1) acc_p += bs * rq.coef;
  }

  // output row p (slot `rp`): finish dL/d warped colours from the transposed maps of rows p-1, p, p+1, then the
  // geometry tail on the taps `tc`, `td` gathered for it at the top of the step
  __device__ __forceinline__ void output_row(int p, const StripRow<T>& rp, const T (&h0)[3][3], const T (&h1)[3][3],
                                             const T (&h2)[3][3], T bsum, const TapRows<T> (&tc)[3], const TapRows<T>& td) {
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
    // the pixel's projection again (from its depth) and the weights / slopes of its tap block (from ix, iy)
    const T vf = T(p);
    const T qx = qcx + qvx * vf, qy = qcy + qvy * vf, qz = qcz + qvz * vf;
    const T X = qx * rp.d + c0, Y = qy * rp.d + c1, Zraw = qz * rp.d + c2;
    const T Z = t_max(Zraw, T(kZMin)), iz = t_rcp(Z);
    T kx = T(0.5) * T(W) * inv_w, ky = T(0.5) * T(H) * inv_h;
    if (!border) {  // overwritten coordinates are exactly -1 (a valid one is >= -1/2)
      kx = rp.ix == T(-1) ? T(0) : kx; ky = rp.iy == T(-1) ? T(0) : ky;
    } else {        // clipped coordinates sit exactly on the bounds (an unclipped one lies strictly inside)
      kx = (rp.ix == T(0) || rp.ix == T(W - 1)) ? T(0) : kx; ky = (rp.iy == T(0) || rp.iy == T(H - 1)) ? T(0) : ky;
    }
    int xa, ya;
    T wxa, wxb, wya, wyb, sxa, sxb, sya, syb;
    hat_axis(rp.ix, W, xa, wxa, wxb, sxa, sxb);
    hat_axis(rp.iy, H, ya, wya, wyb, sya, syb);
    sxa *= kx; sxb *= kx; sya *= ky; syb *= ky;  // slopes with respect to X/Z, Y/Z
    // dL/d diff_depth: directly (geometry loss) and through the weight mask (no detach, loss_functions.py:111-113)
    const T gdd = r_hint * rp.m - (with_mask ? rp.m * bsum : T(0));
    const T w00 = wya * wxa, w01 = wya * wxb, w10 = wyb * wxa, w11 = wyb * wxb;
    const T Dp = td.n.a * w00 + td.n.b * w01 + td.s.a * w10 + td.s.b * w11;
    const T diff = Z - Dp, isum = t_rcp(Z + Dp), raw = t_abs(diff) * isum;
    // diff_depth = clamp(|Z - Dp| / (Z + Dp), 0, 1), loss_functions.py:101
    const T g2 = raw <= T(1) ? gdd * t_sgn(diff) * T(2) * isum * isum : T(0);  // (raw >= 0 always: Z + Dp > 0)
    const T gZ = g2 * Dp, gDp = -g2 * Z;
    // d (sum over the planes of g_plane * sampled value) / d (X/Z, Y/Z): contract over the four planes first
    // (g = dL/d warped colour c, dL/dD_p), then apply the weights / slopes once
    const T na = gI[0] * tc[0].n.a + gI[1] * tc[1].n.a + gI[2] * tc[2].n.a + gDp * td.n.a;
    const T nb = gI[0] * tc[0].n.b + gI[1] * tc[1].n.b + gI[2] * tc[2].n.b + gDp * td.n.b;
    const T sa = gI[0] * tc[0].s.a + gI[1] * tc[1].s.a + gI[2] * tc[2].s.a + gDp * td.s.a;
    const T sb = gI[0] * tc[0].s.b + gI[1] * tc[1].s.b + gI[2] * tc[2].s.b + gDp * td.s.b;
    const T gix = wya * (sxa * na + sxb * nb) + wyb * (sxa * sa + sxb * sb);
    const T giy = sya * (wxa * na + wxb * nb) + syb * (wxa * sa + wxb * sb);
    // scatter dL/dD_p over the tap block
    if (gDp != T(0)) {
      const int lx = xa - wx0, ly = ya - wy0;
      const T v0 = gDp * w00, v1 = gDp * w01, v2 = gDp * w10, v3 = gDp * w11;
      if (unsigned(lx) < unsigned(kSWinW - 1) && unsigned(ly) < unsigned(kSWinH - 1) && win_fits(win, gDp)) {
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
    // gix, giy are gradients with respect to X/Z, Y/Z; Z = clamp(Zraw, min = 1e-3) passes gradient where Zraw >= 1e-3
    const T dX = gix * iz, dY = giy * iz;
    const T dZ = Zraw >= T(kZMin) ? gZ - (gix * X + giy * Y) * iz * iz : T(0);
    // dL/d(A K^-1) accumulated against d (u, v, 1) (turned into dL/dA in the unit's epilogue); dL/dc
    const T tX = dX * rp.d, tY = dY * rp.d, tZ = dZ * rp.d;
    acc[0] += tX * uf; acc[1] += tX * vf; acc[2] += tX;
    acc[3] += tY * uf; acc[4] += tY * vf; acc[5] += tY;
    acc[6] += tZ * uf; acc[7] += tZ * vf; acc[8] += tZ;
    acc[9] += dX; acc[10] += dY; acc[11] += dZ;
    // dL/d depth = <d(X, Y, Z')/d depth, (dX, dY, dZ)> = <q, .>
    st_at(g_dense, (unsigned(p) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), qx * dX + qy * dY + qz * dZ);
  }

  // step k: warp row t = r0 - 2 + k (slot S), statistics of row t - 1, output of row t - 2
  template <int S>
  __device__ __forceinline__ void step(int k) {
    constexpr int S1 = (S + 2) % 3, S2 = (S + 1) % 3;  // slots of rows t - 1, t - 2 (and of row t - 3's maps: S)
    const int t = r0 - 2 + k;
    // the tail's gather for row t - 2 goes out first: it is consumed at the very end of the step
    TapRows<T> tc[3], td;
    {
      const unsigned tap = rows[S2].tap;
      const unsigned off = ((tap >> 16) * unsigned(W) + (tap & 0xffffu)) * unsigned(sizeof(T));
      const unsigned off_s = off + unsigned(W) * unsigned(sizeof(T));
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        tc[c].n = ld_at(reinterpret_cast<const TapPair<T>*>(ref_img + c * plane), off);
        tc[c].s = ld_at(reinterpret_cast<const TapPair<T>*>(ref_img + c * plane), off_s);
      }
      td.n = ld_at(reinterpret_cast<const TapPair<T>*>(ref_depth), off);
      td.s = ld_at(reinterpret_cast<const TapPair<T>*>(ref_depth), off_s);
    }
    warp_row(rows[S], t, k == 0);
    if (with_ssim) hsums(rows[S], hs[S]);
    if (k >= 2) stats_row(t - 1, rows[S1], hs[S2], hs[S1], hs[S], ht[S1], bsum_q[S1]);
    if (k >= 4) output_row(t - 2, rows[S2], ht[S], ht[S2], ht[S1], bsum_q[S2], tc, td);
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
          if (v != Cell(0)) atomicAdd(g_scatter + unsigned(wy0 + ly) * unsigned(W) + unsigned(wx0 + lx), T(win_value(v)));
        }
      }
    }
  }
};

template <typename T, unsigned kFlags>
__device__ __forceinline__ void strip_unit(const PairArgs<T>& pa, int b, int seg, int strip, int nbx, int nby, int rs, int B,
                                           int H, int W, unsigned flags_arg, T r_hint,
                                           typename StripCell<T>::type* __restrict__ win) {
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
  s.r0 = seg * rs;
  s.r1 = s.r0 + rs < H ? s.r0 + rs : H;
  s.px = strip * kStripOut - 2 + s.lane;
  s.u = reflect_index(s.px, W);
  s.col_in = s.px >= 0 && s.px < W;
  s.own_x = s.lane >= 2 && s.lane <= kWave - 3 && s.px < W;
  s.uf = T(s.u);
  {
    // M = A K^-1 (BatchConsts): X = (M (u, v, 1)) d + c
    const BatchConsts<T>& bc = pa.consts[b];
    const T* __restrict__ M = bc.M;
    s.qcx = M[0] * s.uf + M[2]; s.qcy = M[3] * s.uf + M[5]; s.qcz = M[6] * s.uf + M[8];
    s.qvx = M[1]; s.qvy = M[4]; s.qvz = M[7];
    s.c0 = bc.c[0]; s.c1 = bc.c[1]; s.c2 = bc.c[2];
  }
  s.inv_w = T(2) / T(W - 1); s.inv_h = T(2) / T(H - 1);
  s.border_cols = strip == 0 || (strip + 1) * kStripOut + 2 >= W;  // wave-uniform: wl / wr may differ from 1 here
  s.r_hint = r_hint;
  s.win = win;
  s.wx0 = 0; s.wy0 = 0;
  for (int i = s.lane; i < kSWinW * kSWinH; i += kWave) win[i] = typename StripCell<T>::type(0);
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    s.rows[q].tap = 0u;  // (the tail's gather is issued every step, also before its first row exists)
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
    const T* __restrict__ Kinv = pa.consts[b].Kinv;
    T gA[12];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j)
        gA[3 * i + j] = s.acc[3 * i] * Kinv[3 * j] + s.acc[3 * i + 1] * Kinv[3 * j + 1] + s.acc[3 * i + 2] * Kinv[3 * j + 2];
    gA[9] = s.acc[9]; gA[10] = s.acc[10]; gA[11] = s.acc[11];
    bool lead;
    const int idx = wave_sum_packed<12>(gA, lead);
    if (lead) pa.gPp[12 * unit + idx] = double(gA[0]);
  }
}

// The speculative forward: kStripWaves independent work units per workgroup, XCD-aware order.
template <typename T, unsigned kFlags = kRuntimeFlags>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? 2 : 1) void pair_strip_kernel(PairBatch<T> pb, int B, int H, int W,
                                                                                      int nbx, int nby, int rs, int nunits,
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
  strip_unit<T, kFlags>(pb.p[pair], b, seg, strip, nbx, nby, rs, B, H, W, flags, r_hint, win[wave]);
}

}  // namespace scsfm
