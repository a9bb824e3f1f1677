// SSIM arithmetic shared by the fused pair kernels and the stand-alone SSIM entry points
// (loss_functions.py:11-42).
//
// The (x, y) = (target, warped) samples of a tile live interleaved in LDS, so one 8-byte LDS read
// feeds both images and the window sums run on 2-wide vectors: on gfx950 the compiler turns those
// into v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (two fp32 lanes per instruction).  The five
// 3x3 window sums slide down each thread's column strip: STRIP + 2 row sums, then 3-row sums.
#pragma once
#include "scsfm_common.h"

namespace scsfm {

template <typename T> struct Tile { static constexpr int kH = kTileH; };
template <> struct Tile<double> { static constexpr int kH = 8; };  // keeps fp64 LDS under 64 KiB

template <typename T> struct Vec2;
template <> struct Vec2<float> { typedef float type __attribute__((vector_size(8))); };
template <> struct Vec2<double> { typedef double type __attribute__((vector_size(16))); };

template <typename T>
__device__ __forceinline__ typename Vec2<T>::type splat2(T a) { typename Vec2<T>::type v = {a, a}; return v; }
template <typename T>
__device__ __forceinline__ typename Vec2<T>::type make2(T a, T b) { typename Vec2<T>::type v = {a, b}; return v; }

template <typename T>
__device__ __forceinline__ T clamp01(T x) { return t_min(t_max(x, T(0)), T(1)); }

// Window sums of one pixel: s1 = (sum x, sum y), s2 = (sum x^2, sum y^2), sxy = sum x*y over the
// reflect-padded 3x3 neighbourhood.
template <typename T>
struct WinSums {
  typename Vec2<T>::type s1, s2;
  T sxy;
};

template <typename T>
struct SsimStats {
  T mux, muy, n1, n2, d1, d2, idd, S, raw;  // idd = 1 / (d1 * d2)
};

// loss_functions.py:31-42.  One reciprocal per pixel and channel.
template <typename T>
__device__ __forceinline__ SsimStats<T> ssim_stats(const WinSums<T>& w) {
  typedef typename Vec2<T>::type V2;
  SsimStats<T> r;
  const T k = T(1) / T(9);
  const V2 mu = w.s1 * splat2(k);
  const V2 musq = mu * mu;
  const V2 sig = w.s2 * splat2(k) - musq;  // (sigma_x, sigma_y) = E[.^2] - mu^2
  const T mxy = mu[0] * mu[1];
  const T sigxy = w.sxy * k - mxy;
  r.mux = mu[0];
  r.muy = mu[1];
  r.n1 = T(2) * mxy + T(kSsimC1);
  r.n2 = T(2) * sigxy + T(kSsimC2);
  r.d1 = musq[0] + musq[1] + T(kSsimC1);
  r.d2 = sig[0] + sig[1] + T(kSsimC2);
  r.idd = t_rcp(r.d1 * r.d2);
  r.S = (r.n1 * r.n2) * r.idd;
  r.raw = (T(1) - r.S) * T(0.5);
  return r;
}

// 1/9 * dS-weighted gradient of the SSIM term with respect to (mu_y, E[y^2], E[xy]) -- what the
// transpose of the box filter scatters back onto the warped image y.  gS = dL/dS.
template <typename T>
__device__ __forceinline__ void ssim_grad_y(const SsimStats<T>& st, T gS, T& g_mu, T& g_e2, T& g_exy) {
  const T q = gS * st.idd * (T(1) / T(9));
  g_mu = T(2) * q * (st.mux * (st.n2 - st.n1) - st.muy * st.S * (st.d2 - st.d1));
  g_e2 = -q * st.S * st.d1;
  g_exy = T(2) * q * st.n1;
}
// ... and with respect to (mu_x, E[x^2]); E[xy] is shared.
template <typename T>
__device__ __forceinline__ void ssim_grad_x(const SsimStats<T>& st, T gS, T& g_mu) {
  const T q = gS * st.idd * (T(1) / T(9));
  g_mu = T(2) * q * (st.muy * (st.n2 - st.n1) - st.mux * st.S * (st.d2 - st.d1));
}

// Window sums down a column strip.  `tile` is one channel plane of interleaved (x, y) samples with a
// one-sample ring: tile[row][col] holds image position (row - 1, col - 1) relative to the tile.
// The strip's first pixel sits at tile row `row0 + 1`, column `col + 1`.
template <typename T, int STRIP>
__device__ __forceinline__ void strip_window_sums(const typename Vec2<T>::type (*tile)[kHaloW], int row0, int col,
                                                  WinSums<T>* out, typename Vec2<T>::type* centre) {
  typedef typename Vec2<T>::type V2;
  V2 h1[STRIP + 2], h2[STRIP + 2];
  T hxy[STRIP + 2];
#pragma unroll
  for (int r = 0; r < STRIP + 2; ++r) {
    const V2 a = tile[row0 + r][col], b = tile[row0 + r][col + 1], c = tile[row0 + r][col + 2];
    h1[r] = a + b + c;
    h2[r] = a * a + b * b + c * c;
    hxy[r] = a[0] * a[1] + b[0] * b[1] + c[0] * c[1];
    if (r >= 1 && r <= STRIP) centre[r - 1] = b;
  }
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    out[k].s1 = h1[k] + h1[k + 1] + h1[k + 2];
    out[k].s2 = h2[k] + h2[k + 1] + h2[k + 2];
    out[k].sxy = hxy[k] + hxy[k + 1] + hxy[k + 2];
  }
}

// Ring position r (0 .. 2*kHaloW + 2*TH - 1) -> (hy, hx) on the border of the (TH+2) x kHaloW tile.
template <int TH>
__device__ __forceinline__ void ring_pos(int r, int& hy, int& hx) {
  if (r < kHaloW) { hy = 0; hx = r; }
  else if (r < 2 * kHaloW) { hy = TH + 1; hx = r - kHaloW; }
  else { r -= 2 * kHaloW; hy = 1 + (r >> 1); hx = (r & 1) ? kHaloW - 1 : 0; }
}

// Weight of output q = p + d in the transpose of (ReflectionPad2d(1) o 3x3 box) at input p, along
// one axis of length n: an output on the image border is reached twice from its inner neighbour.
template <typename T>
__device__ __forceinline__ T reflect_mult(int d, int p, int n) {
  return ((d == -1 && p == 1) || (d == 1 && p == n - 2)) ? T(2) : T(1);
}

// Transposed box filter down a column strip, separably: per map, STRIP + 2 horizontally weighted
// row sums, then STRIP vertical combinations.  g[m] is a TH x kTileW map; (ly0, col) is the strip's
// first pixel in tile coordinates, (px, py0) in image coordinates.  Rows / columns outside the tile
// are clamped: they only reach outputs on the tile's rim, which the caller discards.
template <typename T, int STRIP, int TH, int NMAP>
__device__ __forceinline__ void strip_box_transpose(const T (*g)[TH][kTileW], int ly0, int col, int px, int py0,
                                                    int H, int W, T (*out)[NMAP]) {
  const T wl = reflect_mult<T>(-1, px, W), wr = reflect_mult<T>(1, px, W);
  const int cl = col > 0 ? col - 1 : 0, cr = col < kTileW - 1 ? col + 1 : kTileW - 1;
  T h[NMAP][STRIP + 2];
#pragma unroll
  for (int j = 0; j < STRIP + 2; ++j) {
    int r = ly0 - 1 + j;
    r = r < 0 ? 0 : (r > TH - 1 ? TH - 1 : r);
#pragma unroll
    for (int m = 0; m < NMAP; ++m) h[m][j] = wl * g[m][r][cl] + g[m][r][col] + wr * g[m][r][cr];
  }
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const T wt = reflect_mult<T>(-1, py0 + k, H), wb = reflect_mult<T>(1, py0 + k, H);
#pragma unroll
    for (int m = 0; m < NMAP; ++m) out[k][m] = wt * h[m][k] + h[m][k + 1] + wb * h[m][k + 2];
  }
}

}  // namespace scsfm
