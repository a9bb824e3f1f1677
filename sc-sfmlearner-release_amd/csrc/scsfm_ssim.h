// SSIM arithmetic shared by the fused pair kernels and the stand-alone SSIM entry points
// (loss_functions.py:11-42).
#pragma once
#include "scsfm_common.h"

namespace scsfm {

template <typename T> struct Tile { static constexpr int kH = kTileH; };
template <> struct Tile<double> { static constexpr int kH = 8; };  // keeps fp64 LDS under 64 KiB

template <typename T>
struct SsimStats {
  T mux, muy, n1, n2, d1, d2, S, raw;
};

// Five 3x3 window sums -> SSIM terms (loss_functions.py:31-42).
template <typename T>
__device__ __forceinline__ SsimStats<T> ssim_stats(T sx, T sy, T sxx, T syy, T sxy) {
  SsimStats<T> r;
  const T k = T(1) / T(9);
  r.mux = sx * k;
  r.muy = sy * k;
  const T sigx = sxx * k - r.mux * r.mux;
  const T sigy = syy * k - r.muy * r.muy;
  const T sigxy = sxy * k - r.mux * r.muy;
  r.n1 = T(2) * r.mux * r.muy + T(kSsimC1);
  r.n2 = T(2) * sigxy + T(kSsimC2);
  r.d1 = r.mux * r.mux + r.muy * r.muy + T(kSsimC1);
  r.d2 = sigx + sigy + T(kSsimC2);
  r.S = (r.n1 * r.n2) / (r.d1 * r.d2);
  r.raw = (T(1) - r.S) * T(0.5);
  return r;
}

template <typename T>
__device__ __forceinline__ T clamp01(T x) { return t_min(t_max(x, T(0)), T(1)); }

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

}  // namespace scsfm
