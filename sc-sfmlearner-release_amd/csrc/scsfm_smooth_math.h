// Edge-aware smoothness (get_smooth_loss, loss_functions.py:133-152): the per-pixel arithmetic shared by the smooth
// kernels (scsfm_smooth.hip) and by the speculative forward's tile (scsfm_spec_tile.h), which evaluates the smooth
// loss of its TARGET frame on the way -- the frame's depth and colours are in its registers anyway (round 6).
#pragma once
#include "scsfm_common.h"

namespace scsfm {

template <typename T>
struct Px {  // depth and colours of one pixel
  T d, c0, c1, c2;
};
// exp(-mean_c |I(p) - I(q)|), loss_functions.py:148-152
template <typename T>
__device__ __forceinline__ T edge_weight(const Px<T>& a, const Px<T>& b) {
  return t_exp_weight((t_abs(a.c0 - b.c0) + t_abs(a.c1 - b.c1) + t_abs(a.c2 - b.c2)) * T(-1.0 / 3.0));
}
// sgn(x) for the depth differences of the smooth loss: -1, 0, +1 as t_sgn, in two instructions (a scaling that saturates
// any non-zero fp32 difference of two depths to beyond +-1, then a clamp) instead of two compares and two selects.
// Exact for x = 0 and for |x| >= 2^-100; differences of distinct fp32 depths are >= an ulp of the smaller one.
__device__ __forceinline__ float t_sgn_unit(float x) { return t_med3(x * 1.2676506e30f, -1.0f, 1.0f); }
__device__ __forceinline__ double t_sgn_unit(double x) { return t_sgn(x); }
template <typename T>
__device__ __forceinline__ Px<T> lane_right_px(const Px<T>& v) {
  Px<T> r;
  r.d = lane_right(v.d); r.c0 = lane_right(v.c0); r.c1 = lane_right(v.c1); r.c2 = lane_right(v.c2);
  return r;
}

// One thread's column strip of a tile whose lanes are adjacent columns: rows y0 .. y0 + STRIP - 1 at column x, `row[k]` the
// strip's pixels, `up` / `down` the pixels above and below it (any value where that row lies outside the image).  A pixel is
// OWNED by this thread if own_x (a lane-dependent column predicate that excludes lanes 0 and 63 -- their horizontal
// neighbours are missing) && own_row[k] && y < H.  For owned pixels: sums[0] += D, sums[1] += |dx D| w_x, sums[2] += |dy D| w_y
// over the edges (p, right) and (p, below) that exist, and -- edge != nullptr -- the pixel's summed edge terms
// (+-sgn(dD) w / cnt over its four edges: what the backward streams) are stored.  Exactly the arithmetic, in the order, of
// smooth_fwd_kernel: the two produce the same edge plane bit for bit; the sums differ in grouping only.
template <typename T, int STRIP>
__device__ __forceinline__ void smooth_strip(const Px<T> (&row)[STRIP], const Px<T>& up, const Px<T>& down, int x, int y0,
                                             bool own_x, const bool (&own_row)[STRIP], T icx, T icy, int H, int W,
                                             T* __restrict__ edge, T (&sums)[3]) {
  // icx = T(1 / (B H (W - 1))), icy = T(1 / (B (H - 1) W)): the reciprocal edge counts, rounded once on the host (two fp64
  // divisions per thread otherwise).  Predicates as 0 / 1 factors: the row parts are wave-uniform (scalar selects), a
  // multiplication by 1 is exact and one by 0 gives the 0 the select would (every operand is a finite in-image value).
  const T fx = (x >= 0 && x + 1 < W) ? T(1) : T(0), fo = own_x ? T(1) : T(0);
  T ty_prev = (y0 > 0 && y0 < H) ? t_sgn_unit(up.d - row[0].d) * edge_weight(up, row[0]) * icy : T(0);
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int y = y0 + k;
    const Px<T> cur = row[k], below = k + 1 < STRIP ? row[(k + 1) % STRIP] : down, right = lane_right_px(cur);
    const bool rx = y >= 0 && y < H, ry = y >= 0 && y + 1 < H;  // (uniform) the row has horizontal / downward edges
    const T wx = edge_weight(cur, right) * (rx ? fx : T(0)), wy = ry ? edge_weight(cur, below) : T(0);
    const T dx = cur.d - right.d, dy = cur.d - below.d;
    const T mo = (own_row[k] && y < H) ? fo : T(0);
    sums[0] += cur.d * mo; sums[1] += (t_abs(dx) * wx) * mo; sums[2] += (t_abs(dy) * wy) * mo;
    const T tx = t_sgn_unit(dx) * wx * icx, ty = t_sgn_unit(dy) * wy * icy;
    const T tx_left = lane_left(tx);  // the pixel's left edge is its left neighbour's right edge
    if (edge && own_x && own_row[k] && y < H)
      st_at(edge, (unsigned(y) * unsigned(W) + unsigned(x)) * unsigned(sizeof(T)), tx - tx_left + ty - ty_prev);
    ty_prev = ty;
  }
}

}  // namespace scsfm
