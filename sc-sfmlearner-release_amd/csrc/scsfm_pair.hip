// compute_pairwise_loss (loss_functions.py:95-119) for one (tgt, ref) pair-direction, fused:
// back-projection + SE(3) + projection + bilinear warp of image and depth (inverse_warp.py:230-269),
// valid / auto mask, clamped L1, SSIM (loss_functions.py:11-42), depth inconsistency, the
// self-discovered weight mask and the two masked means (loss_functions.py:123-129).
//
// Forward  : pair_fwd_kernel  -> per-block partial sums {sum photo*m, sum geom*m, sum m}
//            pair_finalize_kernel -> the two gated losses + the backward coefficients (on device;
//            the reference's `if mask.sum() > 10000` host sync disappears)
// Backward : pair_bwd_kernel recomputes the warp inside the tile (nothing but 3 sums is kept from
//            the forward), runs the SSIM backward as a 3x3 gather with reflection multiplicities,
//            writes dL/d tgt_depth densely, scatters dL/d ref_depth with fp32 atomics and reduces
//            dL/d(A|c) per batch element; pose_bwd_kernel finishes dL/d pose.
//
// Tiling (gfx950): 256 threads = 4 waves; a wave spans 64 consecutive pixels of a row (256 B
// coalesced rows), each thread owns a vertical strip of the 64 x TH tile, so the SSIM window sums
// slide down the strip and every LDS read is a conflict-free row access.  The warped image and
// the target image of the tile + 1-pixel ring live in LDS; ring positions outside the image hold
// the reflected pixel (ReflectionPad2d(1)).
#include "scsfm_geom.h"
#include "scsfm_ssim.h"

namespace scsfm {

// Warp one pixel (already reflected into the image): the three warped colours, the target
// colours, and optionally everything else the loss needs at an owned pixel.
template <typename T>
struct PixelOut {
  T Iw[3], It[3];
};

template <typename T>
__device__ __forceinline__ Sample<T> warp_colours(const BatchConsts<T>& bc, int u, int v, int H, int W, unsigned flags,
                                                  const T* __restrict__ tgt_img, const T* __restrict__ ref_img,
                                                  const T* __restrict__ tgt_depth, PixelOut<T>& o) {
  const long plane = (long)H * W, p = (long)v * W + u;
  const Sample<T> s = project_pixel(bc, u, v, tgt_depth[p], H, W, flags);
  T t[4];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    load_taps(ref_img + c * plane, s, W, t);
    o.Iw[c] = bilerp(t, s.fx, s.fy);
    o.It[c] = tgt_img[c * plane + p];
  }
  return s;
}

// ==========================================================================================
// Forward
// ==========================================================================================
template <typename T, bool kSsim>
__global__ __launch_bounds__(kThreads) void pair_fwd_kernel(
    int H, int W, unsigned flags, const T* __restrict__ tgt_img, const T* __restrict__ ref_img,
    const T* __restrict__ tgt_depth, const T* __restrict__ ref_depth, const BatchConsts<T>* __restrict__ consts,
    double* __restrict__ partials) {
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ T sIw[kSsim ? 3 : 1][kSsim ? TH + 2 : 1][kSsim ? kHaloW : 1];
  __shared__ T sIt[kSsim ? 3 : 1][kSsim ? TH + 2 : 1][kSsim ? kHaloW : 1];
  __shared__ double red[3 * (kThreads / kWave)];

  const int b = blockIdx.z, col = threadIdx.x & (kWave - 1), strip = threadIdx.x / kWave;
  const int tx0 = blockIdx.x * kTileW, ty0 = blockIdx.y * TH;
  const bool with_mask = (flags & SCSFM_WITH_MASK) != 0,
             with_auto = (flags & SCSFM_WITH_AUTO_MASK) != 0;
  const BatchConsts<T> bc = consts[b];
  const long plane = (long)H * W;
  tgt_img += (long)b * 3 * plane;
  ref_img += (long)b * 3 * plane;
  tgt_depth += (long)b * plane;
  ref_depth += (long)b * plane;

  T m[STRIP], dd[STRIP], l1[STRIP][3];
  // ---- phase 1a: the pixels this thread owns -------------------------------------------------
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, gx = tx0 + col, gy = ty0 + ly;
    const bool inimg = gx < W && gy < H;
    const int u = reflect_index(gx, W), v = reflect_index(gy, H);
    PixelOut<T> o;
    const Sample<T> s = warp_colours(bc, u, v, H, W, flags, tgt_img, ref_img, tgt_depth, o);
    if (kSsim) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { sIw[c][ly + 1][col + 1] = o.Iw[c]; sIt[c][ly + 1][col + 1] = o.It[c]; }
    }
    m[k] = T(0); dd[k] = T(0);
#pragma unroll
    for (int c = 0; c < 3; ++c) l1[k][c] = clamp01(t_abs(o.It[c] - o.Iw[c]));  // loss_functions.py:99
    if (inimg) {
      T t[4];
      load_taps(ref_depth, s, W, t);
      const T Dp = bilerp(t, s.fx, s.fy);
      dd[k] = clamp01(t_abs(s.Z - Dp) / (s.Z + Dp));  // loss_functions.py:101
      T mk = s.valid ? T(1) : T(0);
      if (with_auto) {  // loss_functions.py:103-105
        const long p = (long)v * W + u;
        const T ident = (t_abs(o.It[0] - ref_img[p]) + t_abs(o.It[1] - ref_img[plane + p]) +
                         t_abs(o.It[2] - ref_img[2 * plane + p])) / T(3);
        const T warped = (l1[k][0] + l1[k][1] + l1[k][2]) / T(3);
        mk = (warped < ident) ? mk : T(0);
      }
      m[k] = mk;
    }
  }
  // ---- phase 1b: the 1-pixel ring (SSIM windows of the tile's border pixels) ------------------
  if (kSsim) {
    if (threadIdx.x < 2 * kHaloW + 2 * TH) {
      int hy, hx;
      ring_pos<TH>(threadIdx.x, hy, hx);
      const int u = reflect_index(tx0 + hx - 1, W), v = reflect_index(ty0 + hy - 1, H);
      PixelOut<T> o;
      warp_colours(bc, u, v, H, W, flags, tgt_img, ref_img, tgt_depth, o);
#pragma unroll
      for (int c = 0; c < 3; ++c) { sIw[c][hy][hx] = o.Iw[c]; sIt[c][hy][hx] = o.It[c]; }
    }
    __syncthreads();
  }
  // ---- phase 2: SSIM down the strip, blend, weight, accumulate -------------------------------
  T acc_p = T(0), acc_g = T(0), acc_m = T(0);
  T photo[STRIP];
#pragma unroll
  for (int k = 0; k < STRIP; ++k) photo[k] = T(0);
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (kSsim) {
      T hx_[STRIP + 2], hy_[STRIP + 2], hxx[STRIP + 2], hyy[STRIP + 2], hxy[STRIP + 2];
#pragma unroll
      for (int r = 0; r < STRIP + 2; ++r) {
        const int row = strip * STRIP + r;
        const T x0 = sIt[c][row][col], x1 = sIt[c][row][col + 1], x2 = sIt[c][row][col + 2];
        const T y0 = sIw[c][row][col], y1 = sIw[c][row][col + 1], y2 = sIw[c][row][col + 2];
        hx_[r] = x0 + x1 + x2;
        hy_[r] = y0 + y1 + y2;
        hxx[r] = x0 * x0 + x1 * x1 + x2 * x2;
        hyy[r] = y0 * y0 + y1 * y1 + y2 * y2;
        hxy[r] = x0 * y0 + x1 * y1 + x2 * y2;
      }
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const SsimStats<T> st = ssim_stats(hx_[k] + hx_[k + 1] + hx_[k + 2], hy_[k] + hy_[k + 1] + hy_[k + 2],
                                           hxx[k] + hxx[k + 1] + hxx[k + 2], hyy[k] + hyy[k + 1] + hyy[k + 2],
                                           hxy[k] + hxy[k + 1] + hxy[k + 2]);
        photo[k] += T(0.15) * l1[k][c] + T(0.85) * clamp01(st.raw);  // loss_functions.py:109
      }
    } else {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) photo[k] += l1[k][c];
    }
  }
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const T w = with_mask ? (T(1) - dd[k]) : T(1);  // loss_functions.py:111-113
    acc_p += photo[k] * w * m[k];
    acc_g += dd[k] * m[k];
    acc_m += m[k];
  }
  double v[3] = {double(acc_p), double(acc_g), double(acc_m)};
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    double* o = partials + 3 * ((long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    o[0] = v[0]; o[1] = v[1]; o[2] = v[2];
  }
}

// The gates and divisions of mean_on_mask (loss_functions.py:123-129) on the three sums; also
// publishes the coefficients the backward multiplies the upstream gradients with.
// out[8] = {photo, geom, S_photo, S_geom, S_mask, 0, 0, 0}
template <typename T>
__device__ __forceinline__ void publish_losses(double Sp, double Sg, double Sm, double* __restrict__ sums,
                                               T* __restrict__ out) {
  // the photo mask is expanded over 3 channels before it is counted (loss_functions.py:124-125)
  const bool gate_p = 3.0 * Sm > kMaskGate, gate_g = Sm > kMaskGate;
  const double photo = gate_p ? Sp / (3.0 * Sm) : 0.0, geom = gate_g ? Sg / Sm : 0.0;
  sums[0] = Sp; sums[1] = Sg; sums[2] = Sm; sums[3] = photo; sums[4] = geom;
  sums[5] = gate_p ? 1.0 / (3.0 * Sm) : 0.0;  // d photo / d (diff_img_c * m)
  sums[6] = gate_g ? 1.0 / Sm : 0.0;          // d geom  / d (diff_depth * m)
  sums[7] = 0.0;
  out[0] = T(photo); out[1] = T(geom); out[2] = T(Sp); out[3] = T(Sg); out[4] = T(Sm);
  out[5] = T(0); out[6] = T(0); out[7] = T(0);
}

// One block: reduce the partials in fp64, apply the gates of mean_on_mask, publish the losses and
// the coefficients the backward multiplies the upstream gradients with.
template <typename T>
__global__ __launch_bounds__(kThreads) void pair_finalize_kernel(int nblocks, const double* __restrict__ partials,
                                                                 double* __restrict__ sums, T* __restrict__ out) {
  __shared__ double red[3 * (kThreads / kWave)];
  double v[3] = {0, 0, 0};
  for (int i = threadIdx.x; i < nblocks; i += kThreads) {
    v[0] += partials[3 * i]; v[1] += partials[3 * i + 1]; v[2] += partials[3 * i + 2];
  }
  block_sum<3>(v, red);
  if (threadIdx.x == 0) publish_losses(v[0], v[1], v[2], sums, out);
}

// Data-parallel "exact" mode: the caller all-reduces out[2..4] (the three raw sums)
// over the ranks, then re-runs the gate / division on the global sums so that both the loss and the
// backward coefficients are those of the concatenated batch.
template <typename T>
__global__ void pair_refinalize_kernel(double* __restrict__ sums, T* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) publish_losses(double(out[2]), double(out[3]), double(out[4]), sums, out);
}

// ==========================================================================================
// Backward
// ==========================================================================================
template <typename T, bool kSsim>
__global__ __launch_bounds__(kThreads) void pair_bwd_kernel(
    int H, int W, unsigned flags, const T* __restrict__ tgt_img, const T* __restrict__ ref_img,
    const T* __restrict__ tgt_depth, const T* __restrict__ ref_depth, const BatchConsts<T>* __restrict__ consts,
    const double* __restrict__ sums, const T* __restrict__ g_photo, const T* __restrict__ g_geom,
    T* __restrict__ g_tgt_depth, T* __restrict__ g_ref_depth, double* __restrict__ gP) {
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ T sIw[kSsim ? 3 : 1][kSsim ? TH + 2 : 1][kSsim ? kHaloW : 1];
  __shared__ T sIt[kSsim ? 3 : 1][kSsim ? TH + 2 : 1][kSsim ? kHaloW : 1];
  __shared__ T sG[kSsim ? 3 : 1][kSsim ? TH : 1][kSsim ? kTileW : 1];
  __shared__ double red[12 * (kThreads / kWave)];

  // upstream gradient x d(masked mean)/d(sum): zero when the 10000-pixel gate was closed
  const T a = T(sums[5]) * g_photo[0];
  const T bg = T(sums[6]) * g_geom[0];
  if (a == T(0) && bg == T(0)) return;  // workgroup-uniform: nothing to propagate

  const int b = blockIdx.z, col = threadIdx.x & (kWave - 1), strip = threadIdx.x / kWave;
  // the 64 x TH compute domain starts one pixel before the 62 x (TH-2) block of outputs
  const int ox = blockIdx.x * (kTileW - 2) - 1, oy = blockIdx.y * (TH - 2) - 1;
  const bool with_mask = (flags & SCSFM_WITH_MASK) != 0,
             with_auto = (flags & SCSFM_WITH_AUTO_MASK) != 0;
  const BatchConsts<T> bc = consts[b];
  const long plane = (long)H * W;
  tgt_img += (long)b * 3 * plane;
  ref_img += (long)b * 3 * plane;
  tgt_depth += (long)b * plane;
  ref_depth += (long)b * plane;
  g_tgt_depth += (long)b * plane;
  g_ref_depth += (long)b * plane;

  T coef[STRIP];             // a * m * (1 - dd): weight of blend_c(q) in the loss
  T mq[STRIP], wq[STRIP];    // mask and (1 - dd) of the owned pixel
  T dIx[STRIP][3], dIy[STRIP][3], l1s[STRIP][3];  // d I_w,c / d(ix, iy); -sgn(It - Iw) gated by the clamp
  T l1v[STRIP][3];
  T gix[STRIP], giy[STRIP], bsum[STRIP];
  bool inimg[STRIP];
  // ---- phase 1a ------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, gx = ox + col, gy = oy + ly;
    inimg[k] = gx >= 0 && gx < W && gy >= 0 && gy < H;
    const int u = reflect_index(gx, W), v = reflect_index(gy, H);
    const long p = (long)v * W + u;
    const Sample<T> s = project_pixel(bc, u, v, tgt_depth[p], H, W, flags);
    T t[4], Iw[3], It[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      load_taps(ref_img + c * plane, s, W, t);
      Iw[c] = bilerp(t, s.fx, s.fy);
      dIx[k][c] = bilerp_dx(t, s.fx, s.fy);
      dIy[k][c] = bilerp_dy(t, s.fx, s.fy);
      It[c] = tgt_img[c * plane + p];
      const T d = It[c] - Iw[c];
      l1v[k][c] = clamp01(t_abs(d));
      // d clamp(|d|, 0, 1) / d Iw: the clamp passes gradient on [0, 1] inclusive, abs uses sgn
      l1s[k][c] = (t_abs(d) <= T(1)) ? -t_sgn(d) : T(0);
      if (kSsim) { sIw[c][ly + 1][col + 1] = Iw[c]; sIt[c][ly + 1][col + 1] = It[c]; }
    }
    mq[k] = T(0); wq[k] = T(1); coef[k] = T(0);
    gix[k] = T(0); giy[k] = T(0); bsum[k] = T(0);
    if (inimg[k]) {
      load_taps(ref_depth, s, W, t);
      const T Dp = bilerp(t, s.fx, s.fy);
      const T ddk = clamp01(t_abs(s.Z - Dp) / (s.Z + Dp));
      T mk = s.valid ? T(1) : T(0);
      if (with_auto) {
        const T ident = (t_abs(It[0] - ref_img[p]) + t_abs(It[1] - ref_img[plane + p]) +
                         t_abs(It[2] - ref_img[2 * plane + p])) / T(3);
        const T warped = (l1v[k][0] + l1v[k][1] + l1v[k][2]) / T(3);
        mk = (warped < ident) ? mk : T(0);
      }
      mq[k] = mk;
      wq[k] = with_mask ? (T(1) - ddk) : T(1);
      coef[k] = a * mk * wq[k];
    }
  }
  // ---- phase 1b: ring ------------------------------------------------------------------------
  if (kSsim) {
    if (threadIdx.x < 2 * kHaloW + 2 * TH) {
      int hy, hx;
      ring_pos<TH>(threadIdx.x, hy, hx);
      const int u = reflect_index(ox + hx - 1, W), v = reflect_index(oy + hy - 1, H);
      PixelOut<T> o;
      warp_colours(bc, u, v, H, W, flags, tgt_img, ref_img, tgt_depth, o);
#pragma unroll
      for (int c = 0; c < 3; ++c) { sIw[c][hy][hx] = o.Iw[c]; sIt[c][hy][hx] = o.It[c]; }
    }
    __syncthreads();
  }
  // ---- phases 2/3, one colour channel at a time ------------------------------------------------
  const int px = ox + col;
  const bool in_x = col >= 1 && col <= kTileW - 2;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    if (kSsim) {
      // phase 2: forward statistics at every owned pixel q; publish 1/9 * (g_mu_y, g_Eyy, g_Exy)(q)
      T hx_[STRIP + 2], hy_[STRIP + 2], hxx[STRIP + 2], hyy[STRIP + 2], hxy[STRIP + 2];
#pragma unroll
      for (int r = 0; r < STRIP + 2; ++r) {
        const int row = strip * STRIP + r;
        const T x0 = sIt[c][row][col], x1 = sIt[c][row][col + 1], x2 = sIt[c][row][col + 2];
        const T y0 = sIw[c][row][col], y1 = sIw[c][row][col + 1], y2 = sIw[c][row][col + 2];
        hx_[r] = x0 + x1 + x2;
        hy_[r] = y0 + y1 + y2;
        hxx[r] = x0 * x0 + x1 * x1 + x2 * x2;
        hyy[r] = y0 * y0 + y1 * y1 + y2 * y2;
        hxy[r] = x0 * y0 + x1 * y1 + x2 * y2;
      }
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int ly = strip * STRIP + k;
        const SsimStats<T> st = ssim_stats(hx_[k] + hx_[k + 1] + hx_[k + 2], hy_[k] + hy_[k + 1] + hy_[k + 2],
                                           hxx[k] + hxx[k + 1] + hxx[k + 2], hyy[k] + hyy[k + 1] + hyy[k + 2],
                                           hxy[k] + hxy[k + 1] + hxy[k + 2]);
        bsum[k] += T(0.15) * l1v[k][c] + T(0.85) * clamp01(st.raw);
        // s = clamp((1 - S)/2, 0, 1): d s / d S = -1/2 inside the clamp (inclusive bounds)
        const T gS = (st.raw >= T(0) && st.raw <= T(1)) ? coef[k] * T(0.85) * T(-0.5) : T(0);
        const T idd = T(1) / (st.d1 * st.d2);
        const T ninth = T(1) / T(9);
        T g1 = T(0), g2 = T(0), g3 = T(0);
        if (gS != T(0)) {
          g1 = gS * ((T(2) * st.mux * st.n2 - T(2) * st.mux * st.n1) * idd -
                     st.S * (T(2) * st.muy / st.d1 - T(2) * st.muy / st.d2)) * ninth;  // d/d mu_y
          g2 = -gS * st.S / st.d2 * ninth;                                              // d/d E[y^2]
          g3 = gS * T(2) * st.n1 * idd * ninth;                                         // d/d E[xy]
        }
        sG[0][ly][col] = g1; sG[1][ly][col] = g2; sG[2][ly][col] = g3;
      }
      __syncthreads();
      // phase 3: transpose of (reflect-pad + 3x3 box): 3x3 gather; an output on the image border
      // is reached twice from its inner neighbour (pad[-1] = x[1])
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const int ly = strip * STRIP + k, py = oy + ly;
        const bool mine = in_x && ly >= 1 && ly <= TH - 2 && inimg[k];
        if (mine) {
          T s1 = T(0), s2 = T(0), s3 = T(0);
#pragma unroll
          for (int dy = -1; dy <= 1; ++dy) {
            const T wy = reflect_mult<T>(dy, py, H);
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
              const T w = reflect_mult<T>(dx, px, W) * wy;
              s1 += w * sG[0][ly + dy][col + dx];
              s2 += w * sG[1][ly + dy][col + dx];
              s3 += w * sG[2][ly + dy][col + dx];
            }
          }
          const T y = sIw[c][ly + 1][col + 1], x = sIt[c][ly + 1][col + 1];
          const T gI = s1 + T(2) * y * s2 + x * s3 + coef[k] * T(0.15) * l1s[k][c];
          gix[k] += gI * dIx[k][c];
          giy[k] += gI * dIy[k][c];
        }
      }
      __syncthreads();
    } else {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        bsum[k] += l1v[k][c];
        const T gI = coef[k] * l1s[k][c];
        gix[k] += gI * dIx[k][c];
        giy[k] += gI * dIy[k][c];
      }
    }
  }
  // ---- phase 4: depth-consistency term, scatter, geometry chain --------------------------------
  T acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = T(0);
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, py = oy + ly;
    const bool mine = in_x && ly >= 1 && ly <= TH - 2 && inimg[k];
    if (!mine) continue;  // note: m(p) = 0 still receives SSIM gradient through its neighbours' windows
    const long p = (long)py * W + px;
    const T d = tgt_depth[p];
    const Sample<T> s = project_pixel(bc, px, py, d, H, W, flags);
    T t[4];
    load_taps(ref_depth, s, W, t);
    const T Dp = bilerp(t, s.fx, s.fy);
    const T diff = s.Z - Dp, sum = s.Z + Dp;
    const T raw = t_abs(diff) / sum;
    // dL/d diff_depth: directly (geometry loss) and through the weight mask (no detach,
    // loss_functions.py:111-113)
    const T g_dd = bg * mq[k] - (with_mask ? a * mq[k] * bsum[k] : T(0));
    T gZ = T(0), gDp = T(0);
    if (raw >= T(0) && raw <= T(1)) {
      const T sg = t_sgn(diff), i2 = T(1) / (sum * sum);
      gZ = g_dd * (sg * T(2) * Dp * i2);
      gDp = -g_dd * (sg * T(2) * s.Z * i2);
    }
    const T gx_ = gix[k] + gDp * bilerp_dx(t, s.fx, s.fy);
    const T gy_ = giy[k] + gDp * bilerp_dy(t, s.fx, s.fy);
    scatter_taps(g_ref_depth, s, W, gDp);
    g_tgt_depth[p] += pixel_geometry_bwd(bc, s, d, gx_, gy_, gZ, H, W, acc);
  }
  double accd[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) accd[i] = double(acc[i]);
  block_sum<12>(accd, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i)
      if (accd[i] != 0.0) atomicAdd(gP + 12 * b + i, accd[i]);
  }
}

// ------------------------------------------------------------------------------------------
// Host side of the C ABI.
// ------------------------------------------------------------------------------------------
template <typename T>
static int pair_fwd(int B, int H, int W, const T* tgt_img, const T* ref_img, const T* tgt_depth, const T* ref_depth,
                    const T* pose, const T* K, unsigned flags, void* ws, T* out, void* stream_) {
  if (B <= 0 || H < 2 || W < 2 || !tgt_img || !ref_img || !tgt_depth || !ref_depth || !pose || !K || !ws || !out)
    return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const PairWs l = pair_ws_layout(B, H, W);
  char* base = reinterpret_cast<char*>(ws);
  auto* consts = reinterpret_cast<BatchConsts<T>*>(base);
  double* sums = reinterpret_cast<double*>(base + l.off_sums);
  double* partials = reinterpret_cast<double*>(base + l.off_partials);
  hipLaunchKernelGGL((prep_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, stream, B, pose, K, consts);
  dim3 grid(ceil_div(W, kTileW), ceil_div(H, Tile<T>::kH), B);
  if (flags & SCSFM_WITH_SSIM)
    hipLaunchKernelGGL((pair_fwd_kernel<T, true>), grid, dim3(kThreads), 0, stream, H, W, flags, tgt_img, ref_img,
                       tgt_depth, ref_depth, (const BatchConsts<T>*)consts, partials);
  else
    hipLaunchKernelGGL((pair_fwd_kernel<T, false>), grid, dim3(kThreads), 0, stream, H, W, flags, tgt_img, ref_img,
                       tgt_depth, ref_depth, (const BatchConsts<T>*)consts, partials);
  hipLaunchKernelGGL((pair_finalize_kernel<T>), dim3(1), dim3(kThreads), 0, stream, (int)(grid.x * grid.y * grid.z),
                     (const double*)partials, sums, out);
  return (int)hipGetLastError();
}

template <typename T>
static int pair_bwd(int B, int H, int W, const T* tgt_img, const T* ref_img, const T* tgt_depth, const T* ref_depth,
                    const T* pose, const T* K, unsigned flags, void* ws, const T* g_photo, const T* g_geom,
                    T* g_tgt_depth, T* g_ref_depth, T* g_pose, void* stream_) {
  if (B <= 0 || H < 2 || W < 2 || !tgt_img || !ref_img || !tgt_depth || !ref_depth || !pose || !K || !ws ||
      !g_photo || !g_geom || !g_tgt_depth || !g_ref_depth || !g_pose)
    return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const PairWs l = pair_ws_layout(B, H, W);
  char* base = reinterpret_cast<char*>(ws);
  auto* consts = reinterpret_cast<const BatchConsts<T>*>(base);
  const double* sums = reinterpret_cast<const double*>(base + l.off_sums);
  double* gP = reinterpret_cast<double*>(base + l.off_gP);
  hipError_t e = hipMemsetAsync(gP, 0, (size_t)B * 12 * sizeof(double), stream);
  if (e != hipSuccess) return (int)e;
  dim3 grid(ceil_div(W, kTileW - 2), ceil_div(H, Tile<T>::kH - 2), B);
  if (flags & SCSFM_WITH_SSIM)
    hipLaunchKernelGGL((pair_bwd_kernel<T, true>), grid, dim3(kThreads), 0, stream, H, W, flags, tgt_img, ref_img,
                       tgt_depth, ref_depth, consts, sums, g_photo, g_geom, g_tgt_depth, g_ref_depth, gP);
  else
    hipLaunchKernelGGL((pair_bwd_kernel<T, false>), grid, dim3(kThreads), 0, stream, H, W, flags, tgt_img, ref_img,
                       tgt_depth, ref_depth, consts, sums, g_photo, g_geom, g_tgt_depth, g_ref_depth, gP);
  hipLaunchKernelGGL((pose_bwd_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, stream, B, pose, K, (const double*)gP,
                     g_pose);
  return (int)hipGetLastError();
}

template <typename T>
static int pair_refinalize(int B, int H, int W, void* ws, T* out, void* stream) {
  if (B <= 0 || H < 2 || W < 2 || !ws || !out) return SCSFM_ERR_ARG;
  const PairWs l = pair_ws_layout(B, H, W);
  double* sums = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + l.off_sums);
  hipLaunchKernelGGL((pair_refinalize_kernel<T>), dim3(1), dim3(kWave), 0, (hipStream_t)stream, sums, out);
  return (int)hipGetLastError();
}

}  // namespace scsfm

extern "C" {

size_t scsfm_pair_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H < 2 || W < 2) return 0;
  // sized for the smaller (fp64) tile so that one workspace serves both precisions
  scsfm::PairWs l = scsfm::pair_ws_layout(B, H, W);
  size_t nb = (size_t)scsfm::ceil_div(W, scsfm::kTileW) * scsfm::ceil_div(H, scsfm::Tile<double>::kH) * B;
  return (l.off_partials + nb * 3 * sizeof(double) + 255) & ~(size_t)255;
}

#define SCSFM_PAIR_API(SUF, T)                                                                                        \
  int scsfm_pair_fwd_##SUF(int B, int H, int W, const T* tgt_img, const T* ref_img, const T* tgt_depth,               \
                           const T* ref_depth, const T* pose, const T* K, unsigned flags, void* ws, T* out,           \
                           void* stream) {                                                                            \
    return scsfm::pair_fwd<T>(B, H, W, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags, ws, out, stream);      \
  }                                                                                                                   \
  int scsfm_pair_bwd_##SUF(int B, int H, int W, const T* tgt_img, const T* ref_img, const T* tgt_depth,               \
                           const T* ref_depth, const T* pose, const T* K, unsigned flags, void* ws,                   \
                           const T* g_photo, const T* g_geom, T* g_tgt_depth, T* g_ref_depth, T* g_pose,              \
                           void* stream) {                                                                            \
    return scsfm::pair_bwd<T>(B, H, W, tgt_img, ref_img, tgt_depth, ref_depth, pose, K, flags, ws, g_photo, g_geom,   \
                              g_tgt_depth, g_ref_depth, g_pose, stream);                                              \
  }

int scsfm_pair_refinalize_f32(int B, int H, int W, void* ws, float* out, void* stream) {
  return scsfm::pair_refinalize<float>(B, H, W, ws, out, stream);
}
int scsfm_pair_refinalize_f64(int B, int H, int W, void* ws, double* out, void* stream) {
  return scsfm::pair_refinalize<double>(B, H, W, ws, out, stream);
}

SCSFM_PAIR_API(f32, float)
SCSFM_PAIR_API(f64, double)

}  // extern "C"
