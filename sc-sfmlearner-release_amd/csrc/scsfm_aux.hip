// Stand-alone versions of two public helpers of the reference's loss module: the SSIM layer
// (loss_functions.py:11-45) and mean_on_mask (loss_functions.py:123-129).  The training loss does
// not call them (it runs the same arithmetic fused in scsfm_pair.hip); they exist because the
// names are part of the module's public surface.
#include "scsfm_ssim.h"

namespace scsfm {

// ------------------------------------------------------------------------------------------
// SSIM map.  Planes are independent; tile / strip layout as in the pair kernels.
// ------------------------------------------------------------------------------------------
template <typename T, int TH>
__device__ __forceinline__ void ssim_fill_tile(typename Vec2<T>::type (*sxy)[kHaloW], const T* __restrict__ x,
                                               const T* __restrict__ y, int ox, int oy, int H, int W) {
  // (TH+2) x 66 positions, origin (ox-1, oy-1); out-of-image positions hold the reflected sample
  for (int i = threadIdx.x; i < (TH + 2) * kHaloW; i += kThreads) {
    const int hy = i / kHaloW, hx = i - hy * kHaloW;
    const unsigned p = unsigned(reflect_index(oy + hy - 1, H)) * unsigned(W) + unsigned(reflect_index(ox + hx - 1, W));
    sxy[hy][hx] = make2(x[p], y[p]);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void ssim_fwd_kernel(int H, int W, const T* __restrict__ x,
                                                            const T* __restrict__ y, T* __restrict__ out) {
  typedef typename Vec2<T>::type V2;
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ V2 sxy[TH + 2][kHaloW];
  const size_t plane = (size_t)H * W;
  x += blockIdx.z * plane; y += blockIdx.z * plane; out += blockIdx.z * plane;
  const int col = threadIdx.x & (kWave - 1), strip = threadIdx.x / kWave;
  const int ox = blockIdx.x * kTileW, oy = blockIdx.y * TH;
  ssim_fill_tile<T, TH>(sxy, x, y, ox, oy, H, W);
  __syncthreads();
  WinSums<T> ws[STRIP];
  V2 centre[STRIP];
  strip_window_sums<T, STRIP>(sxy, strip * STRIP, col, ws, centre);
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int gx = ox + col, gy = oy + strip * STRIP + k;
    if (gx < W && gy < H) out[(size_t)gy * W + gx] = clamp01(ssim_stats(ws[k]).raw);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void ssim_bwd_kernel(int H, int W, const T* __restrict__ x,
                                                            const T* __restrict__ y, const T* __restrict__ g_out,
                                                            T* __restrict__ g_x, T* __restrict__ g_y) {
  typedef typename Vec2<T>::type V2;
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ V2 sxy[TH + 2][kHaloW];
  __shared__ T sG[4][TH][kTileW];  // 1/9 * (g_mu_x, g_mu_y, g_E[x^2] = g_E[y^2], g_E[xy])
  const size_t plane = (size_t)H * W;
  x += blockIdx.z * plane; y += blockIdx.z * plane; g_out += blockIdx.z * plane;
  if (g_x) g_x += blockIdx.z * plane;
  if (g_y) g_y += blockIdx.z * plane;
  const int col = threadIdx.x & (kWave - 1), strip = threadIdx.x / kWave;
  const int ox = blockIdx.x * (kTileW - 2) - 1, oy = blockIdx.y * (TH - 2) - 1;
  ssim_fill_tile<T, TH>(sxy, x, y, ox, oy, H, W);
  __syncthreads();
  WinSums<T> ws[STRIP];
  V2 centre[STRIP];
  strip_window_sums<T, STRIP>(sxy, strip * STRIP, col, ws, centre);
  const int px = ox + col, py0 = oy + strip * STRIP;
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, py = py0 + k;
    T g1 = T(0), g2 = T(0), g3 = T(0), g4 = T(0);
    if (px >= 0 && px < W && py >= 0 && py < H) {
      const SsimStats<T> st = ssim_stats(ws[k]);
      if (st.raw >= T(0) && st.raw <= T(1)) {
        const T gS = g_out[(size_t)py * W + px] * T(-0.5);
        ssim_grad_y(st, gS, g2, g3, g4);
        ssim_grad_x(st, gS, g1);
      }
    }
    sG[0][ly][col] = g1; sG[1][ly][col] = g2; sG[2][ly][col] = g3; sG[3][ly][col] = g4;
  }
  __syncthreads();
  T gt[STRIP][4];
  strip_box_transpose<T, STRIP, TH, 4>(sG, strip * STRIP, col, px, py0, H, W, gt);
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, py = py0 + k;
    if (col < 1 || col > kTileW - 2 || ly < 1 || ly > TH - 2 || px >= W || py >= H) continue;
    const T xv = centre[k][0], yv = centre[k][1];
    if (g_x) g_x[(size_t)py * W + px] = gt[k][0] + T(2) * xv * gt[k][2] + yv * gt[k][3];
    if (g_y) g_y[(size_t)py * W + px] = gt[k][1] + T(2) * yv * gt[k][2] + xv * gt[k][3];
  }
}

// ------------------------------------------------------------------------------------------
// mean_on_mask
// ------------------------------------------------------------------------------------------
constexpr int kMmBlocks = 1024;

template <typename T>
__global__ __launch_bounds__(kThreads) void masked_mean_fwd_kernel(long n, int C, int Cm, long HW,
                                                                   const T* __restrict__ diff,
                                                                   const T* __restrict__ mask,
                                                                   double* __restrict__ partials) {
  __shared__ double red[2 * (kThreads / kWave)];
  double v[2] = {0, 0};
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const long bc = i / HW, hw = i - bc * HW;
    const long b = bc / C, c = bc - b * C;
    const T m = mask[(b * Cm + (Cm == 1 ? 0 : c)) * HW + hw];
    v[0] += double(diff[i] * m);
    v[1] += double(m);
  }
  block_sum<2>(v, red);
  if (threadIdx.x == 0) { partials[2 * blockIdx.x] = v[0]; partials[2 * blockIdx.x + 1] = v[1]; }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void masked_mean_finalize_kernel(int nblocks, double* __restrict__ ws,
                                                                        T* __restrict__ out) {
  __shared__ double red[2 * (kThreads / kWave)];
  const double* partials = ws + 4;
  double v[2] = {0, 0};
  for (int i = threadIdx.x; i < nblocks; i += kThreads) { v[0] += partials[2 * i]; v[1] += partials[2 * i + 1]; }
  block_sum<2>(v, red);
  if (threadIdx.x == 0) {
    const bool gate = v[1] > kMaskGate;
    ws[0] = v[0]; ws[1] = v[1]; ws[2] = gate ? 1.0 / v[1] : 0.0; ws[3] = 0.0;
    out[0] = gate ? T(v[0] / v[1]) : T(0);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void masked_mean_bwd_kernel(long n, int C, int Cm, long HW,
                                                                   const T* __restrict__ mask,
                                                                   const double* __restrict__ ws,
                                                                   const T* __restrict__ g, T* __restrict__ g_diff) {
  const T scale = g[0] * T(ws[2]);
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < n; i += (long)gridDim.x * kThreads) {
    const long bc = i / HW, hw = i - bc * HW;
    const long b = bc / C, c = bc - b * C;
    g_diff[i] = scale * mask[(b * Cm + (Cm == 1 ? 0 : c)) * HW + hw];
  }
}

// dL/d mask of mean_on_mask (the reference's autograd reaches a floating-point mask: out = sum(diff m) / sum(m) over
// the EXPANDED mask, loss_functions.py:123-129): for an entry of the [B, Cm, HW] mask,
//   g * sum over the channels it is expanded to of (diff / S - N / S^2),   N = sum(diff m), S = sum(m); 0 when gated off.
template <typename T>
__global__ __launch_bounds__(kThreads) void masked_mean_bwd_mask_kernel(long nm, int C, int Cm, long HW,
                                                                        const T* __restrict__ diff,
                                                                        const double* __restrict__ ws,
                                                                        const T* __restrict__ g, T* __restrict__ g_mask) {
  const T inv = T(ws[2]), off = T(ws[0] * ws[2] * ws[2]), gg = g[0];
  for (long i = (long)blockIdx.x * kThreads + threadIdx.x; i < nm; i += (long)gridDim.x * kThreads) {
    const long bc = i / HW, hw = i - bc * HW;
    T v;
    if (Cm == 1) {
      v = T(0);
      for (int c = 0; c < C; ++c) v += diff[(bc * C + c) * HW + hw] * inv - off;
    } else {
      v = diff[i] * inv - off;
    }
    g_mask[i] = inv != T(0) ? gg * v : T(0);
  }
}

// ------------------------------------------------------------------------------------------
template <typename T>
static int ssim_fwd(int N, int H, int W, const T* x, const T* y, T* out, void* stream) {
  clear_status();
  if (N <= 0 || H < 2 || W < 2 || !dims_ok<T>(N, H, W) || !x || !y || !out) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((ssim_fwd_kernel<T>), dim3(ceil_div(W, kTileW), ceil_div(H, Tile<T>::kH), N), dim3(kThreads), 0,
                     (hipStream_t)stream, H, W, x, y, out);
  return launch_status();
}
template <typename T>
static int ssim_bwd(int N, int H, int W, const T* x, const T* y, const T* g_out, T* g_x, T* g_y, void* stream) {
  clear_status();
  if (N <= 0 || H < 2 || W < 2 || !dims_ok<T>(N, H, W) || !x || !y || !g_out || (!g_x && !g_y)) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((ssim_bwd_kernel<T>), dim3(ceil_div(W, kTileW - 2), ceil_div(H, Tile<T>::kH - 2), N),
                     dim3(kThreads), 0, (hipStream_t)stream, H, W, x, y, g_out, g_x, g_y);
  return launch_status();
}
template <typename T>
static int masked_mean_fwd(int B, int C, int Cm, int HW, const T* diff, const T* mask, void* ws, T* out,
                           void* stream) {
  clear_status();
  if (B <= 0 || C <= 0 || HW <= 0 || (Cm != 1 && Cm != C) || !diff || !mask || !ws || !out) return SCSFM_ERR_ARG;
  const long n = (long)B * C * HW;
  const int nb = (int)(n / (kThreads * 4) + 1 < kMmBlocks ? n / (kThreads * 4) + 1 : kMmBlocks);
  double* w = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL((masked_mean_fwd_kernel<T>), dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, n, C, Cm,
                     (long)HW, diff, mask, w + 4);
  hipLaunchKernelGGL((masked_mean_finalize_kernel<T>), dim3(1), dim3(kThreads), 0, (hipStream_t)stream, nb, w, out);
  return launch_status();
}
template <typename T>
static int masked_mean_bwd(int B, int C, int Cm, int HW, const T* mask, void* ws, const T* g, T* g_diff,
                           void* stream) {
  clear_status();
  if (B <= 0 || C <= 0 || HW <= 0 || (Cm != 1 && Cm != C) || !mask || !ws || !g || !g_diff) return SCSFM_ERR_ARG;
  const long n = (long)B * C * HW;
  const int nb = (int)(n / (kThreads * 4) + 1 < 2048 ? n / (kThreads * 4) + 1 : 2048);
  hipLaunchKernelGGL((masked_mean_bwd_kernel<T>), dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, n, C, Cm,
                     (long)HW, mask, (const double*)ws, g, g_diff);
  return launch_status();
}

template <typename T>
static int masked_mean_bwd_mask(int B, int C, int Cm, int HW, const T* diff, void* ws, const T* g, T* g_mask,
                                void* stream) {
  clear_status();
  if (B <= 0 || C <= 0 || HW <= 0 || (Cm != 1 && Cm != C) || !diff || !ws || !g || !g_mask) return SCSFM_ERR_ARG;
  const long nm = (long)B * Cm * HW;
  const int nb = (int)(nm / (kThreads * 4) + 1 < 2048 ? nm / (kThreads * 4) + 1 : 2048);
  hipLaunchKernelGGL((masked_mean_bwd_mask_kernel<T>), dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, nm, C, Cm,
                     (long)HW, diff, (const double*)ws, g, g_mask);
  return launch_status();
}

// loss = w1 * photo + w2 * smooth + w3 * geometry (train.py:268) and, for the backward, the three upstream
// gradients {w1 g, w3 g, w2 g} of the photometric, geometry and smooth terms: one single-thread launch each
// instead of five / three elementwise launches.
template <typename T>
__global__ void step_total_kernel(const T* __restrict__ pg, const T* __restrict__ sm, T w1, T w2, T w3,
                                  T* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) {
    const T photo = pg[0], geom = pg[1], smooth = sm[0];
    out[0] = w1 * photo + w2 * smooth + w3 * geom;
    out[1] = photo; out[2] = smooth; out[3] = geom;
  }
}
template <typename T>
__global__ void step_weights_kernel(const T* __restrict__ g, T w1, T w2, T w3, T* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = w1 * g[0]; out[1] = w3 * g[0]; out[2] = w2 * g[0]; }
}
template <typename T>
static int step_total(const T* pg, const T* sm, double w1, double w2, double w3, T* out, void* stream) {
  clear_status();
  if (!pg || !sm || !out) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((step_total_kernel<T>), dim3(1), dim3(kWave), 0, (hipStream_t)stream, pg, sm, T(w1), T(w2), T(w3), out);
  return launch_status();
}
template <typename T>
static int step_weights(const T* g, double w1, double w2, double w3, T* out, void* stream) {
  clear_status();
  if (!g || !out) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((step_weights_kernel<T>), dim3(1), dim3(kWave), 0, (hipStream_t)stream, g, T(w1), T(w2), T(w3), out);
  return launch_status();
}

}  // namespace scsfm

extern "C" {

size_t scsfm_masked_mean_ws_bytes(void) { return (4 + 2 * (size_t)scsfm::kMmBlocks) * sizeof(double); }

#define SCSFM_AUX_API(SUF, T)                                                                                         \
  int scsfm_ssim_fwd_##SUF(int N, int H, int W, const T* x, const T* y, T* out, void* stream) {                       \
    return scsfm::ssim_fwd<T>(N, H, W, x, y, out, stream);                                                            \
  }                                                                                                                   \
  int scsfm_ssim_bwd_##SUF(int N, int H, int W, const T* x, const T* y, const T* g_out, T* g_x, T* g_y,               \
                           void* stream) {                                                                            \
    return scsfm::ssim_bwd<T>(N, H, W, x, y, g_out, g_x, g_y, stream);                                                \
  }                                                                                                                   \
  int scsfm_masked_mean_fwd_##SUF(int B, int C, int Cm, int HW, const T* diff, const T* mask, void* ws, T* out,       \
                                  void* stream) {                                                                     \
    return scsfm::masked_mean_fwd<T>(B, C, Cm, HW, diff, mask, ws, out, stream);                                      \
  }                                                                                                                   \
  int scsfm_masked_mean_bwd_##SUF(int B, int C, int Cm, int HW, const T* mask, void* ws, const T* g, T* g_diff,       \
                                  void* stream) {                                                                     \
    return scsfm::masked_mean_bwd<T>(B, C, Cm, HW, mask, ws, g, g_diff, stream);                                      \
  }                                                                                                                   \
  int scsfm_masked_mean_bwd_mask_##SUF(int B, int C, int Cm, int HW, const T* diff, void* ws, const T* g, T* g_mask,  \
                                       void* stream) {                                                                \
    return scsfm::masked_mean_bwd_mask<T>(B, C, Cm, HW, diff, ws, g, g_mask, stream);                                 \
  }                                                                                                                   \
  int scsfm_step_total_##SUF(const T* photo_geom, const T* smooth, double w_photo, double w_smooth, double w_geom,    \
                             T* out, void* stream) {                                                                  \
    return scsfm::step_total<T>(photo_geom, smooth, w_photo, w_smooth, w_geom, out, stream);                          \
  }                                                                                                                   \
  int scsfm_step_weights_##SUF(const T* g_loss, double w_photo, double w_smooth, double w_geom, T* out,               \
                               void* stream) {                                                                        \
    return scsfm::step_weights<T>(g_loss, w_photo, w_smooth, w_geom, out, stream);                                    \
  }

SCSFM_AUX_API(f32, float)
SCSFM_AUX_API(f64, double)

}  // extern "C"
