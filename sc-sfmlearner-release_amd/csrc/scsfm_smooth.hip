// Edge-aware smoothness of the mean-normalised depth, one frame (get_smooth_loss,
// loss_functions.py:133-152):
//     d = D / (mean_HW(D) + 1e-7);  loss = mean(|dx d| * exp(-mean_c |dx I|)) + mean(|dy d| * exp(-mean_c |dy I|))
// Because the per-image normaliser is positive, |dx d| = |dx D| / den, so one pass over D and I
// yields, per image, sum(D) and the two weighted edge sums; the loss and the mean's contribution
// to the gradient follow from those three numbers:
//     loss   = sum_b L_b / den_b,        L_b = Sx_b / cnt_x + Sy_b / cnt_y
//     dL/dD  = g * [ (1/den_b) * dL_b/dD(p)  -  L_b / (den_b^2 * H * W) ]
#include "scsfm_common.h"

namespace scsfm {

constexpr int kSmRows = 4;  // rows per thread

struct SmoothWs {
  size_t off_img, off_partials, total;
  int nbx, nby;
};
inline SmoothWs smooth_ws_layout(int B, int H, int W) {
  SmoothWs l;
  l.nbx = ceil_div(W, kWave);
  l.nby = ceil_div(H, kSmRows * (kThreads / kWave));
  l.off_img = 0;                                   // double[B][2] = {den_b, L_b}
  l.off_partials = (size_t)B * 2 * sizeof(double); // double[B][nby*nbx][3]
  l.total = (l.off_partials + (size_t)B * l.nbx * l.nby * 3 * sizeof(double) + 255) & ~(size_t)255;
  return l;
}

template <typename T>
__device__ __forceinline__ T edge_weight(const T* __restrict__ img, long plane, long p, long q) {
  const T g = (t_abs(img[p] - img[q]) + t_abs(img[plane + p] - img[plane + q]) +
               t_abs(img[2 * plane + p] - img[2 * plane + q])) / T(3);
  return t_exp(-g);
}

template <typename T>
__global__ __launch_bounds__(kThreads) void smooth_fwd_kernel(int H, int W, const T* __restrict__ depth,
                                                              const T* __restrict__ img,
                                                              double* __restrict__ partials) {
  __shared__ double red[3 * (kThreads / kWave)];
  const int b = blockIdx.z, x = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int y0 = (blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave) * kSmRows;
  const long plane = (long)H * W;
  depth += (long)b * plane;
  img += (long)b * 3 * plane;
  T sd = T(0), sx = T(0), sy = T(0);
  if (x < W) {
#pragma unroll
    for (int r = 0; r < kSmRows; ++r) {
      const int y = y0 + r;
      if (y >= H) break;
      const long p = (long)y * W + x;
      const T d = depth[p];
      sd += d;
      if (x + 1 < W) sx += t_abs(d - depth[p + 1]) * edge_weight(img, plane, p, p + 1);
      if (y + 1 < H) sy += t_abs(d - depth[p + W]) * edge_weight(img, plane, p, p + W);
    }
  }
  T v[3] = {sd, sx, sy};
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    double* o = partials + 3 * ((long)(b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    o[0] = double(v[0]); o[1] = double(v[1]); o[2] = double(v[2]);
  }
}

// One block; each wave reduces whole images (b = wave, wave + 4, ...) with shuffles only, then the
// per-wave loss contributions meet in LDS.
template <typename T>
__global__ __launch_bounds__(kThreads) void smooth_finalize_kernel(int B, int H, int W, int nblk,
                                                                   const double* __restrict__ partials,
                                                                   double* __restrict__ per_img, T* __restrict__ out) {
  __shared__ double red[kThreads / kWave];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const double cnt_x = (double)B * H * (W - 1), cnt_y = (double)B * (H - 1) * W;
  double loss = 0.0;
  for (int b = wave; b < B; b += kThreads / kWave) {
    double v0 = 0, v1 = 0, v2 = 0;
    for (int i = lane; i < nblk; i += kWave) {
      const double* q = partials + 3 * ((size_t)b * nblk + i);
      v0 += q[0]; v1 += q[1]; v2 += q[2];
    }
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
    const double den = v0 / ((double)H * W) + 1e-7;  // mean_HW(D) + 1e-7, loss_functions.py:139-140
    const double L = v1 / cnt_x + v2 / cnt_y;
    if (lane == 0) { per_img[2 * b] = den; per_img[2 * b + 1] = L; }
    loss += L / den;
  }
  if (lane == 0) red[wave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kThreads / kWave; ++w) t += red[w];
    out[0] = T(t);
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void smooth_bwd_kernel(int B, int H, int W, const T* __restrict__ depth,
                                                              const T* __restrict__ img,
                                                              const double* __restrict__ per_img,
                                                              const T* __restrict__ g_loss, T* __restrict__ g_depth) {
  const int b = blockIdx.z, x = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int y0 = (blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave) * kSmRows;
  if (x >= W) return;
  const long plane = (long)H * W;
  depth += (long)b * plane;
  img += (long)b * 3 * plane;
  g_depth += (long)b * plane;
  const T g = g_loss[0];
  const T iden = T(1.0 / per_img[2 * b]);
  const T icx = T(1.0 / ((double)B * H * (W - 1))), icy = T(1.0 / ((double)B * (H - 1) * W));
  const T mean_term = T(per_img[2 * b + 1] / (per_img[2 * b] * per_img[2 * b] * (double)H * W));
#pragma unroll
  for (int r = 0; r < kSmRows; ++r) {
    const int y = y0 + r;
    if (y >= H) break;
    const long p = (long)y * W + x;
    const T d = depth[p];
    T acc = T(0);
    if (x + 1 < W) acc += t_sgn(d - depth[p + 1]) * edge_weight(img, plane, p, p + 1) * icx;
    if (x > 0) acc -= t_sgn(depth[p - 1] - d) * edge_weight(img, plane, p - 1, p) * icx;
    if (y + 1 < H) acc += t_sgn(d - depth[p + W]) * edge_weight(img, plane, p, p + W) * icy;
    if (y > 0) acc -= t_sgn(depth[p - W] - d) * edge_weight(img, plane, p - W, p) * icy;
    g_depth[p] += g * (acc * iden - mean_term);
  }
}

template <typename T>
static int smooth_fwd(int B, int H, int W, const T* depth, const T* img, void* ws, T* out, void* stream_) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !depth || !img || !ws || !out) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const SmoothWs l = smooth_ws_layout(B, H, W);
  char* base = reinterpret_cast<char*>(ws);
  double* per_img = reinterpret_cast<double*>(base + l.off_img);
  double* partials = reinterpret_cast<double*>(base + l.off_partials);
  hipLaunchKernelGGL((smooth_fwd_kernel<T>), dim3(l.nbx, l.nby, B), dim3(kThreads), 0, stream, H, W, depth, img,
                     partials);
  hipLaunchKernelGGL((smooth_finalize_kernel<T>), dim3(1), dim3(kThreads), 0, stream, B, H, W, l.nbx * l.nby,
                     (const double*)partials, per_img, out);
  return launch_status();
}

template <typename T>
static int smooth_bwd(int B, int H, int W, const T* depth, const T* img, void* ws, const T* g_loss, T* g_depth,
                      void* stream_) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !depth || !img || !ws || !g_loss || !g_depth) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const SmoothWs l = smooth_ws_layout(B, H, W);
  const double* per_img = reinterpret_cast<const double*>(reinterpret_cast<char*>(ws) + l.off_img);
  hipLaunchKernelGGL((smooth_bwd_kernel<T>), dim3(l.nbx, l.nby, B), dim3(kThreads), 0, stream, B, H, W, depth, img,
                     per_img, g_loss, g_depth);
  return launch_status();
}

}  // namespace scsfm

extern "C" {

size_t scsfm_smooth_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H < 2 || W < 2) return 0;
  return scsfm::smooth_ws_layout(B, H, W).total;
}

#define SCSFM_SMOOTH_API(SUF, T)                                                                                     \
  int scsfm_smooth_fwd_##SUF(int B, int H, int W, const T* depth, const T* img, void* ws, T* out, void* stream) {    \
    return scsfm::smooth_fwd<T>(B, H, W, depth, img, ws, out, stream);                                               \
  }                                                                                                                  \
  int scsfm_smooth_bwd_##SUF(int B, int H, int W, const T* depth, const T* img, void* ws, const T* g_loss,           \
                             T* g_depth, void* stream) {                                                             \
    return scsfm::smooth_bwd<T>(B, H, W, depth, img, ws, g_loss, g_depth, stream);                                   \
  }

#define SCSFM_SMOOTH_MULTI_API(SUF, T)                                                                               \
  int scsfm_smooth_multi_fwd_##SUF(int n, const void* const* depths, const void* const* imgs, int B, int H, int W,   \
                                   void* ws, T* out, void* stream) {                                                 \
    if (n < 0 || (n > 0 && (!depths || !imgs || !ws || !out))) return SCSFM_ERR_ARG;                                 \
    const size_t stride = scsfm::smooth_ws_layout(B, H, W).total;                                                    \
    for (int i = 0; i < n; ++i) {                                                                                    \
      int rc = scsfm::smooth_fwd<T>(B, H, W, (const T*)depths[i], (const T*)imgs[i], (char*)ws + i * stride,         \
                                    out + i, stream);                                                                \
      if (rc) return rc;                                                                                             \
    }                                                                                                                \
    return SCSFM_OK;                                                                                                 \
  }                                                                                                                  \
  int scsfm_smooth_multi_bwd_##SUF(int n, const void* const* depths, const void* const* imgs, int B, int H, int W,   \
                                   void* ws, const T* g_loss, void* const* g_depths, void* stream) {                 \
    if (n < 0 || (n > 0 && (!depths || !imgs || !ws || !g_loss || !g_depths))) return SCSFM_ERR_ARG;                 \
    const size_t stride = scsfm::smooth_ws_layout(B, H, W).total;                                                    \
    for (int i = 0; i < n; ++i) {                                                                                    \
      if (!g_depths[i]) continue;                                                                                    \
      int rc = scsfm::smooth_bwd<T>(B, H, W, (const T*)depths[i], (const T*)imgs[i], (char*)ws + i * stride, g_loss, \
                                    (T*)g_depths[i], stream);                                                        \
      if (rc) return rc;                                                                                             \
    }                                                                                                                \
    return SCSFM_OK;                                                                                                 \
  }

SCSFM_SMOOTH_MULTI_API(f32, float)
SCSFM_SMOOTH_MULTI_API(f64, double)

SCSFM_SMOOTH_API(f32, float)
SCSFM_SMOOTH_API(f64, double)

}  // extern "C"
