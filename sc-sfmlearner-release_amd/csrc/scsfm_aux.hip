// Stand-alone versions of two public helpers of the reference's loss module: the SSIM layer
// (loss_functions.py:11-45) and mean_on_mask (loss_functions.py:123-129).  The training loss does
// not call them (it runs the same arithmetic fused in scsfm_pair.hip); they exist because the
// names are part of the module's public surface.
#include "scsfm_ssim.h"

namespace scsfm {

// ------------------------------------------------------------------------------------------
// SSIM map.  Planes are independent; tile / strip layout as in the pair kernels.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void ssim_fill_tile(T (*sx)[kHaloW], T (*sy)[kHaloW], const T* __restrict__ x,
                                               const T* __restrict__ y, int ox, int oy, int H, int W, int TH) {
  // (TH+2) x 66 positions, origin (ox-1, oy-1); out-of-image positions hold the reflected sample
  for (int i = threadIdx.x; i < (TH + 2) * kHaloW; i += kThreads) {
    const int hy = i / kHaloW, hx = i - hy * kHaloW;
    const int u = reflect_index(ox + hx - 1, W), v = reflect_index(oy + hy - 1, H);
    sx[hy][hx] = x[(long)v * W + u];
    sy[hy][hx] = y[(long)v * W + u];
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void ssim_fwd_kernel(int H, int W, const T* __restrict__ x,
                                                            const T* __restrict__ y, T* __restrict__ out) {
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ T sx[TH + 2][kHaloW];
  __shared__ T sy[TH + 2][kHaloW];
  const long plane = (long)H * W;
  x += blockIdx.z * plane; y += blockIdx.z * plane; out += blockIdx.z * plane;
  const int col = threadIdx.x & (kWave - 1), strip = threadIdx.x / kWave;
  const int ox = blockIdx.x * kTileW, oy = blockIdx.y * TH;
  ssim_fill_tile(sx, sy, x, y, ox, oy, H, W, TH);
  __syncthreads();
  T hx_[STRIP + 2], hy_[STRIP + 2], hxx[STRIP + 2], hyy[STRIP + 2], hxy[STRIP + 2];
#pragma unroll
  for (int r = 0; r < STRIP + 2; ++r) {
    const int row = strip * STRIP + r;
    const T x0 = sx[row][col], x1 = sx[row][col + 1], x2 = sx[row][col + 2];
    const T y0 = sy[row][col], y1 = sy[row][col + 1], y2 = sy[row][col + 2];
    hx_[r] = x0 + x1 + x2; hy_[r] = y0 + y1 + y2;
    hxx[r] = x0 * x0 + x1 * x1 + x2 * x2; hyy[r] = y0 * y0 + y1 * y1 + y2 * y2; hxy[r] = x0 * y0 + x1 * y1 + x2 * y2;
  }
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int gx = ox + col, gy = oy + strip * STRIP + k;
    if (gx < W && gy < H) {
      const SsimStats<T> st = ssim_stats(hx_[k] + hx_[k + 1] + hx_[k + 2], hy_[k] + hy_[k + 1] + hy_[k + 2],
                                         hxx[k] + hxx[k + 1] + hxx[k + 2], hyy[k] + hyy[k + 1] + hyy[k + 2],
                                         hxy[k] + hxy[k + 1] + hxy[k + 2]);
      out[(long)gy * W + gx] = clamp01(st.raw);
    }
  }
}

template <typename T>
__global__ __launch_bounds__(kThreads) void ssim_bwd_kernel(int H, int W, const T* __restrict__ x,
                                                            const T* __restrict__ y, const T* __restrict__ g_out,
                                                            T* __restrict__ g_x, T* __restrict__ g_y) {
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ T sx[TH + 2][kHaloW];
  __shared__ T sy[TH + 2][kHaloW];
  __shared__ T sG[4][TH][kTileW];  // 1/9 * (g_mu_x, g_mu_y, g_E[x^2] = g_E[y^2], g_E[xy])
  const long plane = (long)H * W;
  x += blockIdx.z * plane; y += blockIdx.z * plane; g_out += blockIdx.z * plane;
  if (g_x) g_x += blockIdx.z * plane;
  if (g_y) g_y += blockIdx.z * plane;
  const int col = threadIdx.x & (kWave - 1), strip = threadIdx.x / kWave;
  const int ox = blockIdx.x * (kTileW - 2) - 1, oy = blockIdx.y * (TH - 2) - 1;
  ssim_fill_tile(sx, sy, x, y, ox, oy, H, W, TH);
  __syncthreads();
  T hx_[STRIP + 2], hy_[STRIP + 2], hxx[STRIP + 2], hyy[STRIP + 2], hxy[STRIP + 2];
#pragma unroll
  for (int r = 0; r < STRIP + 2; ++r) {
    const int row = strip * STRIP + r;
    const T x0 = sx[row][col], x1 = sx[row][col + 1], x2 = sx[row][col + 2];
    const T y0 = sy[row][col], y1 = sy[row][col + 1], y2 = sy[row][col + 2];
    hx_[r] = x0 + x1 + x2; hy_[r] = y0 + y1 + y2;
    hxx[r] = x0 * x0 + x1 * x1 + x2 * x2; hyy[r] = y0 * y0 + y1 * y1 + y2 * y2; hxy[r] = x0 * y0 + x1 * y1 + x2 * y2;
  }
  const int px = ox + col;
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, py = oy + ly;
    T g1 = T(0), g2 = T(0), g3 = T(0), g4 = T(0);
    if (px >= 0 && px < W && py >= 0 && py < H) {
      const SsimStats<T> st = ssim_stats(hx_[k] + hx_[k + 1] + hx_[k + 2], hy_[k] + hy_[k + 1] + hy_[k + 2],
                                         hxx[k] + hxx[k + 1] + hxx[k + 2], hyy[k] + hyy[k + 1] + hyy[k + 2],
                                         hxy[k] + hxy[k + 1] + hxy[k + 2]);
      if (st.raw >= T(0) && st.raw <= T(1)) {
        const T gS = g_out[(long)py * W + px] * T(-0.5);
        const T idd = T(1) / (st.d1 * st.d2), ninth = T(1) / T(9);
        g1 = gS * ((T(2) * st.muy * st.n2 - T(2) * st.muy * st.n1) * idd -
                   st.S * (T(2) * st.mux / st.d1 - T(2) * st.mux / st.d2)) * ninth;
        g2 = gS * ((T(2) * st.mux * st.n2 - T(2) * st.mux * st.n1) * idd -
                   st.S * (T(2) * st.muy / st.d1 - T(2) * st.muy / st.d2)) * ninth;
        g3 = -gS * st.S / st.d2 * ninth;
        g4 = gS * T(2) * st.n1 * idd * ninth;
      }
    }
    sG[0][ly][col] = g1; sG[1][ly][col] = g2; sG[2][ly][col] = g3; sG[3][ly][col] = g4;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, py = oy + ly;
    if (col < 1 || col > kTileW - 2 || ly < 1 || ly > TH - 2 || px >= W || py >= H) continue;
    T s1 = T(0), s2 = T(0), s3 = T(0), s4 = T(0);
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const T wy = reflect_mult<T>(dy, py, H);
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx) {
        const T w = reflect_mult<T>(dx, px, W) * wy;
        s1 += w * sG[0][ly + dy][col + dx];
        s2 += w * sG[1][ly + dy][col + dx];
        s3 += w * sG[2][ly + dy][col + dx];
        s4 += w * sG[3][ly + dy][col + dx];
      }
    }
    const T xv = sx[ly + 1][col + 1], yv = sy[ly + 1][col + 1];
    if (g_x) g_x[(long)py * W + px] = s1 + T(2) * xv * s3 + yv * s4;
    if (g_y) g_y[(long)py * W + px] = s2 + T(2) * yv * s3 + xv * s4;
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

// ------------------------------------------------------------------------------------------
template <typename T>
static int ssim_fwd(int N, int H, int W, const T* x, const T* y, T* out, void* stream) {
  if (N <= 0 || H < 2 || W < 2 || !x || !y || !out) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((ssim_fwd_kernel<T>), dim3(ceil_div(W, kTileW), ceil_div(H, Tile<T>::kH), N), dim3(kThreads), 0,
                     (hipStream_t)stream, H, W, x, y, out);
  return (int)hipGetLastError();
}
template <typename T>
static int ssim_bwd(int N, int H, int W, const T* x, const T* y, const T* g_out, T* g_x, T* g_y, void* stream) {
  if (N <= 0 || H < 2 || W < 2 || !x || !y || !g_out || (!g_x && !g_y)) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((ssim_bwd_kernel<T>), dim3(ceil_div(W, kTileW - 2), ceil_div(H, Tile<T>::kH - 2), N),
                     dim3(kThreads), 0, (hipStream_t)stream, H, W, x, y, g_out, g_x, g_y);
  return (int)hipGetLastError();
}
template <typename T>
static int masked_mean_fwd(int B, int C, int Cm, int HW, const T* diff, const T* mask, void* ws, T* out,
                           void* stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || (Cm != 1 && Cm != C) || !diff || !mask || !ws || !out) return SCSFM_ERR_ARG;
  const long n = (long)B * C * HW;
  const int nb = (int)(n / (kThreads * 4) + 1 < kMmBlocks ? n / (kThreads * 4) + 1 : kMmBlocks);
  double* w = reinterpret_cast<double*>(ws);
  hipLaunchKernelGGL((masked_mean_fwd_kernel<T>), dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, n, C, Cm,
                     (long)HW, diff, mask, w + 4);
  hipLaunchKernelGGL((masked_mean_finalize_kernel<T>), dim3(1), dim3(kThreads), 0, (hipStream_t)stream, nb, w, out);
  return (int)hipGetLastError();
}
template <typename T>
static int masked_mean_bwd(int B, int C, int Cm, int HW, const T* mask, void* ws, const T* g, T* g_diff,
                           void* stream) {
  if (B <= 0 || C <= 0 || HW <= 0 || (Cm != 1 && Cm != C) || !mask || !ws || !g || !g_diff) return SCSFM_ERR_ARG;
  const long n = (long)B * C * HW;
  const int nb = (int)(n / (kThreads * 4) + 1 < 2048 ? n / (kThreads * 4) + 1 : 2048);
  hipLaunchKernelGGL((masked_mean_bwd_kernel<T>), dim3(nb), dim3(kThreads), 0, (hipStream_t)stream, n, C, Cm,
                     (long)HW, mask, (const double*)ws, g, g_diff);
  return (int)hipGetLastError();
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
  }

SCSFM_AUX_API(f32, float)
SCSFM_AUX_API(f64, double)

}  // extern "C"
