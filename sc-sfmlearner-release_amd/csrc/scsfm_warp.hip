// inverse_warp2 as maps (inverse_warp.py:230-269) and pose_vec2mat (inverse_warp.py:139-154):
// the un-fused entry points of the boundary.  The training hot path does not go through here --
// it uses the fused pair kernels in scsfm_pair.hip -- but `inverse_warp2` / `pose_vec2mat` are
// public names of the reference's operator API (test_pose.py:71, test_vo.py:76) and the maps are
// what the map-level parity tests compare.
#include "scsfm_geom.h"

namespace scsfm {

template <typename T>
__global__ __launch_bounds__(kThreads) void warp_fwd_kernel(
    int H, int W, unsigned flags, const T* __restrict__ img, const T* __restrict__ depth,
    const T* __restrict__ ref_depth, const BatchConsts<T>* __restrict__ consts, T* __restrict__ out_img,
    T* __restrict__ out_valid, T* __restrict__ out_pdepth, T* __restrict__ out_cdepth) {
  const int b = blockIdx.z;
  const int u = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int v = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  if (u >= W || v >= H) return;
  const BatchConsts<T> bc = consts[b];
  const long plane = (long)H * W, p = (long)v * W + u;
  const Sample<T> s = project_pixel(bc, u, v, depth[b * plane + p], H, W, flags);
#pragma unroll
  for (int c = 0; c < 3; ++c) out_img[(b * 3 + c) * plane + p] = bilerp_rows(load_tap_rows(img + (b * 3 + c) * plane, s), s);
  out_pdepth[b * plane + p] = bilerp_rows(load_tap_rows(ref_depth + b * plane, s), s);
  out_valid[b * plane + p] = s.valid ? T(1) : T(0);
  out_cdepth[b * plane + p] = s.Z;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void warp_bwd_kernel(
    int H, int W, unsigned flags, const T* __restrict__ img, const T* __restrict__ depth,
    const T* __restrict__ ref_depth, const BatchConsts<T>* __restrict__ consts, const T* __restrict__ g_img,
    const T* __restrict__ g_pdepth, const T* __restrict__ g_cdepth, T* __restrict__ g_depth,
    T* __restrict__ g_ref_depth, double* __restrict__ gP) {
  __shared__ double red[12 * (kThreads / kWave)];
  const int b = blockIdx.z;
  const int u = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int v = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  const BatchConsts<T> bc = consts[b];
  const long plane = (long)H * W, p = (long)v * W + u;
  T acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = T(0);
  if (u < W && v < H) {
    const T d = depth[b * plane + p];
    const Sample<T> s = project_pixel(bc, u, v, d, H, W, flags);
    T gix = T(0), giy = T(0), dx, dy;
    if (g_img) {
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        tap_rows_grad(load_tap_rows(img + (b * 3 + c) * plane, s), s, dx, dy);
        const T g = g_img[(b * 3 + c) * plane + p];
        gix += g * dx;
        giy += g * dy;
      }
    }
    if (g_pdepth) {
      tap_rows_grad(load_tap_rows(ref_depth + b * plane, s), s, dx, dy);
      const T g = g_pdepth[b * plane + p];
      gix += g * dx;
      giy += g * dy;
      scatter_taps(g_ref_depth + b * plane, s, g);
    }
    const T gZ = g_cdepth ? g_cdepth[b * plane + p] : T(0);
    g_depth[b * plane + p] += pixel_geometry_bwd(bc, s, u, v, d, gix, giy, gZ, H, W, acc);
  }
  block_sum<12>(acc, red);
  if (threadIdx.x == 0) {
    double g[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) g[i] = double(acc[i]);
    pose_partials_to_A(bc, g);
#pragma unroll
    for (int i = 0; i < 12; ++i) atomicAdd(gP + 12 * b + i, g[i]);
  }
}

// dL/d(sampled image) of inverse_warp2 / inverse_warp: the bilinear splat of dL/d(projected image)
// (grid_sampler_2d_backward on its input; inverse_warp.py:262).  Accumulates with atomics.
template <typename T>
__global__ __launch_bounds__(kThreads) void warp_bwd_image_kernel(int H, int W, unsigned flags, const T* __restrict__ depth,
                                                                  const BatchConsts<T>* __restrict__ consts,
                                                                  const T* __restrict__ g_img, T* __restrict__ g_src) {
  const int b = blockIdx.z;
  const int u = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int v = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  if (u >= W || v >= H) return;
  const BatchConsts<T> bc = consts[b];
  const long plane = (long)H * W, p = (long)v * W + u;
  const Sample<T> s = project_pixel(bc, u, v, depth[b * plane + p], H, W, flags);
#pragma unroll
  for (int c = 0; c < 3; ++c) scatter_taps(g_src + (b * 3 + c) * plane, s, g_img[(b * 3 + c) * plane + p]);
}

// pose_vec2mat forward / backward, one thread per batch element.
template <typename T>
__global__ void pose_mat_fwd_kernel(int B, int mode, const T* __restrict__ vec, T* __restrict__ mat) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const T* p = vec + 6 * b;
  T R[9];
  if (mode == SCSFM_ROT_QUAT) quat_to_R(p[3], p[4], p[5], R); else euler_to_R(p[3], p[4], p[5], R);
  T* m = mat + 12 * b;
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    m[4 * r] = R[3 * r]; m[4 * r + 1] = R[3 * r + 1]; m[4 * r + 2] = R[3 * r + 2]; m[4 * r + 3] = p[r];
  }
}

template <typename T>
__global__ void pose_mat_bwd_kernel(int B, int mode, const T* __restrict__ vec, const T* __restrict__ g_mat,
                                    T* __restrict__ g_vec) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const T* p = vec + 6 * b;
  const T* g = g_mat + 12 * b;
  T gR[9], ga[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) { gR[3 * r] = g[4 * r]; gR[3 * r + 1] = g[4 * r + 1]; gR[3 * r + 2] = g[4 * r + 2]; }
  if (mode == SCSFM_ROT_QUAT) quat_bwd(p[3], p[4], p[5], gR, ga); else euler_bwd(p[3], p[4], p[5], gR, ga);
  T* o = g_vec + 6 * b;
  o[0] = g[3]; o[1] = g[7]; o[2] = g[11];
  o[3] = ga[0]; o[4] = ga[1]; o[5] = ga[2];
}

// ------------------------------------------------------------------------------------------
// Host side of the C ABI.
// ------------------------------------------------------------------------------------------
static inline size_t warp_ws_gP_offset(int B) { return (size_t)B * sizeof(BatchConsts<double>); }

template <typename T>
static int warp_fwd(int B, int H, int W, const T* img, const T* depth, const T* ref_depth, const T* pose,
                    const T* K, unsigned flags, void* ws, T* o_img, T* o_valid, T* o_pd, T* o_cd, void* stream_) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !img || !depth || !ref_depth || !pose || !K || !ws || !o_img || !o_valid ||
      !o_pd || !o_cd)
    return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  auto* consts = reinterpret_cast<BatchConsts<T>*>(ws);
  hipLaunchKernelGGL((prep_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, stream, B, pose, K, consts,
                     (flags & SCSFM_ROT_QUAT_FLAG) ? 1 : 0);
  dim3 grid(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), B);
  hipLaunchKernelGGL((warp_fwd_kernel<T>), grid, dim3(kThreads), 0, stream, H, W, flags,
                     img, depth, ref_depth, (const BatchConsts<T>*)consts, o_img, o_valid, o_pd, o_cd);
  return launch_status();
}

template <typename T>
static int warp_bwd(int B, int H, int W, const T* img, const T* depth, const T* ref_depth, const T* pose,
                    const T* K, unsigned flags, void* ws, const T* g_img, const T* g_pd, const T* g_cd,
                    T* g_depth, T* g_ref_depth, T* g_pose, void* stream_) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !img || !depth || !ref_depth || !pose || !K || !ws || !g_depth || !g_pose)
    return SCSFM_ERR_ARG;
  if (g_pd && !g_ref_depth) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  auto* consts = reinterpret_cast<BatchConsts<T>*>(ws);
  double* gP = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + warp_ws_gP_offset(B));
  hipError_t e = hipMemsetAsync(gP, 0, (size_t)B * 12 * sizeof(double), stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL((prep_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, stream, B, pose, K, consts,
                     (flags & SCSFM_ROT_QUAT_FLAG) ? 1 : 0);
  dim3 grid(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), B);
  hipLaunchKernelGGL((warp_bwd_kernel<T>), grid, dim3(kThreads), 0, stream, H, W, flags,
                     img, depth, ref_depth, (const BatchConsts<T>*)consts, g_img, g_pd, g_cd, g_depth, g_ref_depth,
                     gP);
  hipLaunchKernelGGL((pose_bwd_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, stream, B, pose, K,
                     gP, g_pose, (flags & SCSFM_ROT_QUAT_FLAG) ? 1 : 0);
  return launch_status();
}

// The gradients of the warp's DATA inputs (the reference's autograd reaches them; train.py never asks): after
// warp_bwd on the same workspace.
template <typename T>
static int warp_bwd_inputs(int B, int H, int W, const T* depth, const T* pose, const T* K, unsigned flags, void* ws,
                           const T* g_img, T* g_src, T* g_K, void* stream_) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !pose || !K || !ws || (g_src && (!g_img || !depth))) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  auto* consts = reinterpret_cast<BatchConsts<T>*>(ws);
  const int quat = (flags & SCSFM_ROT_QUAT_FLAG) ? 1 : 0;
  if (g_src) {
    hipLaunchKernelGGL((prep_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, stream, B, pose, K, consts, quat);
    dim3 grid(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), B);
    hipLaunchKernelGGL((warp_bwd_image_kernel<T>), grid, dim3(kThreads), 0, stream, H, W, flags, depth,
                       (const BatchConsts<T>*)consts, g_img, g_src);
  }
  if (g_K) {
    const double* gP = reinterpret_cast<const double*>(reinterpret_cast<const char*>(ws) + warp_ws_gP_offset(B));
    hipLaunchKernelGGL((intrinsics_bwd_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, stream, B, pose, K, gP, g_K, quat);
  }
  return launch_status();
}

// ------------------------------------------------------------------------------------------------------
// pixel2cam (inverse_warp.py:29-44): cam[b, :, v, u] = K^-1_b (u, v, 1) * depth[b, v, u]
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void pixel2cam_fwd_kernel(int H, int W, const T* __restrict__ depth,
                                                                 const T* __restrict__ Kinv, T* __restrict__ cam) {
  const int b = blockIdx.z;
  const int u = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int v = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  if (u >= W || v >= H) return;
  const T* k = Kinv + 9 * b;
  const long plane = (long)H * W, p = (long)v * W + u;
  const T d = depth[b * plane + p], uf = T(u), vf = T(v);
#pragma unroll
  for (int i = 0; i < 3; ++i) cam[(b * 3 + i) * plane + p] = (k[3 * i] * uf + k[3 * i + 1] * vf + k[3 * i + 2]) * d;
}
// dL/d depth = <ray, dL/d cam>  (store)
template <typename T>
__global__ __launch_bounds__(kThreads) void pixel2cam_bwd_kernel(int H, int W, const T* __restrict__ Kinv,
                                                                 const T* __restrict__ g_cam, T* __restrict__ g_depth) {
  const int b = blockIdx.z;
  const int u = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int v = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  if (u >= W || v >= H) return;
  const T* k = Kinv + 9 * b;
  const long plane = (long)H * W, p = (long)v * W + u;
  const T uf = T(u), vf = T(v);
  T g = T(0);
#pragma unroll
  for (int i = 0; i < 3; ++i) g += (k[3 * i] * uf + k[3 * i + 1] * vf + k[3 * i + 2]) * g_cam[(b * 3 + i) * plane + p];
  g_depth[b * plane + p] = g;
}

// dL/d intrinsics_inv [B,3,3] (store) = sum_p dL/d cam(p) (x) (u, v, 1) depth(p): one workgroup per batch element, fp64 sums.
template <typename T>
__global__ __launch_bounds__(kThreads) void pixel2cam_bwd_intrinsics_kernel(int H, int W, const T* __restrict__ depth,
                                                                            const T* __restrict__ g_cam,
                                                                            T* __restrict__ g_Kinv) {
  __shared__ double red[9 * (kThreads / kWave)];
  const int b = blockIdx.x;
  const long plane = (long)H * W;
  double acc[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = 0.0;
  for (long p = threadIdx.x; p < plane; p += kThreads) {
    const int v = int(p / W), u = int(p - (long)v * W);
    const double d = double(depth[b * plane + p]);
    const double q[3] = {double(u) * d, double(v) * d, d};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const double g = double(g_cam[(b * 3 + i) * plane + p]);
      acc[3 * i] += g * q[0]; acc[3 * i + 1] += g * q[1]; acc[3 * i + 2] += g * q[2];
    }
  }
  block_sum<9>(acc, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) g_Kinv[9 * b + i] = T(acc[i]);
  }
}

// ------------------------------------------------------------------------------------------------------
// cam2pixel (inverse_warp.py:47-74) / cam2pixel2 (:194-227): p = rot cam + tr (either may be absent); Z = max(p_z, 1e-3);
// grid[b, v, u] = (2 (X/Z)/(W-1) - 1, 2 (Y/Z)/(H-1) - 1); cam2pixel2 (SCSFM_C2P_OVERWRITE) replaces out-of-range
// coordinates by 2 under zeros padding (no gradient there) and also returns Z.
// ------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kThreads) void cam2pixel_fwd_kernel(int H, int W, unsigned flags, const T* __restrict__ cam,
                                                                 const T* __restrict__ rot, const T* __restrict__ tr,
                                                                 T* __restrict__ grid_out, T* __restrict__ z_out) {
  const int b = blockIdx.z;
  const int u = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int v = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  if (u >= W || v >= H) return;
  const long plane = (long)H * W, p = (long)v * W + u;
  const T c0 = cam[(b * 3) * plane + p], c1 = cam[(b * 3 + 1) * plane + p], c2 = cam[(b * 3 + 2) * plane + p];
  T q[3] = {c0, c1, c2};
  if (rot) {
    const T* r = rot + 9 * b;
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] = r[3 * i] * c0 + r[3 * i + 1] * c1 + r[3 * i + 2] * c2;
  }
  if (tr) {
#pragma unroll
    for (int i = 0; i < 3; ++i) q[i] += tr[3 * b + i];
  }
  const T Z = t_max(q[2], T(kZMin));
  T xn = T(2) * (q[0] / Z) / T(W - 1) - T(1), yn = T(2) * (q[1] / Z) / T(H - 1) - T(1);
  if (flags & SCSFM_C2P_OVERWRITE) {
    if (xn > T(1) || xn < T(-1)) xn = T(2);
    if (yn > T(1) || yn < T(-1)) yn = T(2);
  }
  grid_out[(b * plane + p) * 2] = xn;
  grid_out[(b * plane + p) * 2 + 1] = yn;
  if (z_out) z_out[b * plane + p] = Z;
}

// g_grid [B,H,W,2], g_z [B,1,H,W] (may be NULL) -> g_cam [B,3,H,W] (store), gP [B][12] += (dL/d rot, dL/d tr) in fp64
template <typename T>
__global__ __launch_bounds__(kThreads) void cam2pixel_bwd_kernel(int H, int W, unsigned flags, const T* __restrict__ cam,
                                                                 const T* __restrict__ rot, const T* __restrict__ tr,
                                                                 const T* __restrict__ g_grid, const T* __restrict__ g_z,
                                                                 T* __restrict__ g_cam, double* __restrict__ gP) {
  __shared__ double red[12 * (kThreads / kWave)];
  const int b = blockIdx.z;
  const int u = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int v = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  const long plane = (long)H * W, p = (long)v * W + u;
  T acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = T(0);
  if (u < W && v < H) {
    const T c0 = cam[(b * 3) * plane + p], c1 = cam[(b * 3 + 1) * plane + p], c2 = cam[(b * 3 + 2) * plane + p];
    T q[3] = {c0, c1, c2};
    const T* r = rot ? rot + 9 * b : nullptr;
    if (r) {
#pragma unroll
      for (int i = 0; i < 3; ++i) q[i] = r[3 * i] * c0 + r[3 * i + 1] * c1 + r[3 * i + 2] * c2;
    }
    if (tr) {
#pragma unroll
      for (int i = 0; i < 3; ++i) q[i] += tr[3 * b + i];
    }
    const T Z = t_max(q[2], T(kZMin));
    const T xn = T(2) * (q[0] / Z) / T(W - 1) - T(1), yn = T(2) * (q[1] / Z) / T(H - 1) - T(1);
    T gx = g_grid[(b * plane + p) * 2] * (T(2) / T(W - 1)), gy = g_grid[(b * plane + p) * 2 + 1] * (T(2) / T(H - 1));
    if (flags & SCSFM_C2P_OVERWRITE) {  // overwritten coordinates are constants
      if (xn > T(1) || xn < T(-1)) gx = T(0);
      if (yn > T(1) || yn < T(-1)) gy = T(0);
    }
    const T dq0 = gx / Z, dq1 = gy / Z;
    const T dq2 = q[2] >= T(kZMin) ? (g_z ? g_z[b * plane + p] : T(0)) - (gx * q[0] + gy * q[1]) / (Z * Z) : T(0);
    acc[0] = dq0 * c0; acc[1] = dq0 * c1; acc[2] = dq0 * c2;
    acc[3] = dq1 * c0; acc[4] = dq1 * c1; acc[5] = dq1 * c2;
    acc[6] = dq2 * c0; acc[7] = dq2 * c1; acc[8] = dq2 * c2;
    acc[9] = dq0; acc[10] = dq1; acc[11] = dq2;
    T gc[3] = {dq0, dq1, dq2};
    if (r) {
#pragma unroll
      for (int j = 0; j < 3; ++j) gc[j] = r[j] * dq0 + r[3 + j] * dq1 + r[6 + j] * dq2;
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) g_cam[(b * 3 + j) * plane + p] = gc[j];
  }
  block_sum<12>(acc, red);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 12; ++i) atomicAdd(gP + 12 * b + i, double(acc[i]));
  }
}

template <typename T>
static int pixel2cam_fwd(int B, int H, int W, const T* depth, const T* Kinv, T* cam, void* stream) {
  clear_status();
  if (B <= 0 || H < 1 || W < 1 || !dims_ok<T>(B, H, W) || !depth || !Kinv || !cam) return SCSFM_ERR_ARG;
  dim3 grid(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), B);
  hipLaunchKernelGGL((pixel2cam_fwd_kernel<T>), grid, dim3(kThreads), 0, (hipStream_t)stream, H, W, depth, Kinv, cam);
  return launch_status();
}
template <typename T>
static int pixel2cam_bwd(int B, int H, int W, const T* Kinv, const T* g_cam, T* g_depth, void* stream) {
  clear_status();
  if (B <= 0 || H < 1 || W < 1 || !dims_ok<T>(B, H, W) || !Kinv || !g_cam || !g_depth) return SCSFM_ERR_ARG;
  dim3 grid(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), B);
  hipLaunchKernelGGL((pixel2cam_bwd_kernel<T>), grid, dim3(kThreads), 0, (hipStream_t)stream, H, W, Kinv, g_cam, g_depth);
  return launch_status();
}
template <typename T>
static int pixel2cam_bwd_intrinsics(int B, int H, int W, const T* depth, const T* g_cam, T* g_Kinv, void* stream) {
  clear_status();
  if (B <= 0 || H < 1 || W < 1 || !dims_ok<T>(B, H, W) || !depth || !g_cam || !g_Kinv) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((pixel2cam_bwd_intrinsics_kernel<T>), dim3(B), dim3(kThreads), 0, (hipStream_t)stream, H, W, depth, g_cam,
                     g_Kinv);
  return launch_status();
}
template <typename T>
static int cam2pixel_fwd(int B, int H, int W, const T* cam, const T* rot, const T* tr, unsigned flags, T* grid_out, T* z_out,
                         void* stream) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !cam || !grid_out) return SCSFM_ERR_ARG;
  dim3 grid(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), B);
  hipLaunchKernelGGL((cam2pixel_fwd_kernel<T>), grid, dim3(kThreads), 0, (hipStream_t)stream, H, W, flags, cam, rot, tr,
                     grid_out, z_out);
  return launch_status();
}
template <typename T>
static int cam2pixel_bwd(int B, int H, int W, const T* cam, const T* rot, const T* tr, unsigned flags, const T* g_grid,
                         const T* g_z, T* g_cam, double* gP, void* stream_) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !cam || !g_grid || !g_cam || !gP) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  hipError_t e = hipMemsetAsync(gP, 0, (size_t)B * 12 * sizeof(double), stream);
  if (e != hipSuccess) return (int)e;
  dim3 grid(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), B);
  hipLaunchKernelGGL((cam2pixel_bwd_kernel<T>), grid, dim3(kThreads), 0, stream, H, W, flags, cam, rot, tr, g_grid, g_z, g_cam,
                     gP);
  return launch_status();
}

template <typename T>
static int pose_fwd(int B, const T* vec, int mode, T* mat, void* stream) {
  clear_status();
  if (B <= 0 || !vec || !mat || (mode != SCSFM_ROT_EULER && mode != SCSFM_ROT_QUAT)) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((pose_mat_fwd_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, B, mode, vec,
                     mat);
  return launch_status();
}
template <typename T>
static int pose_bwd(int B, const T* vec, int mode, const T* g_mat, T* g_vec, void* stream) {
  clear_status();
  if (B <= 0 || !vec || !g_mat || !g_vec || (mode != SCSFM_ROT_EULER && mode != SCSFM_ROT_QUAT)) return SCSFM_ERR_ARG;
  hipLaunchKernelGGL((pose_mat_bwd_kernel<T>), dim3(ceil_div(B, 64)), dim3(64), 0, (hipStream_t)stream, B, mode, vec,
                     g_mat, g_vec);
  return launch_status();
}

}  // namespace scsfm

extern "C" {

int scsfm_abi_version(void) { return 9; }

#ifndef SCSFM_SOURCE_ID
#define SCSFM_SOURCE_ID "unknown"
#endif
// (behind a marker, so that scsfm_hip/build.py can read the id from the FILE -- a stale binary may not even load)
static const char g_source_tag[] __attribute__((used)) = "scsfm-source-id:" SCSFM_SOURCE_ID;
int scsfm_source_id(char* buf, size_t n) {
  const volatile char* id = g_source_tag + 16;  // (volatile: the bytes stay in the binary's data, not in immediates)
  if (!buf || n == 0) return SCSFM_ERR_ARG;
  size_t i = 0;
  for (; i + 1 < n && id[i]; ++i) buf[i] = id[i];
  buf[i] = 0;
  return SCSFM_OK;
}

size_t scsfm_warp_ws_bytes(int B) {
  if (B <= 0) return 0;
  return (scsfm::warp_ws_gP_offset(B) + (size_t)B * 12 * sizeof(double) + 255) & ~(size_t)255;
}

#define SCSFM_WARP_API(SUF, T)                                                                                        \
  int scsfm_warp_fwd_##SUF(int B, int H, int W, const T* img, const T* depth, const T* ref_depth, const T* pose,      \
                           const T* K, unsigned flags, void* ws, T* o_img, T* o_valid, T* o_pd, T* o_cd,              \
                           void* stream) {                                                                            \
    return scsfm::warp_fwd<T>(B, H, W, img, depth, ref_depth, pose, K, flags, ws, o_img, o_valid, o_pd, o_cd,         \
                              stream);                                                                                \
  }                                                                                                                   \
  int scsfm_warp_bwd_##SUF(int B, int H, int W, const T* img, const T* depth, const T* ref_depth, const T* pose,      \
                           const T* K, unsigned flags, void* ws, const T* g_img, const T* g_pd, const T* g_cd,        \
                           T* g_depth, T* g_ref_depth, T* g_pose, void* stream) {                                     \
    return scsfm::warp_bwd<T>(B, H, W, img, depth, ref_depth, pose, K, flags, ws, g_img, g_pd, g_cd, g_depth,         \
                              g_ref_depth, g_pose, stream);                                                           \
  }                                                                                                                   \
  int scsfm_warp_bwd_inputs_##SUF(int B, int H, int W, const T* depth, const T* pose, const T* K, unsigned flags,     \
                                  void* ws, const T* g_projected_img, T* g_img, T* g_intrinsics, void* stream) {      \
    return scsfm::warp_bwd_inputs<T>(B, H, W, depth, pose, K, flags, ws, g_projected_img, g_img, g_intrinsics,        \
                                     stream);                                                                         \
  }                                                                                                                   \
  int scsfm_pixel2cam_fwd_##SUF(int B, int H, int W, const T* depth, const T* Kinv, T* cam, void* stream) {           \
    return scsfm::pixel2cam_fwd<T>(B, H, W, depth, Kinv, cam, stream);                                               \
  }                                                                                                                   \
  int scsfm_pixel2cam_bwd_##SUF(int B, int H, int W, const T* Kinv, const T* g_cam, T* g_depth, void* stream) {       \
    return scsfm::pixel2cam_bwd<T>(B, H, W, Kinv, g_cam, g_depth, stream);                                           \
  }                                                                                                                   \
  int scsfm_pixel2cam_bwd_intrinsics_##SUF(int B, int H, int W, const T* depth, const T* g_cam, T* g_intrinsics_inv,  \
                                           void* stream) {                                                            \
    return scsfm::pixel2cam_bwd_intrinsics<T>(B, H, W, depth, g_cam, g_intrinsics_inv, stream);                       \
  }                                                                                                                   \
  int scsfm_cam2pixel_fwd_##SUF(int B, int H, int W, const T* cam, const T* rot, const T* tr, unsigned flags,         \
                                T* grid, T* z, void* stream) {                                                        \
    return scsfm::cam2pixel_fwd<T>(B, H, W, cam, rot, tr, flags, grid, z, stream);                                   \
  }                                                                                                                   \
  int scsfm_cam2pixel_bwd_##SUF(int B, int H, int W, const T* cam, const T* rot, const T* tr, unsigned flags,         \
                                const T* g_grid, const T* g_z, T* g_cam, double* g_rot_tr, void* stream) {            \
    return scsfm::cam2pixel_bwd<T>(B, H, W, cam, rot, tr, flags, g_grid, g_z, g_cam, g_rot_tr, stream);              \
  }                                                                                                                   \
  int scsfm_pose_vec2mat_fwd_##SUF(int B, const T* vec, int mode, T* mat, void* stream) {                             \
    return scsfm::pose_fwd<T>(B, vec, mode, mat, stream);                                                             \
  }                                                                                                                   \
  int scsfm_pose_vec2mat_bwd_##SUF(int B, const T* vec, int mode, const T* g_mat, T* g_vec, void* stream) {           \
    return scsfm::pose_bwd<T>(B, vec, mode, g_mat, g_vec, stream);                                                    \
  }

SCSFM_WARP_API(f32, float)
SCSFM_WARP_API(f64, double)

}  // extern "C"
