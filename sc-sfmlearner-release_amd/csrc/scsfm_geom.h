// Camera geometry shared by every kernel: SE(3) pose (inverse_warp.py:77-154), back-projection
// (inverse_warp.py:29-44), projection (inverse_warp.py:194-227) and the bilinear sampling
// coordinates of ATen's grid_sampler_2d as the reference calls it (inverse_warp.py:262,267:
// bilinear, align_corners=False, zeros|border).
#pragma once
#include "scsfm_common.h"

namespace scsfm {

// ------------------------------------------------------------------------------------------
// Rotations.  R is row-major r[3*i+j].
// ------------------------------------------------------------------------------------------
// R = Rx(rx) Ry(ry) Rz(rz) in closed form (the reference multiplies the three matrices).
template <typename T>
__device__ __forceinline__ void euler_to_R(T rx, T ry, T rz, T* r) {
  T sx, cx, sy, cy, sz, cz;
  t_sincos(rx, &sx, &cx);
  t_sincos(ry, &sy, &cy);
  t_sincos(rz, &sz, &cz);
  r[0] = cy * cz;                 r[1] = -cy * sz;                r[2] = sy;
  r[3] = cx * sz + sx * sy * cz;  r[4] = cx * cz - sx * sy * sz;  r[5] = -sx * cy;
  r[6] = sx * sz - cx * sy * cz;  r[7] = sx * cz + cx * sy * sz;  r[8] = cx * cy;
}

// gR (dL/dR) -> dL/d(rx,ry,rz)
template <typename T>
__device__ __forceinline__ void euler_bwd(T rx, T ry, T rz, const T* g, T* gang) {
  T sx, cx, sy, cy, sz, cz;
  t_sincos(rx, &sx, &cx);
  t_sincos(ry, &sy, &cy);
  t_sincos(rz, &sz, &cz);
  T r[9];
  euler_to_R(rx, ry, rz, r);
  // d/drx: rows 1,2 rotate into each other
  gang[0] = g[3] * (-r[6]) + g[4] * (-r[7]) + g[5] * (-r[8]) + g[6] * r[3] + g[7] * r[4] + g[8] * r[5];
  gang[1] = g[0] * (-sy * cz) + g[1] * (sy * sz) + g[2] * cy
          + g[3] * (sx * cy * cz) + g[4] * (-sx * cy * sz) + g[5] * (sx * sy)
          + g[6] * (-cx * cy * cz) + g[7] * (cx * cy * sz) + g[8] * (-cx * sy);
  // d/drz: columns 0,1 rotate into each other
  gang[2] = g[0] * r[1] + g[1] * (-r[0]) + g[3] * r[4] + g[4] * (-r[3]) + g[6] * r[7] + g[7] * (-r[6]);
}

// inverse_warp.py:115-136: q = (1, x, y, z) / |(1, x, y, z)|
template <typename T>
__device__ __forceinline__ void quat_to_R(T qx, T qy, T qz, T* r) {
  T n = t_sqrt(T(1) + qx * qx + qy * qy + qz * qz);
  T w = T(1) / n, x = qx / n, y = qy / n, z = qz / n;
  r[0] = w * w + x * x - y * y - z * z;  r[1] = 2 * x * y - 2 * w * z;          r[2] = 2 * w * y + 2 * x * z;
  r[3] = 2 * w * z + 2 * x * y;          r[4] = w * w - x * x + y * y - z * z;  r[5] = 2 * y * z - 2 * w * x;
  r[6] = 2 * x * z - 2 * w * y;          r[7] = 2 * w * x + 2 * y * z;          r[8] = w * w - x * x - y * y + z * z;
}

template <typename T>
__device__ __forceinline__ void quat_bwd(T qx, T qy, T qz, const T* g, T* gq3) {
  T n = t_sqrt(T(1) + qx * qx + qy * qy + qz * qz);
  T w = T(1) / n, x = qx / n, y = qy / n, z = qz / n;
  T gw = 2 * (g[0] * w - g[1] * z + g[2] * y + g[3] * z + g[4] * w - g[5] * x - g[6] * y + g[7] * x + g[8] * w);
  T gx = 2 * (g[0] * x + g[1] * y + g[2] * z + g[3] * y - g[4] * x - g[5] * w + g[6] * z + g[7] * w - g[8] * x);
  T gy = 2 * (-g[0] * y + g[1] * x + g[2] * w + g[3] * x + g[4] * y + g[5] * z - g[6] * w + g[7] * z - g[8] * y);
  T gz = 2 * (-g[0] * z - g[1] * w + g[2] * x + g[3] * w - g[4] * z + g[5] * y + g[6] * x + g[7] * y + g[8] * z);
  // through the normalisation q = u/|u|, u = (1, qx, qy, qz): g_u = (g_q - q <g_q, q>) / |u|
  T dot = gw * w + gx * x + gy * y + gz * z;
  gq3[0] = (gx - x * dot) / n;
  gq3[1] = (gy - y * dot) / n;
  gq3[2] = (gz - z * dot) / n;
}

// ------------------------------------------------------------------------------------------
// Per-batch constants.  One thread per batch element.
// ------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ void prep_one(int b, const T* __restrict__ pose, const T* __restrict__ K,
                                         BatchConsts<T>* __restrict__ out, bool quat = false) {
  const T* k = K + 9 * b;
  // K^-1 by the adjugate, evaluated in fp64 and rounded once (the reference calls
  // torch.inverse, an LU factorisation; both agree to 1 ulp on camera matrices).
  double a = k[0], bb = k[1], c = k[2], d = k[3], e = k[4], f = k[5], g = k[6], h = k[7], i = k[8];
  double C00 = e * i - f * h, C01 = -(d * i - f * g), C02 = d * h - e * g;
  double det = a * C00 + bb * C01 + c * C02;
  double inv = 1.0 / det;
  BatchConsts<T> o;
  o.Kinv[0] = T(C00 * inv);  o.Kinv[1] = T(-(bb * i - c * h) * inv);  o.Kinv[2] = T((bb * f - c * e) * inv);
  o.Kinv[3] = T(C01 * inv);  o.Kinv[4] = T((a * i - c * g) * inv);    o.Kinv[5] = T(-(a * f - c * d) * inv);
  o.Kinv[6] = T(C02 * inv);  o.Kinv[7] = T(-(a * h - bb * g) * inv);  o.Kinv[8] = T((a * e - bb * d) * inv);
  const T* p = pose + 6 * b;
  T R[9];
  // inverse_warp2 always uses euler angles (inverse_warp.py:255); the legacy inverse_warp takes a rotation_mode
  if (quat) quat_to_R(p[3], p[4], p[5], R); else euler_to_R(p[3], p[4], p[5], R);
#pragma unroll
  for (int r = 0; r < 3; ++r) {
#pragma unroll
    for (int j = 0; j < 3; ++j) o.A[3 * r + j] = k[3 * r] * R[j] + k[3 * r + 1] * R[3 + j] + k[3 * r + 2] * R[6 + j];
    o.c[r] = k[3 * r] * p[0] + k[3 * r + 1] * p[1] + k[3 * r + 2] * p[2];
  }
  // M = A K^-1, in fp64 from the rounded A (what the kernels would multiply) and the unrounded inverse
  const double Ki[9] = {C00 * inv, -(bb * i - c * h) * inv, (bb * f - c * e) * inv,
                        C01 * inv, (a * i - c * g) * inv, -(a * f - c * d) * inv,
                        C02 * inv, -(a * h - bb * g) * inv, (a * e - bb * d) * inv};
#pragma unroll
  for (int r = 0; r < 3; ++r)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      o.M[3 * r + j] = T(double(o.A[3 * r]) * Ki[j] + double(o.A[3 * r + 1]) * Ki[3 + j] + double(o.A[3 * r + 2]) * Ki[6 + j]);
  o.pad[0] = o.pad[1] = T(0);
  out[b] = o;
}

template <typename T>
__global__ void prep_kernel(int B, const T* __restrict__ pose, const T* __restrict__ K,
                            BatchConsts<T>* __restrict__ out, int quat) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b < B) prep_one(b, pose, K, out, quat != 0);
}

// dL/d(A|c) (12 numbers, fp64) of one batch element -> dL/dpose: gT = K^T gP, then the euler chain.
template <typename T>
__device__ __forceinline__ void pose_from_gP(const T* __restrict__ k, const T* __restrict__ p, const double* g,
                                             T* __restrict__ o, bool quat = false) {
  T gR[9], gt[3];
#pragma unroll
  for (int kk = 0; kk < 3; ++kk) {
#pragma unroll
    for (int j = 0; j < 3; ++j)
      gR[3 * kk + j] = T(double(k[kk]) * g[j] + double(k[3 + kk]) * g[3 + j] + double(k[6 + kk]) * g[6 + j]);
    gt[kk] = T(double(k[kk]) * g[9] + double(k[3 + kk]) * g[10] + double(k[6 + kk]) * g[11]);
  }
  T ga[3];
  if (quat) quat_bwd(p[3], p[4], p[5], gR, ga); else euler_bwd(p[3], p[4], p[5], gR, ga);
  o[0] = gt[0]; o[1] = gt[1]; o[2] = gt[2];
  o[3] = ga[0]; o[4] = ga[1]; o[5] = ga[2];
}

// gP [B][12] accumulated with atomics (warp_bwd path: zeroed before the accumulation, left in the workspace for
// scsfm_warp_bwd_inputs).
template <typename T>
__global__ void pose_bwd_kernel(int B, const T* __restrict__ pose, const T* __restrict__ K,
                                const double* __restrict__ gP, T* __restrict__ gpose, int quat) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double g[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = gP[12 * b + i];
  pose_from_gP(K + 9 * b, pose + 6 * b, g, gpose + 6 * b, quat != 0);
}
template <typename T> __device__ __forceinline__ void intrinsics_from_gP(const T* __restrict__ k, const T* __restrict__ p,
                                                                         const double* g, bool quat, double* gK);
// ... and dL/d intrinsics [B,3,3] (store) from the same sums.
template <typename T>
__global__ void intrinsics_bwd_kernel(int B, const T* __restrict__ pose, const T* __restrict__ K,
                                      const double* __restrict__ gP, T* __restrict__ gK, int quat) {
  int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  double g[12], o[9];
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = gP[12 * b + i];
#pragma unroll
  for (int i = 0; i < 9; ++i) o[i] = 0.0;
  intrinsics_from_gP(K + 9 * b, pose + 6 * b, g, quat != 0, o);
#pragma unroll
  for (int i = 0; i < 9; ++i) gK[9 * b + i] = T(o[i]);
}

// One wave per batch element: ordered fp64 reduction of the per-block partials gPp[b][nblk][12]
// written by pair_bwd_geom_kernel, then the pose chain.  `live` = the geometry pass ran (it skips
// when both upstream coefficients are zero and then leaves the partials untouched).
template <typename T> struct BatchConsts;
template <typename T, typename A> __device__ __forceinline__ void pose_partials_to_A(const BatchConsts<T>& bc, A* acc);
// The 12 sums of one (pair, batch element): ordered fp64 reduction of the per-block partials (one wave), times the
// factor they still lack, converted from sums against the pixel-frame point to dL/dA (g[0..8]) | dL/dc (g[9..11]).
// Valid in lane 0.
template <typename T>
__device__ __forceinline__ void pose_partials_sum(int b, int nblk, double scale, const BatchConsts<T>* __restrict__ consts,
                                                  const double* __restrict__ gPp, bool live, double (&g)[12]) {
  const int lane = threadIdx.x & (kWave - 1);  // one wave per call (of a 64-thread or a larger workgroup)
#pragma unroll
  for (int i = 0; i < 12; ++i) g[i] = 0.0;
  if (live) {
    for (int j = lane; j < nblk; j += kWave) {
      const double* q = gPp + ((size_t)b * nblk + j) * 12;
#pragma unroll
      for (int i = 0; i < 12; ++i) g[i] += q[i];
    }
#pragma unroll
    for (int i = 0; i < 12; ++i) g[i] = wave_sum(g[i]) * scale;
  }
  // the partials are sums against the pixel-frame point depth * (u, v, 1): dL/dA = G K^-T, once per image
  if (lane == 0) pose_partials_to_A(consts[b], g);
}

template <typename T>
__device__ __forceinline__ void pose_reduce_one(int b, int nblk, double scale, const T* __restrict__ pose,
                                                const T* __restrict__ K, const BatchConsts<T>* __restrict__ consts,
                                                const double* __restrict__ gPp,
                                                const double* __restrict__ sums, const T* __restrict__ g_photo,
                                                const T* __restrict__ g_geom, T* __restrict__ gpose) {
  const bool live = !(T(sums[5]) * g_photo[0] == T(0) && T(sums[6]) * g_geom[0] == T(0));
  double g[12];
  pose_partials_sum(b, nblk, scale, consts, gPp, live, g);
  if ((threadIdx.x & (kWave - 1)) == 0) pose_from_gP(K + 9 * b, pose + 6 * b, g, gpose + 6 * b);
}

// dL/d intrinsics of one batch element from dL/dA | dL/dc.  K enters the warp twice (inverse_warp.py:253-260): the
// camera point is K^-1 (u, v, 1) depth and the projection is (A | c) = K [R | t], so with G = dL/dA, gc = dL/dc
//   dL/dK = G R^T + gc t^T                       (A = K R, c = K t)
//         - K^-T R^T K^T G                       (through torch.inverse: -K^-T (dL/dK^-1) K^-T with dL/dK^-1 = A^T P
//                                                 and P = G K^T the sums against the pixel-frame point)
// evaluated in fp64 and ADDED to gK[9] (row-major).  The reference's autograd produces exactly these nine numbers,
// structural zeros of K included.
template <typename T>
__device__ __forceinline__ void intrinsics_from_gP(const T* __restrict__ k, const T* __restrict__ p, const double* g,
                                                   bool quat, double* gK) {
  double Kd[9], R[9], N[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) Kd[i] = double(k[i]);
  if (quat) quat_to_R(double(p[3]), double(p[4]), double(p[5]), R); else euler_to_R(double(p[3]), double(p[4]), double(p[5]), R);
  {
    const double a = Kd[0], bb = Kd[1], c = Kd[2], d = Kd[3], e = Kd[4], f = Kd[5], gg = Kd[6], h = Kd[7], i = Kd[8];
    const double C00 = e * i - f * h, C01 = -(d * i - f * gg), C02 = d * h - e * gg;
    const double inv = 1.0 / (a * C00 + bb * C01 + c * C02);
    N[0] = C00 * inv; N[1] = -(bb * i - c * h) * inv; N[2] = (bb * f - c * e) * inv;
    N[3] = C01 * inv; N[4] = (a * i - c * gg) * inv;  N[5] = -(a * f - c * d) * inv;
    N[6] = C02 * inv; N[7] = -(a * h - bb * gg) * inv; N[8] = (a * e - bb * d) * inv;
  }
  double X[9], Y[9];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) X[3 * a + b] = Kd[a] * g[b] + Kd[3 + a] * g[3 + b] + Kd[6 + a] * g[6 + b];  // K^T G
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) Y[3 * a + b] = R[a] * X[b] + R[3 + a] * X[3 + b] + R[6 + a] * X[6 + b];      // R^T (K^T G)
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int b = 0; b < 3; ++b) {
      const double through_inverse = N[a] * Y[b] + N[3 + a] * Y[3 + b] + N[6 + a] * Y[6 + b];              // K^-T (...)
      const double left = g[3 * a] * R[3 * b] + g[3 * a + 1] * R[3 * b + 1] + g[3 * a + 2] * R[3 * b + 2];   // G R^T
      gK[3 * a + b] += left + g[9 + a] * double(p[b]) - through_inverse;
    }
}

// ------------------------------------------------------------------------------------------
// One target pixel -> where it lands in the reference view.
//
// Bilinear sampling (ATen grid_sampler_2d, bilinear, align_corners=False, zeros | border) is evaluated on a 2 x 2
// block of pixels that always lies INSIDE the image -- columns xa, xa + 1 with xa = clamp(floor(ix), 0, W - 2), rows
// ya, ya + 1 likewise -- with the hat function max(0, 1 - |ix - column|) as the weight of a column.  For a column
// of the block that is a tap this is exactly the bilinear weight; for one that is not (the sampling position lies
// left of column 0, right of column W - 1, ...) it is 0, and every tap outside the image -- which contributes 0 under
// zeros padding -- is simply never addressed.  No bounds test, no predicated load, no select: the weights cost two
// subtractions with |.| / clamp modifiers each (full-rate VALU on gfx950, where compares, selects and min / max run at
// half rate: tools/ubench).
// ------------------------------------------------------------------------------------------
template <typename T>
struct Sample {
  T qx, qy, qz;     // M (u, v, 1) = A K^-1 (u, v, 1): d(X, Y, Zraw) / d depth
  T X, Y, Zraw, Z;  // depth * q + c = A cam + c ; Z = max(Zraw, 1e-3) is the "computed depth"
  T gmx, gmy;       // d ix / d xn (= W/2), zeroed by the zeros-mode overwrite or the border clip
  int xa, ya;       // first column / row of the 2 x 2 block
  unsigned offr[2]; // element offset of (ya, xa) and (ya + 1, xa) inside a plane: one 8-byte load per row fetches a pair
  T wxa, wxb;       // weights of the block's columns ...
  T wya, wyb;       // ... and rows
  T wp[4];          // their products: weights of (n.a, n.b, s.a, s.b)
  T sxa, sxb;       // d (column weight) / d ix: (-1, +1) between the columns, (+1, 0) left of the image, (0, -1) right of it
  T sya, syb;       // d (row weight) / d iy
  bool valid;       // max(|xn|, |yn|) <= 1   (inverse_warp.py:264)
};


template <typename T>
__device__ __forceinline__ T clamp01_(T x) { return t_min(t_max(x, T(0)), T(1)); }

// The block and its weights along one axis: coordinate `ix` (already overwritten / clipped), image extent n.
template <typename T>
__device__ __forceinline__ void hat_axis(T ix, int n, int& ia, T& wa, T& wb, T& sa, T& sb) {
  const T f0 = t_floor(ix);
  const T fa = t_med3(f0, T(0), T(n - 2));  // (NaN-safe: med3 of a NaN returns a bound)
  const T ta = ix - fa, tb = ta - T(1);
  wa = clamp01_(T(1) - t_abs(ta));
  wb = clamp01_(T(1) - t_abs(tb));
  // the position lies before (floor(ix) = -1 < first column) / beyond (floor(ix) > first column) the block: both
  // differences are small integers, so the clamp is an exact 0 / 1 indicator (ix >= -1 by construction)
  const T L = clamp01_(fa - f0), R = clamp01_(f0 - fa);
  sa = T(2) * L + R - T(1);
  sb = T(1) - L - T(2) * R;
  ia = int(fa);
}

template <typename T>
__device__ __forceinline__ Sample<T> project_pixel(const BatchConsts<T>& bc, int u, int v, T depth,
                                                   int H, int W, unsigned flags) {
  const bool border = (flags & SCSFM_PAD_BORDER) != 0;
  // the legacy inverse_warp (inverse_warp.py:157-191 via cam2pixel :47-74) has no overwrite step
  const bool overwrite = !border && (flags & SCSFM_LEGACY_GRID) == 0;
  Sample<T> s;
  const T uf = T(u), vf = T(v);
  // (column part first: the pixels of a thread's strip share u, so the inner sums are evaluated once per thread)
  s.qx = (bc.M[0] * uf + bc.M[2]) + bc.M[1] * vf;
  s.qy = (bc.M[3] * uf + bc.M[5]) + bc.M[4] * vf;
  s.qz = (bc.M[6] * uf + bc.M[8]) + bc.M[7] * vf;
  s.X = s.qx * depth + bc.c[0];
  s.Y = s.qy * depth + bc.c[1];
  s.Zraw = s.qz * depth + bc.c[2];
  s.Z = t_max(s.Zraw, T(kZMin));
  // xn = 2 (X/Z)/(W-1) - 1 (inverse_warp.py:217-218); the two divisions are a reciprocal of Z and a
  // per-launch constant (1-2 ulp from the reference's correctly rounded quotients)
  const T iz = t_rcp(s.Z);
  const T xn = (s.X * iz) * (T(2) / T(W - 1)) - T(1);
  const T yn = (s.Y * iz) * (T(2) / T(H - 1)) - T(1);
  s.gmx = T(0.5) * T(W);
  s.gmy = T(0.5) * T(H);
  const bool vx = t_abs(xn) <= T(1), vy = t_abs(yn) <= T(1);  // (false for NaN)
  s.valid = vx && vy;
  // grid_sampler_unnormalize, align_corners=False
  T ix = ((xn + T(1)) * T(W) - T(1)) * T(0.5);
  T iy = ((yn + T(1)) * T(H) - T(1)) * T(0.5);
  if (overwrite) {
    // inverse_warp.py:219-224: out-of-range coordinates become the constant 2, a sampling position all of whose taps
    // lie outside the image -- like -1 (weight 1 on column -1, weight 0 on column 0), which keeps the block at the
    // image corner
    if (!vx) { ix = T(-1); s.gmx = T(0); }
    if (!vy) { iy = T(-1); s.gmy = T(0); }
  } else if (border) {  // clip_coordinates_set_grad: zero gradient on and outside the bounds
    if (!(ix > T(0))) { ix = T(0); s.gmx = T(0); } else if (!(ix < T(W - 1))) { ix = T(W - 1); s.gmx = T(0); }
    if (!(iy > T(0))) { iy = T(0); s.gmy = T(0); } else if (!(iy < T(H - 1))) { iy = T(H - 1); s.gmy = T(0); }
  } else {
    // legacy grid, zeros padding, any coordinate: positions beyond [-1, W] sample nothing, exactly like -1 and W --
    // and, none of their taps lying inside the image, they have no gradient (grid_sampler_2d_backward)
    if (!(ix >= T(-1) && ix < T(W))) s.gmx = T(0);
    if (!(iy >= T(-1) && iy < T(H))) s.gmy = T(0);
    ix = t_med3(ix, T(-1), T(W));
    iy = t_med3(iy, T(-1), T(H));
  }
  hat_axis(ix, W, s.xa, s.wxa, s.wxb, s.sxa, s.sxb);
  hat_axis(iy, H, s.ya, s.wya, s.wyb, s.sya, s.syb);
  s.offr[0] = unsigned(s.ya) * unsigned(W) + unsigned(s.xa);
  s.offr[1] = s.offr[0] + unsigned(W);
  s.wp[0] = s.wya * s.wxa; s.wp[1] = s.wya * s.wxb; s.wp[2] = s.wyb * s.wxa; s.wp[3] = s.wyb * s.wxb;
  return s;
}

// The 2 x 2 block of one plane: two 8-byte loads (global_load_dwordx2 needs only dword alignment), unpredicated.
template <typename T>
struct TapPair { T a, b; };
template <typename T>
struct TapRows { TapPair<T> n, s; };
template <typename T>
__device__ __forceinline__ TapRows<T> load_tap_rows(const T* __restrict__ plane, const Sample<T>& s) {
  TapRows<T> r;
  r.n = ld_at(reinterpret_cast<const TapPair<T>*>(plane), s.offr[0] * unsigned(sizeof(T)));
  r.s = ld_at(reinterpret_cast<const TapPair<T>*>(plane), s.offr[1] * unsigned(sizeof(T)));
  return r;
}
// The three colour planes of one image of the batch behind ONE buffer resource (gfx950: buffer_load ... s[rsrc], soffset
// offen): the plane's offset travels in the instruction's SCALAR offset operand, the lane's 32-bit byte offset in its
// vector operand.  With a base pointer per plane the compiler, short of scalar registers in the speculative forward,
// formed 64-bit vector addresses (v_lshl_add_u64, half rate) for most of that kernel's image loads.  The host build
// of the CPU simulation reads through the pointer.
template <typename T>
struct Planes3 {
#if defined(__HIP_DEVICE_COMPILE__)
  __amdgpu_buffer_rsrc_t r;
#else
  const T* p;
#endif
  unsigned pb;  // bytes per plane
};
template <typename T>
__device__ __forceinline__ Planes3<T> planes3(const T* __restrict__ base, unsigned plane) {
  Planes3<T> q;
#if defined(__HIP_DEVICE_COMPILE__)
  // raw buffer, no stride, no bounds (every offset is formed from in-image coordinates); dword 3 as for gfx90a / gfx94x
  q.r = __builtin_amdgcn_make_buffer_rsrc(const_cast<T*>(base), 0, 0xffffffffu, 0x00020000);
#else
  q.p = base;
#endif
  q.pb = plane * unsigned(sizeof(T));
  return q;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ float buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float) {
  return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ double buf_ld(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double) {
  return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ TapPair<float> buf_ld_pair(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, float) {
  return __builtin_bit_cast(TapPair<float>, __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0));
}
__device__ __forceinline__ TapPair<double> buf_ld_pair(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff, double) {
  return __builtin_bit_cast(TapPair<double>, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
#endif
// plane c at byte offset `off` inside a plane
template <typename T>
__device__ __forceinline__ T ld_plane(const Planes3<T>& q, int c, unsigned off) {
#if defined(__HIP_DEVICE_COMPILE__)
  return buf_ld(q.r, off, unsigned(c) * q.pb, T());
#else
  return ld_at(q.p, off + unsigned(c) * q.pb);
#endif
}
// the 2 x 2 block of a sample in plane c
template <typename T>
__device__ __forceinline__ TapRows<T> load_tap_rows(const Planes3<T>& q, int c, const Sample<T>& s) {
  TapRows<T> r;
#if defined(__HIP_DEVICE_COMPILE__)
  r.n = buf_ld_pair(q.r, s.offr[0] * unsigned(sizeof(T)), unsigned(c) * q.pb, T());
  r.s = buf_ld_pair(q.r, s.offr[1] * unsigned(sizeof(T)), unsigned(c) * q.pb, T());
#else
  r.n = ld_at(reinterpret_cast<const TapPair<T>*>(q.p), s.offr[0] * unsigned(sizeof(T)) + unsigned(c) * q.pb);
  r.s = ld_at(reinterpret_cast<const TapPair<T>*>(q.p), s.offr[1] * unsigned(sizeof(T)) + unsigned(c) * q.pb);
#endif
  return r;
}

// A depth map as the pair kernels read it.  Full resolution: the [H, W] plane.  kScaled (kernels instantiated for
// multi-scale steps): the map of a coarser scale, [H >> ds, W >> ds], whose nearest up-sampling to (H, W)
// (loss_functions.py:77-82: F.interpolate(..., mode='nearest') of every scale before compute_pairwise_loss) is
// folded into the index: for H, W multiples of 2^ds the up-sampled map is map[y >> ds][x >> ds].  ds is a property
// of the pair (uniform over a workgroup; 0 for the scale-0 pairs of the same launch).  The full-resolution
// instantiation carries none of this: same addresses and loads as before the scaled one existed.
template <typename T, bool kScaled>
struct DepthMap {
  const T* __restrict__ p;
  int ds;
  unsigned wl;  // row length of the stored map = W >> ds
  // full_off: byte offset of (x, y) in a full-resolution plane (the caller has it for the colour planes anyway)
  __device__ __forceinline__ T at(int x, int y, unsigned full_off) const {
    if (!kScaled) return ld_at(p, full_off);
    return ld_at(p, ((unsigned(y) >> ds) * wl + (unsigned(x) >> ds)) * unsigned(sizeof(T)));
  }
  // the 2 x 2 block a sample reads: two 8-byte loads at full resolution, four scalar loads through the index map
  __device__ __forceinline__ TapRows<T> taps(const Sample<T>& s) const {
    if (!kScaled || ds == 0) return load_tap_rows(p, s);
    TapRows<T> r;
    const unsigned xa = unsigned(s.xa) >> ds, xb = unsigned(s.xa + 1) >> ds;
    const unsigned ra = (unsigned(s.ya) >> ds) * wl, rb = (unsigned(s.ya + 1) >> ds) * wl;
    r.n.a = ld_at(p, (ra + xa) * unsigned(sizeof(T)));
    r.n.b = ld_at(p, (ra + xb) * unsigned(sizeof(T)));
    r.s.a = ld_at(p, (rb + xa) * unsigned(sizeof(T)));
    r.s.b = ld_at(p, (rb + xb) * unsigned(sizeof(T)));
    return r;
  }
};
template <bool kScaled, typename T>
__device__ __forceinline__ DepthMap<T, kScaled> depth_map(const T* maps, int b, int H, int W, int ds) {
  DepthMap<T, kScaled> m;
  m.ds = kScaled ? ds : 0;
  m.wl = unsigned(W) >> m.ds;
  m.p = maps + (size_t)b * ((unsigned(H) >> m.ds) * m.wl);
  return m;
}

// The sampled value.
template <typename T>
__device__ __forceinline__ T bilerp_rows(const TapRows<T>& r, const Sample<T>& s) {
  return r.n.a * s.wp[0] + r.n.b * s.wp[1] + r.s.a * s.wp[2] + r.s.b * s.wp[3];
}
// d(sampled value)/d(ix, iy): out-of-image taps count as the value 0 and have no weight (what
// grid_sampler_2d_backward does for the coordinate gradient) -- the slopes of the hat weights.
template <typename T>
__device__ __forceinline__ void tap_rows_grad(const TapRows<T>& r, const Sample<T>& s, T& dx, T& dy) {
  dx = s.wya * (s.sxa * r.n.a + s.sxb * r.n.b) + s.wyb * (s.sxa * r.s.a + s.sxb * r.s.b);
  dy = s.sya * (s.wxa * r.n.a + s.wxb * r.n.b) + s.syb * (s.wxa * r.s.a + s.wxb * r.s.b);
}

// Gradient of one pixel's sampling position back to the target depth and to A|c.
//   gix, giy : dL/d(ix, iy) (un-normalised sampling coordinates)
//   gZ       : dL/d(computed depth) arriving directly
// Returns dL/d depth(p); adds this pixel's contribution to acc[0..8] = dL/dA, acc[9..11] = dL/dc.
// The partials are accumulated against the PIXEL-frame point depth * (u, v, 1); pose_partials_to_A converts a
// block's sums to dL/dA (cam = K^-1 (u, v, 1) depth is linear in it).
template <typename T>
__device__ __forceinline__ T pixel_geometry_bwd(const BatchConsts<T>& bc, const Sample<T>& s, int u, int v, T depth,
                                                T gix, T giy, T gZ, int H, int W, T* acc) {
  // ix = ((xn+1) W - 1)/2, xn = 2 (X/Z)/(W-1) - 1   (inverse_warp.py:217-218)
  const T gqx = gix * s.gmx * (T(2) / T(W - 1));
  const T gqy = giy * s.gmy * (T(2) / T(H - 1));
  const T iz = t_rcp(s.Z);
  const T dX = gqx * iz;
  const T dY = gqy * iz;
  // Z = clamp(Zraw, min=1e-3): gradient passes where Zraw >= 1e-3 (inverse_warp.py:211)
  const T dZ = (s.Zraw >= T(kZMin)) ? (gZ - (gqx * s.X + gqy * s.Y) * iz * iz) : T(0);
  const T pu = T(u) * depth, pv = T(v) * depth;
  acc[0] += dX * pu; acc[1] += dX * pv; acc[2] += dX * depth;
  acc[3] += dY * pu; acc[4] += dY * pv; acc[5] += dY * depth;
  acc[6] += dZ * pu; acc[7] += dZ * pv; acc[8] += dZ * depth;
  acc[9] += dX; acc[10] += dY; acc[11] += dZ;
  return s.qx * dX + s.qy * dY + s.qz * dZ;  // (X, Y, Zraw) = depth * q + c
}
// dL/dA[i][j] = sum_k G[i][k] K^-1[j][k] for G = the sums of pixel_geometry_bwd's acc[0..8]; acc[9..11] = dL/dc stay.
template <typename T, typename A>
__device__ __forceinline__ void pose_partials_to_A(const BatchConsts<T>& bc, A* acc) {
  A g[9];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
      g[3 * i + j] = acc[3 * i] * A(bc.Kinv[3 * j]) + acc[3 * i + 1] * A(bc.Kinv[3 * j + 1]) + acc[3 * i + 2] * A(bc.Kinv[3 * j + 2]);
#pragma unroll
  for (int i = 0; i < 9; ++i) acc[i] = g[i];
}

// Scatter dL/d(projected depth) of one pixel into the gradient of the sampled depth map
// (grid_sampler_2d_backward on the input; only in-image taps receive anything).
template <typename T>
__device__ __forceinline__ void scatter_taps(T* __restrict__ gplane, const Sample<T>& s, T g) {
  if (g == T(0)) return;
  // (a cell of the block that is not a tap has weight 0)
  if (s.wp[0] != T(0)) atomicAdd(gplane + s.offr[0], g * s.wp[0]);
  if (s.wp[1] != T(0)) atomicAdd(gplane + s.offr[0] + 1, g * s.wp[1]);
  if (s.wp[2] != T(0)) atomicAdd(gplane + s.offr[1], g * s.wp[2]);
  if (s.wp[3] != T(0)) atomicAdd(gplane + s.offr[1] + 1, g * s.wp[3]);
}


// The same scatter, staged through an LDS window that covers where a tile of neighbouring pixels
// lands (motion is locally coherent): taps inside the window are LDS atomics, the rest fall back to
// global atomics.  The window is flushed once per block with coalesced atomics (flush_scatter_window),
// which cuts the device-scope atomic traffic from 4 per source pixel to ~1 per touched destination.
#ifndef SCSFM_WIN_W  // tuning knobs (tools/build_variants.sh); the defaults are the product
#define SCSFM_WIN_W 96
#define SCSFM_WIN_H 32
#endif
constexpr int kWinW = SCSFM_WIN_W, kWinH = SCSFM_WIN_H;  // window of a 64 x 16 tile; WH scales with the tile height
#ifndef SCSFM_GEOM_ROWS
#define SCSFM_GEOM_ROWS 4
#endif
constexpr int kGeomRows = SCSFM_GEOM_ROWS;  // rows per thread of the geometry pass (its tile is 64 x 4 kGeomRows)

template <typename T>
__device__ __forceinline__ void lds_add(T* p, T v) {
  (void)__hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// Cells of the staging windows.  fp32 speculative forward: 32-bit FIXED POINT, because on gfx950 the LDS float atomic is
// executed lane by lane (ds_add_f32: 1.25 ns per LANE per CU, measured with tools/ubench/rates2.hip) while the integer one
// is a single pass (ds_add_u32: 2 ns per wave instruction); contributions are rounded to nearest, so a cell's error is at
// most half a unit per contribution whatever the order of the additions -- the window part of the scatter is
// bit-reproducible.  fp64 (gradient-check builds) and the fallback geometry pass: floating cells.
//
// Range (round 6: an exact, always-on guarantee instead of round 5's bounding-box heuristic).  A cell counts in units of
// 2^-e, and e is chosen PER TILE from an upper bound of everything the tile can add to ONE cell: every pixel that enters
// the window adds at most |g| (its four weights are <= 1) with |g| < kFixCap (larger terms take the direct fp32 atomics),
// and |g| = |dL/d diff_depth| 2 Z / (Z + D_p)^2 <= (|r| + 3 [weight mask]) 2 Z / (Z + D_p)^2 =: u -- known in the warp
// phase, when Z and D_p are in registers (|dL/d diff_depth| <= |r| m + m sum_c blend_c and every blend_c <= 1).  With
// U = sum over the tile's owned, unmasked pixels of min(u, kFixCap), no cell can exceed U 2^e + 4 * 868 / 2 (rounding) in
// magnitude, so e = min(20, floor(log2(kFixLimit / U))) cannot wrap -- whatever the warp does to the tile (round 5 found
// a reachable wrap: a scene scaled down a hundredfold after a forward motion of several depths; a NON-uniform compression
// inside a large footprint escaped its heuristic).  U <= 2002 -- every tile of an ordinary warp -- keeps e = 20: the same
// cells, bit for bit, as before.  Cost: 7 vector instructions per pixel and one more value in the bounding box's wave
// reduction.  Launches with SCSFM_DEBUG_CHECK_WINDOW still count every wrap exactly (win_add): the count is now 0 by
// construction, and -DSCSFM_WINDOW_BOUND=0 (tests) switches the bound off to show that the detector works.
#ifndef SCSFM_WINDOW_BOUND
#define SCSFM_WINDOW_BOUND 1
#endif
constexpr float kFixScale = 1048576.0f;  // 2^20: the unit of a tile whose bound allows it (all but pathological warps)
constexpr float kFixInv = 1.0f / 1048576.0f;
constexpr float kFixCap = 64.0f;
constexpr float kFixLimit = 2.1e9f;  // < 2^31 - 4 * 1024 * 0.5 (rounding of every contribution) with room for the bound's own rounding
// (scale, 1 / scale) of a tile from its bound U (uniform over the workgroup)
__device__ __forceinline__ void win_units_of(float U, float& scale, float& inv) {
  scale = kFixScale; inv = kFixInv;
#if SCSFM_WINDOW_BOUND
  const float room = kFixLimit / (U * 1.001f);  // (1.001: the bound is evaluated in fp32, a few ulp either way)
  if (!(room >= kFixScale)) {  // (also taken by a NaN bound: the coarsest unit)
    // 2^floor(log2(room)): the exponent field of room, mantissa cleared; never below 2^-20 (U <= 1024 * kFixCap)
    const unsigned bits = __builtin_bit_cast(unsigned, room > 9.5367431640625e-7f ? room : 9.5367431640625e-7f) & 0x7f800000u;
    scale = __builtin_bit_cast(float, bits);
    inv = __builtin_bit_cast(float, (254u << 23) - bits);  // 2^-k for scale = 2^k
  }
#endif
}
__device__ __forceinline__ bool win_fits(const int*, float g) { return t_abs(g) < kFixCap; }
__device__ __forceinline__ bool win_fits(const double*, double) { return true; }
__device__ __forceinline__ bool win_fits(const float*, float) { return true; }
template <typename T> struct WinCell { typedef int type; };
template <> struct WinCell<double> { typedef double type; };
// floor(x + 0.5) as an integer: ONE instruction on gfx950 (v_cvt_rpi_i32_f32, "round to plus infinity on ties") instead
// of add, floor, convert -- the window takes 16 of these per thread
__device__ __forceinline__ int cvt_round_half_up(float x) {
#if defined(__HIP_DEVICE_COMPILE__)
  int r;
  asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
  return r;
#else
  return (int)floorf(x + 0.5f);
#endif
}
// The scale a window of these cells counts in when the caller names none: 2^20 for fixed point, 1 for floating cells.
template <typename Cell> struct WinScale0 { static constexpr double value = 1.0; };
template <> struct WinScale0<int> { static constexpr double value = 1048576.0; };
template <typename Cell, typename T> __device__ __forceinline__ T win_scale0() { return T(WinScale0<Cell>::value); }
// ... added to a cell
// `ovf` (SCSFM_DEBUG_CHECK_WINDOW launches only; nullptr -- a compile-time constant in the product instantiation --
// otherwise): the add returns the cell's previous value and a signed 32-bit wrap is counted in *ovf (a global word of
// the pair's workspace).  Exact: every wrap of a cell is seen, at the price of a returning LDS atomic.
__device__ __forceinline__ void win_add(int* p, float units, unsigned* ovf = nullptr) {
  const int add = cvt_round_half_up(units);
  if (ovf) {
    const int old = __hip_atomic_fetch_add(p, add, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const int sum = int(unsigned(old) + unsigned(add));
    if (((old ^ sum) & (add ^ sum)) < 0) atomicAdd(ovf, 1u);
  } else {
    lds_add(p, add);
  }
}
__device__ __forceinline__ void win_add(double* p, double v, unsigned* = nullptr) { lds_add(p, v); }
__device__ __forceinline__ void win_add(float* p, float v, unsigned* = nullptr) { lds_add(p, v); }  // (float cells: see geom_tile)
__device__ __forceinline__ float win_value(int c) { return float(c); }  // (in the cell's unit: the flush multiplies by 1 / scale)
__device__ __forceinline__ double win_value(double c) { return c; }
__device__ __forceinline__ float win_value(float c) { return c; }

// The WIDE window of a tile whose footprint is much taller than the window proper (incoherent depth: a direct global
// atomic per lane is the most expensive thing the scatter does -- moving the 17 % of an iid tile's taps that the window
// catches to direct atomics was measured at +160 us per launch).  Such a tile's tail finds next to none of its texels in
// the staged window either, so it does without staging and spends the three staging regions (kWideRows rows of WW cells
// each, one behind the parked gradients of every colour tile) on more window: row WH + 12 r + q (q < 12) lives at
// ext + r * ext_stride + q * WW, i.e. at its place in a contiguous window plus r * step + off cells.
struct WideWin {
  int rows;       // WH, or WH + 3 * kWideRows for a wide tile (uniform over the workgroup)
  int off, step;  // cells: (ext - window) - WH * WW, and ext_stride - kWideRows * WW
};
constexpr int kWideRows = 12;
template <int WH>
__device__ __forceinline__ int wide_adjust(const WideWin& ww, int ly) {  // only called when ww.rows > WH
  const int e = ly - WH;
  const int r = (e * 43) >> 9;  // e / 12 for 0 <= e < 36
  return e >= 0 ? ww.off + r * ww.step : 0;
}

// `scale`: what a value is multiplied with on its way into a cell -- the tile's 2^e for fixed-point cells (win_units_of),
// 1 for floating cells; the flush multiplies the cells by its reciprocal.  Taps that bypass the window add g itself.
template <typename T, typename Cell, int WW, int WH>
__device__ __forceinline__ void scatter_taps_window(Cell (*win)[WW], int wx0, int wy0,
                                                    T* __restrict__ gplane, const Sample<T>& s, T g,
                                                    const WideWin& ww = WideWin{WH, 0, 0}, T scale = win_scale0<Cell, T>(),
                                                    unsigned* ovf = nullptr) {
  if (g == T(0)) return;
  const int lx = s.xa - wx0, ly = s.ya - wy0;
  if (lx >= 0 && lx < WW - 1 && ly >= 0 && ly < ww.rows - 1 && win_fits(&win[0][0], g)) {
    // unpredicated: a cell of the block that is not a tap has weight 0, and adding 0 leaves it at the 0 the flush skips
    const T gu = g * scale;  // (a power of two for fixed-point cells: exact, applied once instead of per tap)
    Cell* rn = &win[0][0] + ly * WW + lx;
    Cell* rs = rn + WW;
    if (ww.rows > WH) { rn += wide_adjust<WH>(ww, ly); rs += wide_adjust<WH>(ww, ly + 1); }  // (uniform branch)
    win_add(rn, gu * s.wp[0], ovf);
    win_add(rn + 1, gu * s.wp[1], ovf);
    win_add(rs, gu * s.wp[2], ovf);
    win_add(rs + 1, gu * s.wp[3], ovf);
  } else {
    scatter_taps(gplane, s, g);
  }
}

// Only cells that received an in-image tap are non-zero, so every flushed cell is a valid pixel.
template <typename T, typename Cell, int WW, int WH>
__device__ __forceinline__ void flush_scatter_window(const Cell (*win)[WW], int wx0, int wy0,
                                                     T* __restrict__ gplane, int W, T unit = T(1) / win_scale0<Cell, T>()) {
  for (int i = threadIdx.x; i < WW * WH; i += kThreads) {
    const int ly = i / WW, lx = i - ly * WW;
    const Cell v = win[ly][lx];
    if (v != Cell(0)) atomicAdd(gplane + unsigned(wy0 + ly) * unsigned(W) + unsigned(wx0 + lx), T(win_value(v)) * unit);
  }
}

// The same flush restricted to the cells [cx0, cx1] x [cy0, cy1] of the window (inclusive, window coordinates;
// clamped into it) that can be non-zero: the bounding box of the block's taps is usually half the window, so
// half as many atomic instructions are issued, each with (nearly) all of its lanes active.
#ifndef SCSFM_FLUSH_STEP  // tuning knob: 0 = the flush of rounds 1-3 (a division per cell)
#define SCSFM_FLUSH_STEP 1
#endif
template <typename T>
__device__ __forceinline__ void atomic_add_at(T* __restrict__ base, unsigned byte_off, T v) {
  atomicAdd(reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off), v);
}
template <typename T, typename Cell, int WW, int WH, int NT = kThreads>
__device__ __forceinline__ void flush_scatter_region(const Cell (*win)[WW], int wx0, int wy0, int cx0,
                                                     int cy0, int cx1, int cy1, T* __restrict__ gplane, int W,
                                                     const WideWin& ww = WideWin{WH, 0, 0}, T inv = T(1) / win_scale0<Cell, T>()) {
  cx0 = cx0 < 0 ? 0 : cx0; cy0 = cy0 < 0 ? 0 : cy0;
  cx1 = cx1 > WW - 1 ? WW - 1 : cx1; cy1 = cy1 > ww.rows - 1 ? ww.rows - 1 : cy1;
  const int w = cx1 - cx0 + 1, h = cy1 - cy0 + 1;
  if (w <= 0 || h <= 0) return;
  const float iw = 1.0f / float(w);
#if SCSFM_FLUSH_STEP
  // Cell i of the region (row-major), i = thread, thread + NT, ...: (row, column) advance by the uniform (NT / w, NT % w)
  // with a carry instead of being divided out per cell, two cells' LDS reads are in flight together, and the global
  // address is a 32-bit byte offset from the uniform plane base (a scatter plane is far below 4 GiB).
  const int n = w * h;
  const int dq = int((float(NT) + 0.5f) * iw), dr = NT - dq * w;  // NT / w, NT % w: exact for 1 <= w <= NT
  int i = threadIdx.x;
  int ry = int((float(i) + 0.5f) * iw);  // i / w, exact for the few thousand cells of a window
  int rx = i - ry * w;
  const bool wide = ww.rows > WH;
  auto cell_of = [&](int y, int x) { const int ly = cy0 + y, lx = cx0 + x; return ly * WW + lx + (wide ? wide_adjust<WH>(ww, ly) : 0); };
  auto dest_of = [&](int y, int x) { return (unsigned(wy0 + cy0 + y) * unsigned(W) + unsigned(wx0 + cx0 + x)) * unsigned(sizeof(T)); };
  for (; i < n; i += 2 * NT) {
    int ry2 = ry + dq, rx2 = rx + dr;
    if (rx2 >= w) { rx2 -= w; ++ry2; }
    const bool second = i + NT < n;
    const Cell v0 = (&win[0][0])[cell_of(ry, rx)];
    const Cell v1 = second ? (&win[0][0])[cell_of(ry2, rx2)] : Cell(0);
    if (v0 != Cell(0)) atomic_add_at(gplane, dest_of(ry, rx), T(win_value(v0)) * inv);
    if (v1 != Cell(0)) atomic_add_at(gplane, dest_of(ry2, rx2), T(win_value(v1)) * inv);
    ry = ry2 + dq; rx = rx2 + dr;
    if (rx >= w) { rx -= w; ++ry; }
  }
#else
  for (int i = threadIdx.x; i < w * h; i += NT) {
    const int ry = int((float(i) + 0.5f) * iw);  // i / w, exact for the few thousand cells of a window
    const int ly = cy0 + ry, lx = cx0 + (i - ry * w);
    const Cell v = (&win[0][0])[ly * WW + lx + (ww.rows > WH ? wide_adjust<WH>(ww, ly) : 0)];
    if (v != Cell(0)) atomicAdd(gplane + unsigned(wy0 + ly) * unsigned(W) + unsigned(wx0 + lx), T(win_value(v)) * inv);
  }
#endif
}

// Where the window of a tile sits: centred on where the tile's centre pixel (ax, ay) lands in the reference
// view.  Every thread evaluates it (one broadcast load + one projection): handing it over from a single
// thread would put that thread's dependent load in front of a barrier for the whole block.
template <typename T, int WW, int WH, typename Map>
__device__ __forceinline__ void window_origin(const BatchConsts<T>& bc, int ax, int ay, const Map& tgt_depth,
                                              int H, int W, unsigned flags, int& wx0, int& wy0) {
  ax = t_clampi(ax, 0, W - 1); ay = t_clampi(ay, 0, H - 1);
  const unsigned off = (unsigned(ay) * unsigned(W) + unsigned(ax)) * unsigned(sizeof(T));
  const Sample<T> sc = project_pixel(bc, ax, ay, tgt_depth.at(ax, ay, off), H, W, flags);
  wx0 = sc.xa - WW / 2;
  wy0 = sc.ya - WH / 2;
}

// Backward of one target pixel through the bilinear sampler and the camera geometry (the per-pixel body of
// the geometry pass): gathers the taps of the three colours and of the reference depth, folds
//   dL/d(ix, iy) = sum_c gI_c * dI_w,c/d(ix, iy) + dL/dD_p * dD_p/d(ix, iy),
// scatters dL/dD_p over the reference-depth taps (LDS window), accumulates dL/d(A|c) into acc[12] and
// returns dL/d tgt_depth(p).  g_dd = dL/d diff_depth(p).
// (Measured alternatives, both slower: a software pipeline that requests pixel r + 1's taps before pixel r is
// consumed, and finishing the whole strip's arithmetic before a separate scatter loop over compact records.)
// The per-pixel body in two stages: geom_fetch projects and issues the 8 tap loads, geom_consume does the arithmetic.
// (A software pipeline over a thread's strip -- row k + 1's gathers in flight while row k is consumed, 143 VGPRs, no
// spills -- was measured again in round 2: 351 us instead of 343 us per launch.  The tail is bound by the vector
// instructions it issues next to the other resident workgroups, not by the latency of its L2-resident gathers.)
template <typename T>
struct GeomTaps {
  Sample<T> s;
  TapRows<T> tc[3], td;
};
template <typename T, typename Map>
__device__ __forceinline__ GeomTaps<T> geom_fetch(const BatchConsts<T>& bc, int px, int py, T d,
                                                  const T* __restrict__ ref_img, const Map& ref_depth, unsigned plane,
                                                  int H, int W, unsigned flags) {
  GeomTaps<T> f;
  f.s = project_pixel(bc, px, py, d, H, W, flags);
#pragma unroll
  for (int c = 0; c < 3; ++c) f.tc[c] = load_tap_rows(ref_img + c * plane, f.s);
  f.td = ref_depth.taps(f.s);
  return f;
}
// The reference view's texels a tile's tail samples, staged in LDS: four planes (three colours, depth) of a
// kStageW x kStageH window at (x0, y0), filled with coalesced row loads.  A 2 x 2 block inside it is read from LDS
// (two 2-dword reads per plane); any other block is gathered from global memory as before.  On gfx950 a per-lane
// gather instruction costs ~16 cycles of the CU's texture addresser whatever its width (tools/ubench/gathers.hip:
// 59 ns per wave for the 8 gathers of one sample), a coalesced dword row 4, an LDS read 4.
constexpr int kStageW = 72, kStageH = kTileH + 2;
template <typename T>
struct StagedTaps {
  const T* colour;  // LDS: plane c of the colours starts at colour + c * stride
  const T* depth;   // LDS
  int stride;
  int x0, y0;
};
// kRows: rows of the staged window; kDepth: the depth plane is staged too (else it is gathered).
template <int kRows, bool kDepth, typename T, typename Map>
__device__ __forceinline__ GeomTaps<T> geom_fetch(const BatchConsts<T>& bc, int px, int py, T d,
                                                  const Planes3<T>& ref_img, const Map& ref_depth,
                                                  int H, int W, unsigned flags, const StagedTaps<T>& st) {
  GeomTaps<T> f;
  f.s = project_pixel(bc, px, py, d, H, W, flags);
  const int lx = f.s.xa - st.x0, ly = f.s.ya - st.y0;
  if (!kDepth) f.td = ref_depth.taps(f.s);
  if (unsigned(lx) <= unsigned(kStageW - 2) && unsigned(ly) <= unsigned(kRows - 2)) {
    const int o = ly * kStageW + lx;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const T* p = st.colour + c * st.stride + o;
      f.tc[c].n.a = lds_ld(p); f.tc[c].n.b = lds_ld(p + 1); f.tc[c].s.a = lds_ld(p + kStageW); f.tc[c].s.b = lds_ld(p + kStageW + 1);
    }
    if (kDepth) {
      const T* q = st.depth + o;
      f.td.n.a = lds_ld(q); f.td.n.b = lds_ld(q + 1); f.td.s.a = lds_ld(q + kStageW); f.td.s.b = lds_ld(q + kStageW + 1);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) f.tc[c] = load_tap_rows(ref_img, c, f.s);
    if (kDepth) f.td = ref_depth.taps(f.s);
  }
  return f;
}

template <typename T, typename Cell, int WW, int WH>
__device__ __forceinline__ T geom_consume(const BatchConsts<T>& bc, const GeomTaps<T>& f, int px, int py, T d, const T (&gI)[3], T g_dd,
                                          int H, int W, unsigned flags, Cell (*win)[WW], int wx0, int wy0,
                                          T* __restrict__ scatter_plane, T* acc, const WideWin& ww = WideWin{WH, 0, 0},
                                          T scale = win_scale0<Cell, T>(), unsigned* ovf = nullptr) {
  const Sample<T>& s = f.s;
  const TapRows<T>(&tc)[3] = f.tc;
  const TapRows<T>& td = f.td;
  const T Dp = bilerp_rows(td, s);
  const T diff = s.Z - Dp, sum = s.Z + Dp;
  const T isum = t_rcp(sum);
  const T raw = t_abs(diff) * isum;
  T gZ = T(0), gDp = T(0);
  if (raw >= T(0) && raw <= T(1)) {  // diff_depth = clamp(|Z - Dp| / (Z + Dp), 0, 1), loss_functions.py:101
    const T sgn = t_sgn(diff), i2 = isum * isum;
    gZ = g_dd * (sgn * T(2) * Dp * i2);
    gDp = -g_dd * (sgn * T(2) * s.Z * i2);
  }
  // d (sum over the four planes of g_plane * sampled value) / d (ix, iy): contract over the planes first (g = dL/d
  // warped colour c, dL/dD_p), then apply the block's weights and slopes once -- 28 operations instead of 56
  TapRows<T> t;
  t.n.a = gI[0] * tc[0].n.a + gI[1] * tc[1].n.a + gI[2] * tc[2].n.a + gDp * td.n.a;
  t.n.b = gI[0] * tc[0].n.b + gI[1] * tc[1].n.b + gI[2] * tc[2].n.b + gDp * td.n.b;
  t.s.a = gI[0] * tc[0].s.a + gI[1] * tc[1].s.a + gI[2] * tc[2].s.a + gDp * td.s.a;
  t.s.b = gI[0] * tc[0].s.b + gI[1] * tc[1].s.b + gI[2] * tc[2].s.b + gDp * td.s.b;
  T gix, giy;
  tap_rows_grad(t, s, gix, giy);
  if (!(flags & SCSFM_DEBUG_X1)) scatter_taps_window<T, Cell, WW, WH>(win, wx0, wy0, scatter_plane, s, gDp, ww, scale, ovf);
  return pixel_geometry_bwd(bc, s, px, py, d, gix, giy, gZ, H, W, acc);
}
template <typename T, typename Cell, int WW, int WH, typename Map>
__device__ __forceinline__ T geom_pixel(const BatchConsts<T>& bc, int px, int py, T d, const T (&gI)[3], T g_dd,
                                        const T* __restrict__ ref_img, const Map& ref_depth,
                                        unsigned plane, int H, int W, unsigned flags, Cell (*win)[WW], int wx0, int wy0,
                                        T* __restrict__ scatter_plane, T* acc, T scale = win_scale0<Cell, T>(),
                                        unsigned* ovf = nullptr) {
  const GeomTaps<T> f = geom_fetch(bc, px, py, d, ref_img, ref_depth, plane, H, W, flags);
  return geom_consume<T, Cell, WW, WH>(bc, f, px, py, d, gI, g_dd, H, W, flags, win, wx0, wy0, scatter_plane, acc,
                                       WideWin{WH, 0, 0}, scale, ovf);
}

}  // namespace scsfm
