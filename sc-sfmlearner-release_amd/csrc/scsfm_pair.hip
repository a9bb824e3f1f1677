// compute_pairwise_loss (loss_functions.py:95-119) for one (tgt, ref) pair-direction, fused:
// back-projection + SE(3) + projection + bilinear warp of image and depth (inverse_warp.py:230-269),
// valid / auto mask, clamped L1, SSIM (loss_functions.py:11-42), depth inconsistency, the
// self-discovered weight mask and the two masked means (loss_functions.py:123-129).
//
// Forward, no backward to follow (validation):
//            pair_fwd_kernel  -> per-block partial sums {sum photo*m, sum geom*m, sum m}
//            pair_finalize_kernel -> the two gated losses + the backward coefficients (on device;
//            the reference's `if mask.sum() > 10000` host sync disappears)
// Forward of a training step (speculative: the upstream gradients are assumed to stand in the ratio of the
// loss weights):
//            pair_fwd_spec_kernel -> the same partial sums AND the whole backward up to the scalar the
//            reduction supplies: tiled SSIM backward, then the geometry tail writes this pair's dense
//            dL/d tgt_depth plane, scatters its dL/d ref_depth plane (LDS window + fp32 atomics) and
//            leaves per-block partials of dL/d(A|c)
// Backward : pair_bwd_photo_kernel + pair_bwd_geom_kernel are the same two passes with the true coefficients;
//            they run only when the speculation does not hold (a guard at their top otherwise);
//            pairs_pose_reduce_kernel finishes dL/d pose, pairs_combine_kernel scales the private planes
//            and stores their sums into the callers' gradient buffers.
//
// Tiling (gfx950): 256 threads = 4 waves; a wave spans 64 consecutive pixels of a row (256 B
// coalesced rows), each thread owns a vertical strip of the 64 x TH tile, so the SSIM window sums
// slide down the strip and every LDS read is a conflict-free row access.  The warped image and
// the target image of the tile + 1-pixel ring live in LDS; ring positions outside the image hold
// the reflected pixel (ReflectionPad2d(1)).
#include <cstdlib>

#include "scsfm_geom.h"
#include "scsfm_ssim.h"

namespace scsfm {

// Device pointers of one pair-direction.  Kernels receive up to kMaxPairs of them by value (kernel
// argument segment, read with scalar loads) and find theirs from blockIdx.z = pair * B + b, so that all
// pair-directions of a training step (2 per reference frame and scale) run as ONE launch per stage:
// 4x the blocks per launch (no half-empty last wave of blocks) and 4x fewer launches.
template <typename T>
struct PairArgs {
  const T* tgt_img; const T* ref_img; const T* tgt_depth; const T* ref_depth; const T* pose;
  BatchConsts<T>* consts; double* sums; double* partials; double* gPp;
  T* out; T* gbuf; T* g_ref_depth; T* g_pose;
  int ds;  // log2 of the down-scale of BOTH depth maps of this pair (scsfm_pair_desc::depth_shift)
  // scsfm_pair_desc::smooth_ws (speculative forward only): this pair also evaluates get_smooth_loss of its TARGET frame
  double* sm_partials;  // nullptr = no; else double[B][tiles per image][3] in the pair's workspace: {sum D, Sx, Sy} per tile
  T* sm_edge;           // nullptr or the frame's edge plane [B,H,W]
  double* sm_img;       // the frame's {mean + 1e-7, L} per image (the start of its smooth workspace)
  T* sm_out;            // nullptr or 1 element: the frame's loss
};
constexpr int kMaxPairs = 8;
// Planes of a pair's gbuf (each B x H x W): what the tiled pass hands to the geometry pass, and the geometry
// pass's private dense output.
constexpr int kPlaneGI = 0;     // planes 0..2: dL/d(warped colour c)
constexpr int kPlaneGdd = 3;    // dL/d diff_depth
constexpr int kPlaneDense = 4;    // dL/d tgt_depth of this pair-direction (dense)
constexpr int kPlaneScatter = 5;  // dL/d ref_depth of this pair-direction (scattered; zeroed before the scatter)
constexpr int kNumPlanes = 6;
// Workgroups of the backward's two passes (persistent: they walk the tiles).  A multiple of the 8 XCDs; 4 per CU.
constexpr int kPersistentGrid = 1024;
template <typename T>
struct PairBatch {
  PairArgs<T> p[kMaxPairs];
};

// The streaming inputs of one pixel (already reflected into the image): depth, target colours and -- for
// the auto-mask -- the un-warped reference colours.  They are loaded for a thread's whole strip before
// anything else, so that a single memory round trip precedes the dependent gathers.
template <typename T, typename Map>
__device__ __forceinline__ void load_pixel(int u, int v, int W, unsigned plane, const T* __restrict__ tgt_img,
                                           const T* __restrict__ ref_img, const Map& tgt_depth,
                                           bool with_ref, T& depth, T (&tgt)[3], T (&ref)[3]) {
  const unsigned off = (unsigned(v) * unsigned(W) + unsigned(u)) * unsigned(sizeof(T));
  depth = tgt_depth.at(u, v, off);
#pragma unroll
  for (int c = 0; c < 3; ++c) tgt[c] = ld_at(tgt_img + c * plane, off);
#pragma unroll
  for (int c = 0; c < 3; ++c) ref[c] = T(0);
  if (with_ref) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ref[c] = ld_at(ref_img + c * plane, off);
  }
}

// ... through one buffer resource per image (Planes3, scsfm_geom.h)
template <typename T, typename Map>
__device__ __forceinline__ void load_pixel(int u, int v, int W, const Planes3<T>& tgt_img, const Planes3<T>& ref_img,
                                           const Map& tgt_depth, bool with_ref, T& depth, T (&tgt)[3], T (&ref)[3]) {
  const unsigned off = (unsigned(v) * unsigned(W) + unsigned(u)) * unsigned(sizeof(T));
  depth = tgt_depth.at(u, v, off);
#pragma unroll
  for (int c = 0; c < 3; ++c) tgt[c] = ld_plane(tgt_img, c, off);
#pragma unroll
  for (int c = 0; c < 3; ++c) ref[c] = T(0);
  if (with_ref) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ref[c] = ld_plane(ref_img, c, off);
  }
}

// Warp one pixel: the (target, warped) colour pairs.
template <typename T>
__device__ __forceinline__ Sample<T> warp_colours(const BatchConsts<T>& bc, int u, int v, T depth, const T (&tgt)[3],
                                                  int H, int W, unsigned flags, const T* __restrict__ ref_img,
                                                  typename Vec2<T>::type* xy) {
  const unsigned plane = unsigned(H) * unsigned(W);
  const Sample<T> s = project_pixel(bc, u, v, depth, H, W, flags);
#pragma unroll
  for (int c = 0; c < 3; ++c) xy[c] = make2(tgt[c], bilerp_rows(load_tap_rows(ref_img + c * plane, s), s));
  return s;
}

// "finalize blocks done" counter of a launch: a word of the first pair's sums block (sums[12..15] are spare).
template <typename T>
__device__ __forceinline__ unsigned* finalize_counter(const PairBatch<T>& pb) {
  return reinterpret_cast<unsigned*>(pb.p[0].sums + 12);
}

// "cells of the scatter window that wrapped" counter of a pair (SCSFM_DEBUG_CHECK_WINDOW launches): another spare word
// of its sums block.  Zeroed with the per-image constants, reported in out[7] by the finalize kernel.
template <typename T>
__device__ __forceinline__ unsigned* window_overflow_counter(const PairArgs<T>& pa) {
  return reinterpret_cast<unsigned*>(pa.sums + 13);
}

// prep_kernel over every (pair, batch element).
template <typename T>
__global__ void pairs_prep_kernel(PairBatch<T> pb, int n, int B, const T* __restrict__ K) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n * B) return;
  if (i == 0) *finalize_counter(pb) = 0u;
  const int pair = i / B, b = i - pair * B;
  if (b == 0) *window_overflow_counter(pb.p[pair]) = 0u;
  prep_one(b, pb.p[pair].pose, K, pb.p[pair].consts);
}

// Mask of an owned pixel: valid (inverse_warp.py:264), optionally AND auto-mask
// (loss_functions.py:103-105: mean_c clamped |It - Iw| < mean_c |It - Ir|, same pixel, un-warped ref).
// Both means share the divisor 3, so the sums are compared.
template <typename T>
__device__ __forceinline__ T pixel_mask(const Sample<T>& s, bool with_auto, const typename Vec2<T>::type* xy,
                                        const T (&ref)[3]) {
  T m = s.valid ? T(1) : T(0);
  if (with_auto) {
    const T ident = t_abs(xy[0][0] - ref[0]) + t_abs(xy[1][0] - ref[1]) + t_abs(xy[2][0] - ref[2]);
    const T warped = clamp01(t_abs(xy[0][0] - xy[0][1])) + clamp01(t_abs(xy[1][0] - xy[1][1])) +
                     clamp01(t_abs(xy[2][0] - xy[2][1]));
    m = (warped < ident) ? m : T(0);
  }
  return m;
}

// ==========================================================================================
// Forward
// ==========================================================================================
#ifndef SCSFM_FWD_BLOCKS  // tuning knobs of the plain forward: workgroups per CU it is compiled for, pixels whose gathers
#define SCSFM_FWD_BLOCKS 4  // are in flight together
#endif
#ifndef SCSFM_FWD_GROUP
#define SCSFM_FWD_GROUP 2
#endif
template <typename T, bool kSsim, bool kScaled>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? SCSFM_FWD_BLOCKS : 1) void pair_fwd_kernel(PairBatch<T> pb, int B, int H, int W, unsigned flags) {
  const BlockId blk = xcd_block_id();
  const int pair = blk.z / B, b = blk.z - pair * B;
  const PairArgs<T>& pa = pb.p[pair];
  const T* __restrict__ tgt_img = pa.tgt_img;
  const T* __restrict__ ref_img = pa.ref_img;
  const BatchConsts<T>* __restrict__ consts = pa.consts;
  double* __restrict__ partials = pa.partials;
  typedef typename Vec2<T>::type V2;
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ V2 sXY[kSsim ? 3 : 1][kSsim ? TH + 2 : 1][kSsim ? kHaloW : 1];  // (target, warped) + 1-pixel ring
  __shared__ double red[3 * (kThreads / kWave)];

  // (the wave index as a scalar for row arithmetic and as a vector register for LDS addresses: scsfm_spec_tile.h)
  const int col = threadIdx.x & (kWave - 1), strip = __builtin_amdgcn_readfirstlane(int(threadIdx.x) / kWave);
  const int lrow = (int(threadIdx.x) / kWave) * STRIP;
  const int tx0 = blk.x * kTileW, ty0 = blk.y * TH;
  const bool with_mask = (flags & SCSFM_WITH_MASK) != 0, with_auto = (flags & SCSFM_WITH_AUTO_MASK) != 0;
  const BatchConsts<T> bc = consts[b];
  const unsigned plane = unsigned(H) * unsigned(W);
  tgt_img += (size_t)b * 3 * plane;
  ref_img += (size_t)b * 3 * plane;
  const Planes3<T> tgtP = planes3(tgt_img, plane), refP = planes3(ref_img, plane);
  const DepthMap<T, kScaled> tgt_depth = depth_map<kScaled>(pa.tgt_depth, b, H, W, pa.ds);
  const DepthMap<T, kScaled> ref_depth = depth_map<kScaled>(pa.ref_depth, b, H, W, pa.ds);

  T m[STRIP], dd[STRIP], l1sum[STRIP];
  // ---- phase 0: every streaming load of the strip and of this thread's ring pixel ---------------
  const int gx = tx0 + col, u = reflect_index(gx, W);
  T in_d[STRIP], in_t[STRIP][3], in_r[STRIP][3], rin_d = T(0), rin_t[3] = {T(0), T(0), T(0)}, rin_r[3];
#pragma unroll
  for (int k = 0; k < STRIP; ++k)
    load_pixel(u, reflect_index(ty0 + strip * STRIP + k, H), W, tgtP, refP, tgt_depth, with_auto, in_d[k], in_t[k], in_r[k]);
  const bool has_ring = kSsim && threadIdx.x < 2 * kHaloW + 2 * TH;
  int ru = 0, rv = 0, rhy = 0, rhx = 0;
  if (has_ring) {
    ring_pos<TH>(threadIdx.x, rhy, rhx);
    ru = reflect_index(tx0 + rhx - 1, W); rv = reflect_index(ty0 + rhy - 1, H);
    load_pixel(ru, rv, W, tgtP, refP, tgt_depth, false, rin_d, rin_t, rin_r);
  }
  // ---- phase 1a: the pixels this thread owns -------------------------------------------------
  // (two pixels' gathers in flight together, as in the speculative forward)
  constexpr int G = STRIP % SCSFM_FWD_GROUP == 0 ? SCSFM_FWD_GROUP : 1;
#pragma unroll
  for (int k0 = 0; k0 < STRIP; k0 += G) {
    Sample<T> sm[G];
    TapRows<T> tc[G][3], td[G];
#pragma unroll
    for (int j = 0; j < G; ++j) {
      sm[j] = project_pixel(bc, u, reflect_index(ty0 + strip * STRIP + k0 + j, H), in_d[k0 + j], H, W, flags);
#pragma unroll
      for (int c = 0; c < 3; ++c) tc[j][c] = load_tap_rows(refP, c, sm[j]);
      td[j] = ref_depth.taps(sm[j]);
    }
#pragma unroll
    for (int j = 0; j < G; ++j) {
      const int k = k0 + j, ly = strip * STRIP + k, gy = ty0 + ly;
      const bool inimg = gx < W && gy < H;
      const Sample<T>& s = sm[j];
      V2 xy[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) xy[c] = make2(in_t[k][c], bilerp_rows(tc[j][c], s));
      if (kSsim) {
#pragma unroll
        for (int c = 0; c < 3; ++c) sXY[c][lrow + k + 1][col + 1] = xy[c];
      }
      l1sum[k] = clamp01(t_abs(xy[0][0] - xy[0][1])) + clamp01(t_abs(xy[1][0] - xy[1][1])) +
                 clamp01(t_abs(xy[2][0] - xy[2][1]));  // loss_functions.py:99, summed over colours
      const T Dp = bilerp_rows(td[j], s);
      dd[k] = clamp01(t_abs(s.Z - Dp) * t_rcp(s.Z + Dp));  // loss_functions.py:101
      m[k] = inimg ? pixel_mask(s, with_auto, xy, in_r[k]) : T(0);
    }
  }
  // ---- phase 1b: the 1-pixel ring (SSIM windows of the tile's border pixels) ------------------
  if (kSsim) {
    if (has_ring) {
      const Sample<T> rs = project_pixel(bc, ru, rv, rin_d, H, W, flags);
#pragma unroll
      for (int c = 0; c < 3; ++c) sXY[c][rhy][rhx] = make2(rin_t[c], bilerp_rows(load_tap_rows(refP, c, rs), rs));
    }
    __syncthreads();
  }
  // ---- phase 2: SSIM down the strip, blend, weight, accumulate -------------------------------
  T photo[STRIP];
#pragma unroll
  for (int k = 0; k < STRIP; ++k) photo[k] = kSsim ? T(0.15) * l1sum[k] : l1sum[k];  // loss_functions.py:109
  if constexpr (kSsim) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      WinSums<T> ws[STRIP];
      V2 centre[STRIP];
      strip_window_sums<T, STRIP>(sXY[c], lrow, col, ws, centre);
#pragma unroll
      for (int k = 0; k < STRIP; ++k) photo[k] += T(0.85) * clamp01(ssim_stats(ws[k]).raw);
    }
  }
  T acc_p = T(0), acc_g = T(0), acc_m = T(0);
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const T w = with_mask ? (T(1) - dd[k]) : T(1);  // loss_functions.py:111-113
    acc_p += photo[k] * w * m[k];
    acc_g += dd[k] * m[k];
    acc_m += m[k];
  }
  T v[3] = {acc_p, acc_g, acc_m};
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    double* o = partials + 3 * ((size_t)(b * gridDim.y + blk.y) * gridDim.x + blk.x);
    o[0] = double(v[0]); o[1] = double(v[1]); o[2] = double(v[2]);
  }
}

// The gates and divisions of mean_on_mask (loss_functions.py:123-129) on the three sums; also
// publishes the coefficients the backward multiplies the upstream gradients with.
// out[8] = {photo, geom, S_photo, S_geom, S_mask, 0, 0, wrapped window cells (debug launches; else 0)}
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
// `total` (optional): the sums over all pair-directions of a call of the two losses -- what
// compute_photo_and_geometry_loss returns -- finished by whichever block comes last, in pair order.
// The step's objective formed by the finalize launch (scsfm_pairs_fwd_step): out[4] = {w1 photo + w2 smooth + w3 geometry,
// photo, smooth, geometry}; out == nullptr: not wanted.
template <typename T>
struct StepTotal {
  T* smooth_total;  // nullptr or 1 element: the sum of the frames' smooth losses (scsfm_pair_desc::smooth_total)
  T* out;
  T w1, w2, w3;
};
// Grid: one workgroup per pair-direction for the pair losses and -- when a pair of the launch carries its target frame's
// smooth loss -- a second row of workgroups (blockIdx.x >= npairs) for the frames' records, so that the two chains of L2
// round trips run side by side instead of one behind the other (12-14 us -> what the longer of the two takes); whichever
// workgroup of either role arrives last adds up the call's totals.
template <typename T>
__global__ __launch_bounds__(kThreads) void pair_finalize_kernel(PairBatch<T> pb, int npairs, int nblocks, int nblk_img, double spec,
                                                                 double w_photo, double w_geom, T* total, int first,
                                                                 const double* __restrict__ hint, StepTotal<T> st, int H, int W) {
  __shared__ double red[3 * (kThreads / kWave)];
  const bool smooth_role = (int)blockIdx.x >= npairs;  // (uniform per workgroup)
  const PairArgs<T>& pa = pb.p[smooth_role ? blockIdx.x - npairs : blockIdx.x];
  const double* __restrict__ partials = pa.partials;
  double* __restrict__ sums = pa.sums;
  T* __restrict__ out = pa.out;
  double v[3] = {0, 0, 0};
  if (!smooth_role) {
    // four records per thread in flight (the loop is a chain of L2 round trips: 13 of them for the 3192 records of a
    // configs[1] launch with one record per iteration, 4 this way); same order of additions per thread on every run
    constexpr int U = 4;
    for (int i0 = threadIdx.x; i0 < nblocks; i0 += U * kThreads) {
      double q[U][3];
#pragma unroll
      for (int j = 0; j < U; ++j) {
        const int i = i0 + j * kThreads;
        const bool ok = i < nblocks;
        const double* r = partials + 3 * (ok ? i : 0);
        q[j][0] = r[0]; q[j][1] = r[1]; q[j][2] = r[2];
        if (!ok) { q[j][0] = 0.0; q[j][1] = 0.0; q[j][2] = 0.0; }
      }
#pragma unroll
      for (int j = 0; j < U; ++j) { v[0] += q[j][0]; v[1] += q[j][1]; v[2] += q[j][2]; }
    }
    block_sum<3>(v, red);
  } else if (pa.sm_partials != nullptr) {
    // the target frame's smooth loss from the tiles' {sum D, Sx, Sy} records (smooth_finalize_kernel's arithmetic: each
    // wave reduces whole images with shuffles, in a fixed order, then the waves' contributions meet in LDS)
    __shared__ double sm_red[kThreads / kWave];
    const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave, B = nblocks / nblk_img;
    const int nrec = nblk_img;  // one record per tile of an image
    const double cnt_x = (double)B * H * (W - 1), cnt_y = (double)B * (H - 1) * W;
    double loss = 0.0;
    // Three images of a wave and five records of each per lane in flight: the 12 x 266 records of a configs[1] launch are
    // ONE L2 round trip per wave (image after image, eight records at a time, this loop cost the launch 6 us).  The order
    // of the additions is fixed: records in steps of 64 per lane, the wave butterfly, images in ascending order.
    constexpr int IM = 3, U5 = 5, NW = kThreads / kWave;
    for (int b0 = wave; b0 < B; b0 += IM * NW) {
      double v[IM][3];
#pragma unroll
      for (int m = 0; m < IM; ++m) { v[m][0] = 0.0; v[m][1] = 0.0; v[m][2] = 0.0; }
      for (int i0 = lane; i0 < nrec; i0 += U5 * kWave) {
        double q[IM][U5][3];
#pragma unroll
        for (int m = 0; m < IM; ++m) {
          const int bm = b0 + m * NW;
#pragma unroll
          for (int j = 0; j < U5; ++j) {
            const int i = i0 + j * kWave;
            const bool ok = i < nrec && bm < B;
            const double* r = pa.sm_partials + 3 * (ok ? (size_t)bm * nrec + i : 0);
            q[m][j][0] = r[0]; q[m][j][1] = r[1]; q[m][j][2] = r[2];
            if (!ok) { q[m][j][0] = 0.0; q[m][j][1] = 0.0; q[m][j][2] = 0.0; }
          }
        }
#pragma unroll
        for (int m = 0; m < IM; ++m)
#pragma unroll
          for (int j = 0; j < U5; ++j) { v[m][0] += q[m][j][0]; v[m][1] += q[m][j][1]; v[m][2] += q[m][j][2]; }
      }
#pragma unroll
      for (int m = 0; m < IM; ++m) {
        const int bm = b0 + m * NW;
        const double v0 = wave_sum(v[m][0]), v1 = wave_sum(v[m][1]), v2 = wave_sum(v[m][2]);
        if (bm < B) {  // (uniform)
          const double den = v0 / ((double)H * W) + 1e-7;  // mean_HW(D) + 1e-7, loss_functions.py:139-140
          const double L = v1 / cnt_x + v2 / cnt_y;
          if (lane == 0) { pa.sm_img[2 * bm] = den; pa.sm_img[2 * bm + 1] = L; }
          loss += L / den;
        }
      }
    }
    if (lane == 0) sm_red[wave] = loss;
    __syncthreads();
    if (threadIdx.x == 0 && pa.sm_out) {
      double t = 0;
      for (int w = 0; w < kThreads / kWave; ++w) t += sm_red[w];
      pa.sm_out[0] = T(t);
    }
  }
  if (threadIdx.x == 0) {
    if (!smooth_role) {
      publish_losses(v[0], v[1], v[2], sums, out);
      out[7] = T(*window_overflow_counter(pa));  // (0 unless the forward was launched with SCSFM_DEBUG_CHECK_WINDOW and a cell wrapped)
      if (hint && spec != 0.0) {  // the weights the speculative forward read from the device (scsfm_pair_desc::hint)
        w_photo = hint[0]; w_geom = hint[1];
        if (w_photo == 0.0) spec = 0.0;  // nothing to factor out: the backward runs its own passes
      }
      sums[8] = spec; sums[9] = w_photo; sums[10] = w_geom;
      sums[11] = double(nblk_img);  // partial records per image the forward left (the pose reduction of the backward reads them)
    }
    if (total) {
      __threadfence();
      if (atomicAdd(finalize_counter(pb), 1u) == gridDim.x - 1) {
        __threadfence();
        T photo = T(0), geom = T(0);
        for (int i = 0; i < npairs; ++i) {
          const volatile T* o = pb.p[i].out;
          photo += o[0]; geom += o[1];
        }
        total[0] = first ? photo : total[0] + photo;
        total[1] = first ? geom : total[1] + geom;
        if (st.smooth_total) {  // the frames' smooth losses, in descriptor order
          T smooth = T(0);
          for (int i = 0; i < npairs; ++i)
            if (pb.p[i].sm_partials && pb.p[i].sm_out) smooth += *const_cast<const volatile T*>(pb.p[i].sm_out);
          st.smooth_total[0] = first ? smooth : st.smooth_total[0] + smooth;
          if (st.out) {
            st.out[0] = st.w1 * total[0] + st.w2 * st.smooth_total[0] + st.w3 * total[1];
            st.out[1] = total[0]; st.out[2] = st.smooth_total[0]; st.out[3] = total[1];
          }
        }
      }
    }
  }
}

// Data-parallel "exact" mode: the caller all-reduces out[2..4] (the three raw sums)
// over the ranks, then re-runs the gate / division on the global sums so that both the loss and the
// backward coefficients are those of the concatenated batch.
template <typename T>
__global__ void pair_refinalize_kernel(double* __restrict__ sums, T* __restrict__ out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) publish_losses(double(out[2]), double(out[3]), double(out[4]), sums, out);
}

// ==========================================================================================
// Backward, pass A (tiled): dL/d(warped colours) and dL/d(diff_depth) per pixel.
//
// Recomputes the warp inside the tile (nothing but 3 sums is kept from the forward), runs the SSIM
// backward -- forward statistics at every pixel of the 64 x TH domain, then the transpose of
// (reflect-pad + box) as a separable 3x3 gather -- and writes four planes for the 62 x (TH-2) interior:
// gbuf[c] = dL/dI_w,c (c = 0..2), gbuf[3] = dL/d diff_depth, which pass B consumes.  These two passes only run
// when the speculative forward's results do not stand (spec_valid) or no speculative forward ran.
// ==========================================================================================
//
// The SPECULATIVE FORWARD (scsfm_spec_tile.h) runs the forward with unit photo coefficient and the geometry / photo
// coefficient ratio r = 3 w_geom / w_photo the caller expects the upstream gradients to have (the loss weights are
// constants of a training run, train.py:268).  It produces the three sums of the forward AND carries on through
// both passes of the backward for the pixels it owns: the pair's dense / scatter planes and pose partials, all up
// to the common factor a = g_photo / (3 S_m), which is only known after the reduction and is applied when the
// planes are combined.  If the upstream gradients turn out different (spec_valid), passes A and B run normally.
template <typename T>
__device__ __forceinline__ bool spec_valid(const double* __restrict__ sums, const T* __restrict__ g_photo,
                                           const T* __restrict__ g_geom) {
  if (sums[8] == 0.0) return false;          // the forward was not speculative
  if (sums[5] == 0.0) return true;           // photo gate closed: every gradient is zero
  const double gate_g = sums[6] != 0.0 ? 1.0 : 0.0;
  return double(g_geom[0]) * gate_g * sums[9] == double(g_photo[0]) * sums[10];  // products of floats: exact
}

// What the next forward speculates on (scsfm_pair_desc::hint): the upstream gradients this backward saw -- unless they
// are not finite (the overflow step of a loss-scaled run: NaN / Inf say nothing about the loop's weights, and a NaN in
// the hint would make every later forward run its speculative pass for nothing); such a step leaves the hint alone.
// A ZERO photo gradient is remembered: a run with -p 0 then takes the plain forward from its second step on.
template <typename T>
__device__ __forceinline__ void remember_upstream(double* __restrict__ hint, const T* __restrict__ g_photo,
                                                  const T* __restrict__ g_geom) {
  const double a = double(g_photo[0]), b = double(g_geom[0]);
  if (a - a == 0.0 && b - b == 0.0) { hint[0] = a; hint[1] = b; }  // (the differences are NaN for NaN and +-Inf)
}

// Bit p: the backward has to run its two passes for pair p (its upstream coefficients are not both zero and the
// speculative forward did not already do the work).  Evaluated once per workgroup, all pairs' loads in flight
// together; a launch in which no pair needs anything ends here.
template <typename T>
__device__ __forceinline__ unsigned pairs_to_run(const PairBatch<T>& pb, int npairs, const T* __restrict__ g_photo,
                                                 const T* __restrict__ g_geom) {
  unsigned live = 0;
#pragma unroll
  for (int p = 0; p < kMaxPairs; ++p) {
    if (p < npairs) {
      const double* __restrict__ s = pb.p[p].sums;
      const bool zero = T(s[5]) * g_photo[0] == T(0) && T(s[6]) * g_geom[0] == T(0);
      if (!zero && !spec_valid(s, g_photo, g_geom)) live |= 1u << p;
    }
  }
  return live;
}

#ifndef SCSFM_PHOTO_BLOCKS  // tuning knob (tools/build_variants.sh): workgroups per CU the tiled kernels are compiled for
#define SCSFM_PHOTO_BLOCKS 4
#endif
// kFlags: kRuntimeFlags = obey `flags_arg`; any other value = the flag word as a compile-time constant (the
// configuration every training run uses gets its own instantiation: its uniform branches fold away and the
// scheduler sees longer straight-line blocks).
constexpr unsigned kRuntimeFlags = 0xffffffffu;
constexpr unsigned kTrainFlags = SCSFM_WITH_SSIM | SCSFM_WITH_MASK | SCSFM_WITH_AUTO_MASK;  // zeros padding
}  // namespace scsfm
#include "scsfm_spec_tile.h"  // the speculative forward proper (needs PairArgs / PairBatch / the plane indices above)
#ifdef SCSFM_WITH_MARCH       // tuning / test builds only (tools/build_variants.sh, tests/hostsim; -Ivariants/src): the column-march
#include "scsfm_march.h"      // variant of the speculative forward (variants/src/), selected at run time with SCSFM_SPEC_KERNEL=march
#endif
namespace scsfm {

// The speculative forward: one tile per workgroup, XCD-aware order.
// (kStageFwd: the forward warp's taps staged in LDS as well -- scsfm_spec_tile.h; tuning / test builds instantiate both)
template <typename T, bool kSsim, unsigned kFlags = kRuntimeFlags, bool kScaled = false, bool kStageFwd = false>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? SCSFM_PHOTO_BLOCKS : 1) void pair_fwd_spec_kernel(PairBatch<T> pb, int B, int H, int W,
                                                                                  unsigned flags, T r_hint,
                                                                                  const double* __restrict__ hint, T sm_icx, T sm_icy) {
  if (hint) r_hint = hint[0] != 0.0 ? T(3.0 * hint[1] / hint[0]) : T(0);  // (scsfm_pair_desc::hint: the device's pair wins)
  spec_tile<T, kSsim, kScaled, kFlags, kStageFwd>(xcd_block_id(), (int)gridDim.x, (int)gridDim.y, pb, B, H, W, flags, nullptr, nullptr, r_hint,
                                                  sm_icx, sm_icy);
}
#ifdef SCSFM_WITH_MARCH
#ifndef SCSFM_MARCH_WAVES_PER_SIMD  // waves per SIMD the march is compiled for (168 VGPRs at 3: no spills; 128 at 4: spills)
#define SCSFM_MARCH_WAVES_PER_SIMD 3
#endif
// ... and as a column march: one (band, segment) of a (pair, batch element) per workgroup (SCSFM_SPEC_KERNEL=march).
template <typename T, bool kSsim, unsigned kFlags = kRuntimeFlags, bool kScaled = false>
__global__ __launch_bounds__(March<T>::kWaves * kWave, sizeof(T) == 4 ? SCSFM_MARCH_WAVES_PER_SIMD : 1) void pair_march_kernel(
    PairBatch<T> pb, int B, int H, int W, unsigned flags, T r_hint, int seg_rows, const double* __restrict__ hint) {
  if (hint) r_hint = hint[0] != 0.0 ? T(3.0 * hint[1] / hint[0]) : T(0);
  march_segment<T, kSsim, kScaled, kFlags>(xcd_block_id(), (int)gridDim.x, (int)gridDim.y, seg_rows, pb, B, H, W, flags, r_hint);
}
#endif

// One tile of pass A (blk = logical tile of an nbx x nby x (pairs * B) tiling).
// kImages (scsfm_pairs_bwd_inputs): the same pass produces the gradients of the two IMAGES instead of the planes for
// pass B -- dL/d(warped colour) is splatted over the reference image's taps (grid_sampler_2d_backward on its input,
// inverse_warp.py:262), and dL/d(target colour) -- the x side of the SSIM term, whose E[x^2] and E[xy] maps are those of
// the y side, and the L1 term with the opposite sign -- is added to the target image's gradient.  Atomic adds: an image
// is the target of one pair-direction and the reference of another.
template <typename T>
struct ImageGrads {
  T* tgt[kMaxPairs];
  T* ref[kMaxPairs];
};
template <typename T, bool kSsim, bool kScaled, bool kImages = false>
__device__ __forceinline__ void photo_tile(const BlockId blk, int nbx, int nby, const PairBatch<T>& pb, int B, int H,
                                           int W, unsigned flags, const T* __restrict__ g_photo,
                                           const T* __restrict__ g_geom, const ImageGrads<T>* ig = nullptr) {
  const int pair = blk.z / B, b = blk.z - pair * B;
  const PairArgs<T>& pa = pb.p[pair];
  const T* __restrict__ tgt_img = pa.tgt_img;
  const T* __restrict__ ref_img = pa.ref_img;
  const BatchConsts<T>* __restrict__ consts = pa.consts;
  const double* __restrict__ sums = pa.sums;
  T* __restrict__ gbuf = pa.gbuf;
  typedef typename Vec2<T>::type V2;
  constexpr int TH = Tile<T>::kH, STRIP = TH / (kThreads / kWave);
  __shared__ V2 sXY[kSsim ? 3 : 1][kSsim ? TH + 2 : 1][kSsim ? kHaloW : 1];
  constexpr int NMAP = kImages ? 4 : 3;
  __shared__ T sG[kSsim ? NMAP : 1][kSsim ? TH : 1][kSsim ? kTileW : 1];  // 1/9 (g_mu_y, g_E[y^2], g_E[xy] [, g_mu_x]), one colour

  // upstream gradient x d(masked mean)/d(sum): zero when the 10000-pixel gate was closed
  const T a = T(sums[5]) * g_photo[0];
  const T bg = T(sums[6]) * g_geom[0];
  if (a == T(0) && bg == T(0)) return;        // workgroup-uniform: pass B skips as well
  if (!kImages && spec_valid(sums, g_photo, g_geom)) return;  // the forward already left the planes in gbuf
  T* __restrict__ gi_tgt = kImages ? ig->tgt[pair] : nullptr;
  T* __restrict__ gi_ref = kImages ? ig->ref[pair] : nullptr;
  if (kImages && !gi_tgt && !gi_ref) return;
  Sample<T> ss[kImages ? STRIP : 1];

  const int col = threadIdx.x & (kWave - 1), strip = threadIdx.x / kWave;
  // the 64 x TH compute domain starts one pixel before the 62 x (TH-2) block of outputs
  const int ox = blk.x * (kTileW - 2) - 1, oy = blk.y * (TH - 2) - 1;
  const bool with_mask = (flags & SCSFM_WITH_MASK) != 0, with_auto = (flags & SCSFM_WITH_AUTO_MASK) != 0;
  const BatchConsts<T> bc = consts[b];
  const unsigned plane = unsigned(H) * unsigned(W);
  const size_t gplane = (size_t)B * plane;  // one gbuf plane spans the whole batch
  tgt_img += (size_t)b * 3 * plane;
  ref_img += (size_t)b * 3 * plane;
  const DepthMap<T, kScaled> tgt_depth = depth_map<kScaled>(pa.tgt_depth, b, H, W, pa.ds);
  const DepthMap<T, kScaled> ref_depth = depth_map<kScaled>(pa.ref_depth, b, H, W, pa.ds);
  gbuf += (size_t)b * plane;

  const int px = ox + col, py0 = oy + strip * STRIP;
  const bool in_x = col >= 1 && col <= kTileW - 2 && px < W;
  T coef[STRIP];  // a * m * (1 - dd): weight of blend_c(q) in the loss
  T mq[STRIP];    // mask of the owned pixel
  T bsum[STRIP];  // sum_c blend_c of the owned pixel
  V2 cen[kSsim ? 1 : STRIP][kSsim ? 1 : 3];
  // ---- phase 0: every streaming load of the strip and of this thread's ring pixel ---------------
  const int u = reflect_index(px, W);
  T in_d[STRIP], in_t[STRIP][3], in_r[STRIP][3], rin_d = T(0), rin_t[3] = {T(0), T(0), T(0)}, rin_r[3];
#pragma unroll
  for (int k = 0; k < STRIP; ++k)
    load_pixel(u, reflect_index(py0 + k, H), W, plane, tgt_img, ref_img, tgt_depth, with_auto, in_d[k], in_t[k],
               in_r[k]);
  const bool has_ring = kSsim && threadIdx.x < 2 * kHaloW + 2 * TH;
  int ru = 0, rv = 0, rhy = 0, rhx = 0;
  if (has_ring) {
    ring_pos<TH>(threadIdx.x, rhy, rhx);
    ru = reflect_index(ox + rhx - 1, W); rv = reflect_index(oy + rhy - 1, H);
    load_pixel(ru, rv, W, plane, tgt_img, ref_img, tgt_depth, false, rin_d, rin_t, rin_r);
  }
  // ---- phase 1a ------------------------------------------------------------------------------
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, py = py0 + k;
    const bool inimg = px >= 0 && px < W && py >= 0 && py < H;
    const int v = reflect_index(py, H);
    V2 xy[3];
    const Sample<T> s = warp_colours(bc, u, v, in_d[k], in_t[k], H, W, flags, ref_img, xy);
    if constexpr (kImages) ss[k] = s;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      if constexpr (kSsim) sXY[c][ly + 1][col + 1] = xy[c]; else cen[k][c] = xy[c];
    }
    const T Dp = bilerp_rows(ref_depth.taps(s), s);
    const T ddk = clamp01(t_abs(s.Z - Dp) * t_rcp(s.Z + Dp));
    mq[k] = inimg ? pixel_mask(s, with_auto, xy, in_r[k]) : T(0);
    coef[k] = a * mq[k] * (with_mask ? (T(1) - ddk) : T(1));
    bsum[k] = T(0);
  }
  // ---- phase 1b: ring ------------------------------------------------------------------------
  if constexpr (kSsim) {
    if (has_ring) {
      V2 xy[3];
      warp_colours(bc, ru, rv, rin_d, rin_t, H, W, flags, ref_img, xy);
#pragma unroll
      for (int c = 0; c < 3; ++c) sXY[c][rhy][rhx] = xy[c];
    }
    __syncthreads();
  }
  // ---- phases 2/3, one colour channel at a time ------------------------------------------------
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    T gI[STRIP], gIt[kImages ? STRIP : 1];
    if constexpr (kSsim) {
      // phase 2: forward statistics at every owned pixel q; publish 1/9 (g_mu_y, g_E[y^2], g_E[xy])(q)
      WinSums<T> ws[STRIP];
      V2 centre[STRIP];
      strip_window_sums<T, STRIP>(sXY[c], strip * STRIP, col, ws, centre);
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const SsimStats<T> st = ssim_stats(ws[k]);
        bsum[k] += T(0.85) * clamp01(st.raw);
        // s = clamp((1 - S)/2, 0, 1): d s / d S = -1/2 inside the clamp (inclusive bounds)
        const T gS = clamp01(st.raw) == st.raw ? coef[k] * T(0.85) * T(-0.5) : T(0);  // (i.e. 0 <= raw <= 1)
        T g1, g2, g3;
        ssim_grad_y(st, gS, g1, g2, g3);
        const int ly = strip * STRIP + k;
        sG[0][ly][col] = g1; sG[1][ly][col] = g2; sG[2][ly][col] = g3;
        if constexpr (kImages) { T g1x; ssim_grad_x(st, gS, g1x); sG[NMAP - 1][ly][col] = g1x; }
      }
      __syncthreads();
      // phase 3: transpose of (reflect-pad + 3x3 box) as a separable 3x3 gather
      T gt[STRIP][NMAP];
      strip_box_transpose<T, STRIP, TH, NMAP>(sG, strip * STRIP, col, px, py0, H, W, gt);
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const T x = centre[k][0], y = centre[k][1], d = x - y;
        bsum[k] += T(0.15) * clamp01(t_abs(d));
        // d clamp(|d|, 0, 1) / d Iw: the clamp passes gradient on [0, 1] inclusive, abs uses sgn
        const T l1g = (t_abs(d) <= T(1)) ? -t_sgn(d) : T(0);
        gI[k] = gt[k][0] + T(2) * y * gt[k][1] + x * gt[k][2] + coef[k] * T(0.15) * l1g;
        if constexpr (kImages) gIt[k] = gt[k][NMAP - 1] + T(2) * x * gt[k][1] + y * gt[k][2] - coef[k] * T(0.15) * l1g;
      }
      if (c < 2) __syncthreads();  // sG is rewritten by the next colour
    } else {
#pragma unroll
      for (int k = 0; k < STRIP; ++k) {
        const T d = cen[k][c][0] - cen[k][c][1];
        bsum[k] += clamp01(t_abs(d));
        gI[k] = coef[k] * ((t_abs(d) <= T(1)) ? -t_sgn(d) : T(0));
        if constexpr (kImages) gIt[k] = -gI[k];
      }
    }
#pragma unroll
    for (int k = 0; k < STRIP; ++k) {
      const int ly = strip * STRIP + k, py = py0 + k;
      // note: m(p) = 0 still receives SSIM gradient through its neighbours' windows
      if (in_x && ly >= 1 && ly <= TH - 2 && py < H) {
        if constexpr (kImages) {
          const size_t cplane = ((size_t)b * 3 + c) * plane;
          if (gi_tgt) atomicAdd(gi_tgt + cplane + unsigned(py) * unsigned(W) + unsigned(px), gIt[k]);
          if (gi_ref) scatter_taps(gi_ref + cplane, ss[k], gI[k]);
        } else {
          st_at(gbuf + (kPlaneGI + c) * gplane, (unsigned(py) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), gI[k]);
        }
      }
    }
  }
  if constexpr (kImages) return;
  // dL/d diff_depth: directly (geometry loss) and through the weight mask (no detach, loss_functions.py:111-113);
  // handed to pass B.  This tile's part of the pair's scatter plane is cleared on the way (pass B only runs when
  // this pass did)
#pragma unroll
  for (int k = 0; k < STRIP; ++k) {
    const int ly = strip * STRIP + k, py = py0 + k;
    if (in_x && ly >= 1 && ly <= TH - 2 && py < H) {
      const unsigned off = (unsigned(py) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T));
      st_at(gbuf + kPlaneGdd * gplane, off, bg * mq[k] - (with_mask ? a * mq[k] * bsum[k] : T(0)));
      st_at(gbuf + kPlaneScatter * gplane, off, T(0));
    }
  }
}

// Pass A of the backward.  Launched with a small persistent grid that walks the tiles: when the speculative
// forward's results stand (the usual case) every tile returns at once and the launch costs ~1 us instead of
// the ~12 us of ten thousand empty workgroups.
template <typename T, bool kSsim, bool kScaled>
// (three workgroups per CU: at four the tile code of this pass spilled six registers -- it only ever runs on the first
// step after the loss weights changed)
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? 3 : 1) void pair_bwd_photo_kernel(
    PairBatch<T> pb, int nbx, int nby, int nz, int B, int H, int W, unsigned flags, const T* __restrict__ g_photo,
    const T* __restrict__ g_geom) {
  const unsigned live = pairs_to_run(pb, nz / B, g_photo, g_geom);
  if (!live) return;
  const int n = nbx * nby * nz;
  for (int t = blockIdx.x; t < n; t += gridDim.x) {
    const BlockId blk = xcd_tile_of(t, nbx, nby, nz);
    if (!((live >> (blk.z / B)) & 1u)) continue;
    photo_tile<T, kSsim, kScaled>(blk, nbx, nby, pb, B, H, W, flags, g_photo, g_geom);
    __syncthreads();  // the tile's LDS is reused
  }
}

// The image gradients of up to kMaxPairs pair-directions (scsfm_pairs_bwd_inputs): one tile per workgroup.
template <typename T, bool kSsim, bool kScaled>
__global__ __launch_bounds__(kThreads) void pair_bwd_images_kernel(PairBatch<T> pb, ImageGrads<T> ig, int B, int H, int W,
                                                                   unsigned flags, const T* __restrict__ g_photo,
                                                                   const T* __restrict__ g_geom) {
  photo_tile<T, kSsim, kScaled, true>(xcd_block_id(), (int)gridDim.x, (int)gridDim.y, pb, B, H, W, flags, g_photo, g_geom, &ig);
}

// ==========================================================================================
// Backward, pass B (per pixel): through the bilinear sampler and the camera geometry.
//   dL/d(ix, iy) = sum_c dL/dI_w,c * dI_w,c/d(ix, iy) + dL/dD_p * dD_p/d(ix, iy)
//   dL/d ref_depth: atomic scatter of dL/dD_p over the taps; dL/d tgt_depth: dense accumulate;
//   dL/d(A|c): block reduction, fp64 atomics per batch element.
// ==========================================================================================
#ifndef SCSFM_GEOM_BLOCKS  // tuning knob: workgroups per CU the geometry pass is compiled for
#define SCSFM_GEOM_BLOCKS 4
#endif
template <typename T, bool kScaled>
__device__ __forceinline__ void geom_tile(const BlockId blk, int nbx, int nby, const PairBatch<T>& pb, int B, int H, int W,
                                          unsigned flags, const T* __restrict__ g_photo, const T* __restrict__ g_geom) {
  const int pair = blk.z / B, b = blk.z - pair * B;
  const PairArgs<T>& pa = pb.p[pair];
  const T* __restrict__ ref_img = pa.ref_img;
  const BatchConsts<T>* __restrict__ consts = pa.consts;
  const double* __restrict__ sums = pa.sums;
  const T* __restrict__ gbuf = pa.gbuf;
  double* __restrict__ gP = pa.gPp;
  constexpr int ROWS = kGeomRows;  // pixels per thread: a block covers a 64 x (4 ROWS) tile
  __shared__ double red[12 * (kThreads / kWave)];
  // FLOATING cells (round 6).  Rounds 4-5 staged this pass's scatter in the fixed-point cells of the speculative forward
  // (-10 us on a pass that runs once per change of the loss weights) behind a two-corner compression heuristic; unlike
  // the speculative tile this pass has no cheap bound of what a tile adds to one cell before it scatters (its terms carry
  // the final coefficients and Z, D_p only exist per pixel), so the exact guarantee here is a cell that cannot wrap: the
  // LDS float atomic, lane by lane.  Order-dependent in the last ulp, like the direct atomics of the taps that miss the
  // window (and like grid_sampler_2d_backward itself, inverse_warp.py:267).
  typedef T Cell;
  __shared__ Cell win[kWinH][kWinW];  // staging window of the scatter into dL/d ref_depth
  if (T(sums[5]) * g_photo[0] == T(0) && T(sums[6]) * g_geom[0] == T(0)) return;  // as in pass A
  if (spec_valid(sums, g_photo, g_geom)) return;  // the speculative forward already ran this pass in its tail
  const int px = blk.x * kWave + (threadIdx.x & (kWave - 1));
  const int py0 = (blk.y * (kThreads / kWave) + threadIdx.x / kWave) * ROWS;
  const BatchConsts<T> bc = consts[b];
  const unsigned plane = unsigned(H) * unsigned(W);
  const size_t gplane = (size_t)B * plane;
  ref_img += (size_t)b * 3 * plane;
  const DepthMap<T, kScaled> tgt_depth = depth_map<kScaled>(pa.tgt_depth, b, H, W, pa.ds);
  const DepthMap<T, kScaled> ref_depth = depth_map<kScaled>(pa.ref_depth, b, H, W, pa.ds);
  gbuf += (size_t)b * plane;
  // both outputs go to private planes of this pair's gbuf (dense: plain stores; scatter: atomics into the plane
  // pass A zeroed); pairs_combine_kernel adds them to the callers' buffers.  That keeps every pair-direction of
  // a step in one launch: the same depth map is the dense target of one pair and the scatter target of another.
  T* __restrict__ g_dense = pa.gbuf + kPlaneDense * gplane + (size_t)b * plane;
  T* __restrict__ g_scatter = pa.gbuf + kPlaneScatter * gplane + (size_t)b * plane;
  for (int i = threadIdx.x; i < kWinW * kWinH; i += kThreads) (&win[0][0])[i] = Cell(0);
  __syncthreads();
  int wx0, wy0;
  window_origin<T, kWinW, kWinH>(bc, blk.x * kWave + kWave / 2, blk.y * (kThreads / kWave) * ROWS + 2 * ROWS, tgt_depth,
                                 H, W, flags, wx0, wy0);
  T acc[12];
#pragma unroll
  for (int i = 0; i < 12; ++i) acc[i] = T(0);
  // all streaming loads of the strip first (5 per pixel), so that they are in flight together before the
  // first dependent gather
  T in_d[ROWS], in_g[ROWS][4];
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int py = py0 + r;
    const int cy = py < H ? py : H - 1, cx = px < W ? px : W - 1;
    const unsigned p = (unsigned(cy) * unsigned(W) + unsigned(cx)) * unsigned(sizeof(T));
    in_d[r] = tgt_depth.at(cx, cy, p);
#pragma unroll
    for (int c = 0; c < 3; ++c) in_g[r][c] = ld_at(gbuf + (kPlaneGI + c) * gplane, p);
    in_g[r][3] = ld_at(gbuf + kPlaneGdd * gplane, p);
  }
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    const int py = py0 + r;
    if (px >= W || py >= H) continue;
    const T gI[3] = {in_g[r][0], in_g[r][1], in_g[r][2]};
    const T gd = geom_pixel<T, Cell, kWinW, kWinH>(bc, px, py, in_d[r], gI, in_g[r][3], ref_img, ref_depth, plane, H, W, flags,
                                             win, wx0, wy0, g_scatter, acc, T(1), nullptr);
    st_at(g_dense, (unsigned(py) * unsigned(W) + unsigned(px)) * unsigned(sizeof(T)), (flags & SCSFM_DEBUG_X2) ? T(0) : gd);
  }
  __syncthreads();
  if (!(flags & (SCSFM_DEBUG_X1 | SCSFM_DEBUG_X5))) flush_scatter_window<T, Cell, kWinW, kWinH>(win, wx0, wy0, g_scatter, W, T(1));
  if (flags & SCSFM_DEBUG_X3) {  // profiling: keep the partials defined
    if (threadIdx.x == 0)
      for (int i = 0; i < 12; ++i) gP[12 * ((size_t)(b * nby + blk.y) * nbx + blk.x) + i] = 0.0;
    return;
  }
  block_sum<12>(acc, red);
  if (threadIdx.x == 0) {
    double* o = gP + 12 * ((size_t)(b * nby + blk.y) * nbx + blk.x);
#pragma unroll
    for (int i = 0; i < 12; ++i) o[i] = double(acc[i]);  // (raw: the pose reducer applies K^-1)
  }
}

// Persistent grid over the tiles, natural order (with the XCD-contiguous order of the tiled kernels this pass
// measured 2 % slower: its scatter / flush atomics then hit one image's lines from a single XCD at a time).
// Like pass A it returns at once per tile when the speculative forward's results stand.
template <typename T, bool kScaled>
__global__ __launch_bounds__(kThreads, sizeof(T) == 4 ? SCSFM_GEOM_BLOCKS : 1) void pair_bwd_geom_kernel(
    PairBatch<T> pb, int nbx, int nby, int nz, int B, int H, int W, unsigned flags, const T* __restrict__ g_photo,
    const T* __restrict__ g_geom) {
  const unsigned live = pairs_to_run(pb, nz / B, g_photo, g_geom);
  if (!live) return;
  const int n = nbx * nby * nz;
  for (int t = blockIdx.x; t < n; t += gridDim.x) {
    BlockId blk;
    blk.x = t % nbx;
    const int q = t / nbx;
    blk.y = q % nby;
    blk.z = q / nby;
    if (!((live >> (blk.z / B)) & 1u)) continue;
    geom_tile<T, kScaled>(blk, nbx, nby, pb, B, H, W, flags, g_photo, g_geom);
    __syncthreads();  // the window is reused
  }
}

// The factor the planes and pose partials of a pair still lack: those of a (valid) speculative forward were
// computed with unit photo coefficient, those of the backward's own passes are final.
template <typename T>
__device__ __forceinline__ T pair_scale(const double* __restrict__ sums, const T* __restrict__ g_photo,
                                        const T* __restrict__ g_geom) {
  return spec_valid(sums, g_photo, g_geom) ? T(sums[5]) * g_photo[0] : T(1);
}

// After a backward whose upstream gradients did NOT stand in the hinted ratio, passes A and B have overwritten the
// speculative planes and pose partials with fully scaled values: the workspace no longer holds a speculative
// forward.  Clearing the flag makes any later backward on the same workspace (retain_graph=True) recompute with
// its own coefficients instead of rescaling those planes once more.  The flag only ever goes 1 -> 0 while
// spec_valid() is already false, so the other workgroups of the launch that evaluate spec_valid() concurrently
// see the same answer either way.  (A backward with both coefficients zero ran no pass and leaves it alone.)
template <typename T>
__device__ __forceinline__ void retire_speculation(double* __restrict__ sums, const T* __restrict__ g_photo,
                                                   const T* __restrict__ g_geom) {
  const bool zero = T(sums[5]) * g_photo[0] == T(0) && T(sums[6]) * g_geom[0] == T(0);
  if (sums[8] != 0.0 && !zero && !spec_valid(sums, g_photo, g_geom)) sums[8] = 0.0;
}

// One wave per (pair, batch element): reduce the per-block partials of dL/d(A|c), finish dL/dpose.
// nblk_spec / nblk_geom: blocks per image of the kernel that wrote them (geometry tail of the speculative
// forward, or the geometry pass).
template <typename T>
__global__ void pairs_pose_reduce_kernel(PairBatch<T> pb, int B, int nblk_geom, const T* __restrict__ K,
                                         const T* __restrict__ g_photo, const T* __restrict__ g_geom, double* __restrict__ hint) {
  const int pair = blockIdx.x / B, b = blockIdx.x - pair * B;
  if (hint && blockIdx.x == 0 && threadIdx.x == 0) remember_upstream(hint, g_photo, g_geom);
  const PairArgs<T>& pa = pb.p[pair];
  const bool spec = spec_valid(pa.sums, g_photo, g_geom);
  pose_reduce_one(b, spec ? int(pa.sums[11]) : nblk_geom, double(pair_scale(pa.sums, g_photo, g_geom)), pa.pose, K, pa.consts,
                  pa.gPp, pa.sums, g_photo, g_geom, pa.g_pose);
  if (b == 0 && threadIdx.x == 0) retire_speculation(pa.sums, g_photo, g_geom);
}

// dL/d intrinsics [B,3,3] of up to kMaxPairs pair-directions: one wave per batch element re-reduces every pair's pose
// partials (still in the workspaces after scsfm_pairs_bwd, in the state pairs_pose_reduce_kernel found them) and adds
// up the pairs' contributions in pair order (fp64).
template <typename T>
__global__ void pairs_intrinsics_kernel(PairBatch<T> pb, int npairs, int B, int nblk_geom, const T* __restrict__ K,
                                        const T* __restrict__ g_photo, const T* __restrict__ g_geom, T* __restrict__ gK,
                                        int accumulate) {
  const int b = blockIdx.x;
  double tot[9];
#pragma unroll
  for (int i = 0; i < 9; ++i) tot[i] = 0.0;
  for (int p = 0; p < npairs; ++p) {
    const PairArgs<T>& pa = pb.p[p];
    const bool spec = spec_valid(pa.sums, g_photo, g_geom);
    const bool live = !(T(pa.sums[5]) * g_photo[0] == T(0) && T(pa.sums[6]) * g_geom[0] == T(0));
    double g[12];
    pose_partials_sum(b, spec ? int(pa.sums[11]) : nblk_geom, double(pair_scale(pa.sums, g_photo, g_geom)), pa.consts, pa.gPp,
                      live, g);
    if (threadIdx.x == 0) intrinsics_from_gP(K + 9 * b, pa.pose + 6 * b, g, false, tot);
  }
  if (threadIdx.x == 0) {
#pragma unroll
    for (int i = 0; i < 9; ++i) gK[9 * b + i] = accumulate ? gK[9 * b + i] + T(tot[i]) : T(tot[i]);
  }
}

// dst[d] (+)= sum_k scale_k * src_k: the dense planes of the pairs whose target depth map dst is and the scatter
// planes of the pairs that sampled it.  A pair whose passes skipped (both upstream coefficients zero) left its
// planes untouched and is skipped here too.  The first row of workgroups (blockIdx.y == 0) does what
// pairs_pose_reduce_kernel does, one wave per (pair, batch element): dL/dpose rides along for free.
template <typename T>
struct CombineSrc {
  const T* plane;
  const double* sums;
  int dst;  // index into CombineBatch::dst
};
template <typename T>
struct CombineBatch {
  T* dst[2 * kMaxPairs];
  int store[2 * kMaxPairs];  // 1: dst = sum (the first time a call touches this buffer), 0: dst += sum
  int ds[2 * kMaxPairs];     // dst is the gradient of a [H >> ds, W >> ds] map: the planes are sum-pooled into it
  int nd, nsrc;
  CombineSrc<T> src[2 * kMaxPairs];  // sorted by dst: the sources of dst d are src[first[d] .. first[d] + count[d])
  int first[2 * kMaxPairs], count[2 * kMaxPairs];
  // scsfm_pairs_bwd_smooth: the smooth loss's gradient of the same map rides along (loss_functions.py:132-159; what
  // smooth_bwd_kernel would add in a launch of its own, re-reading and re-writing the map): dst += g_smooth *
  // (edge * (1 / den_b) - L_b / (den_b^2 H W)) with edge = the plane the smooth forward left and per_img = its
  // {den_b, L_b} records.  nullptr: nothing to add.
  const T* sm_edge[2 * kMaxPairs];
  const double* sm_img[2 * kMaxPairs];
  const T* g_smooth;
};
// Per-image constants of the smooth term of destination d for image b: v = e * c.x - c.y.
template <typename T>
struct SmoothCoef { T x, y; };
template <typename T>
__device__ __forceinline__ SmoothCoef<T> smooth_coef(const CombineBatch<T>& cb, int d, int b, size_t plane) {
  SmoothCoef<T> c;
  const double den = cb.sm_img[d][2 * b], L = cb.sm_img[d][2 * b + 1];
  const T g = cb.g_smooth[0];
  c.x = g * T(1.0 / den);
  c.y = g * T(L / (den * den * (double)plane));
  return c;
}

template <typename T>
struct alignas(16) Quad { T v[16 / sizeof(T)]; };  // 16-byte vector access

// This destination's sources, with the factor each still lacks (0: nothing to add).
template <typename T>
__device__ __forceinline__ bool combine_sources(const CombineBatch<T>& cb, int d, const T* __restrict__ g_photo,
                                                const T* __restrict__ g_geom, const T* (&src)[2 * kMaxPairs],
                                                T (&scale)[2 * kMaxPairs]) {
  bool aligned = (reinterpret_cast<size_t>(cb.dst[d]) & 15) == 0;
#pragma unroll
  for (int k = 0; k < 2 * kMaxPairs; ++k) {
    src[k] = nullptr;
    scale[k] = T(0);
    if (k < cb.nsrc && cb.src[k].dst == d) {
      const double* s = cb.src[k].sums;
      const bool live = !(T(s[5]) * g_photo[0] == T(0) && T(s[6]) * g_geom[0] == T(0));
      src[k] = cb.src[k].plane;
      scale[k] = live ? pair_scale(s, g_photo, g_geom) : T(0);
      aligned = aligned && (reinterpret_cast<size_t>(src[k]) & 15) == 0;
    }
  }
  return aligned;
}

// The 16-byte part of one destination with exactly NS sources, all loads of a thread's (up to) two quads issued before
// the first is used.  A source whose pair is not live (both upstream coefficients zero: its planes may never have been
// written) is replaced by a live one with weight 0, so that no load sits behind a branch; returns false (nothing done)
// when no source is live and the general loop should store the zeros.
template <typename T, int NS>
__device__ __forceinline__ bool combine_quads(const CombineBatch<T>& cb, int d, size_t q0, size_t nq, bool store,
                                              const T* __restrict__ g_photo, const T* __restrict__ g_geom, bool smooth,
                                              SmoothCoef<T> sm) {
  constexpr int Q = 16 / sizeof(T);
  const Quad<T>* p[NS];
  T sc[NS];
  const Quad<T>* safe = nullptr;
  const int f = cb.first[d];
#pragma unroll
  for (int j = 0; j < NS; ++j) {
    const CombineSrc<T>& c = cb.src[f + j];
    const bool live = !(T(c.sums[5]) * g_photo[0] == T(0) && T(c.sums[6]) * g_geom[0] == T(0));
    sc[j] = live ? pair_scale(c.sums, g_photo, g_geom) : T(0);
    p[j] = reinterpret_cast<const Quad<T>*>(c.plane);
    if (sc[j] != T(0) && !safe) safe = p[j];
  }
  if (!safe) return false;
#pragma unroll
  for (int j = 0; j < NS; ++j) p[j] = sc[j] != T(0) ? p[j] : safe;
  Quad<T>* __restrict__ out = reinterpret_cast<Quad<T>*>(cb.dst[d]);
  // (without a smooth term the edge loads read a live plane with weight 0: no load behind a branch)
  const Quad<T>* __restrict__ pe = smooth ? reinterpret_cast<const Quad<T>*>(cb.sm_edge[d]) : safe;
  const T ex = smooth ? sm.x : T(0), ey = smooth ? sm.y : T(0);
  const size_t stride = (size_t)gridDim.x * kThreads, end = q0 + nq;
  size_t i = q0 + (size_t)blockIdx.x * kThreads + threadIdx.x;
  for (; i + stride < end; i += 2 * stride) {
    Quad<T> x[2][NS], o[2], e[2];
#pragma unroll
    for (int j = 0; j < NS; ++j) { x[0][j] = p[j][i]; x[1][j] = p[j][i + stride]; }
    e[0] = pe[i]; e[1] = pe[i + stride];
    if (!store) { o[0] = out[i]; o[1] = out[i + stride]; }
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      Quad<T> acc;
#pragma unroll
      for (int q = 0; q < Q; ++q) acc.v[q] = store ? T(0) : o[h].v[q];
#pragma unroll
      for (int j = 0; j < NS; ++j)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc.v[q] += sc[j] * x[h][j].v[q];
#pragma unroll
      for (int q = 0; q < Q; ++q) acc.v[q] += ex * e[h].v[q] - ey;
      out[i + h * stride] = acc;
    }
  }
  for (; i < end; i += stride) {
    Quad<T> acc;
#pragma unroll
    for (int q = 0; q < Q; ++q) acc.v[q] = T(0);
    if (!store) acc = out[i];
#pragma unroll
    for (int j = 0; j < NS; ++j) {
      const Quad<T> x = p[j][i];
#pragma unroll
      for (int q = 0; q < Q; ++q) acc.v[q] += sc[j] * x.v[q];
    }
    const Quad<T> e = pe[i];
#pragma unroll
    for (int q = 0; q < Q; ++q) acc.v[q] += ex * e.v[q] - ey;
    out[i] = acc;
  }
  return true;
}

template <typename T>
__global__ __launch_bounds__(kThreads) void pairs_combine_kernel(CombineBatch<T> cb, size_t n, PairBatch<T> pb, int npairs,
                                                                 int B, int nblk_geom,
                                                                 const T* __restrict__ K, const T* __restrict__ g_photo,
                                                                 const T* __restrict__ g_geom, double* __restrict__ hint) {
  // The upstream gradients this backward saw are what the next forward speculates on (scsfm_pair_desc::hint; nothing
  // in this launch or before it on the stream reads the two doubles any more: the forward kernels did).
  if (hint && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) remember_upstream(hint, g_photo, g_geom);
  // row 0 of the grid is dispatched first: the pose waves (one latency-bound reduction each) start at once and
  // finish under the streaming rows instead of after them
  const int d = (int)blockIdx.y - 1;
  if (d < 0) {  // the pose row
    if (blockIdx.z != 0) return;
    const int item = blockIdx.x * (kThreads / kWave) + threadIdx.x / kWave;
    if (item < npairs * B) {
      const int pair = item / B, b = item - pair * B;
      const PairArgs<T>& pa = pb.p[pair];
      const bool spec = spec_valid(pa.sums, g_photo, g_geom);
      pose_reduce_one(b, spec ? int(pa.sums[11]) : nblk_geom, double(pair_scale(pa.sums, g_photo, g_geom)), pa.pose, K,
                      pa.consts, pa.gPp, pa.sums, g_photo, g_geom, pa.g_pose);
      if (b == 0 && (threadIdx.x & (kWave - 1)) == 0) retire_speculation(pa.sums, g_photo, g_geom);
    }
    return;
  }
  if (cb.ds[d]) return;  // a coarser scale's map: pairs_combine_pooled_kernel
  constexpr int Q = 16 / sizeof(T);
  // one image of the batch per blockIdx.z: the smooth term's per-image constants are workgroup-uniform
  const int b = blockIdx.z;
  const size_t plane = n / (size_t)B, e0 = (size_t)b * plane;
  T* __restrict__ dst = cb.dst[d];
  const bool store = cb.store[d] != 0;
  const bool smooth = cb.sm_edge[d] != nullptr;
  SmoothCoef<T> sm;
  sm.x = T(0); sm.y = T(0);
  if (smooth) sm = smooth_coef(cb, d, b, plane);
  const T* src[2 * kMaxPairs];
  T scale[2 * kMaxPairs];
  bool aligned = combine_sources(cb, d, g_photo, g_geom, src, scale);
  aligned = aligned && (plane % Q) == 0 && (!smooth || (reinterpret_cast<size_t>(cb.sm_edge[d]) & 15) == 0);
  // 16-byte accesses where every plane (and an image's first element in it) is 16-byte aligned; the scalar loop below
  // takes whatever is left
  const size_t nq = aligned ? plane / Q : 0, q0 = aligned ? e0 / Q : 0;
  // The usual shapes -- a map with 2 .. 4 sources (the dense plane of the pairs it is the target of, the scatter plane
  // of those that sampled it) -- run with every load of a thread's two quads in flight together (combine_quads);
  // anything else takes the general loop.
  bool done = false;
  switch (cb.count[d]) {
    case 2: done = combine_quads<T, 2>(cb, d, q0, nq, store, g_photo, g_geom, smooth, sm); break;
    case 3: done = combine_quads<T, 3>(cb, d, q0, nq, store, g_photo, g_geom, smooth, sm); break;
    case 4: done = combine_quads<T, 4>(cb, d, q0, nq, store, g_photo, g_geom, smooth, sm); break;
    default: break;
  }
  for (size_t i = q0 + (size_t)blockIdx.x * kThreads + threadIdx.x; !done && i < q0 + nq; i += (size_t)gridDim.x * kThreads) {
    Quad<T> acc;
#pragma unroll
    for (int j = 0; j < Q; ++j) acc.v[j] = T(0);
#pragma unroll
    for (int k = 0; k < 2 * kMaxPairs; ++k) {
      if (scale[k] != T(0)) {
        const Quad<T> x = reinterpret_cast<const Quad<T>*>(src[k])[i];
#pragma unroll
        for (int j = 0; j < Q; ++j) acc.v[j] += scale[k] * x.v[j];
      }
    }
    if (smooth) {
      const Quad<T> e = reinterpret_cast<const Quad<T>*>(cb.sm_edge[d])[i];
#pragma unroll
      for (int j = 0; j < Q; ++j) acc.v[j] += sm.x * e.v[j] - sm.y;
    }
    if (!store) {
      const Quad<T> o = reinterpret_cast<const Quad<T>*>(dst)[i];
#pragma unroll
      for (int j = 0; j < Q; ++j) acc.v[j] += o.v[j];
    }
    reinterpret_cast<Quad<T>*>(dst)[i] = acc;
  }
  for (size_t i = e0 + nq * Q + (size_t)blockIdx.x * kThreads + threadIdx.x; i < e0 + plane; i += (size_t)gridDim.x * kThreads) {
    T acc = T(0);
#pragma unroll
    for (int k = 0; k < 2 * kMaxPairs; ++k)
      if (scale[k] != T(0)) acc += scale[k] * src[k][i];
    if (smooth) acc += sm.x * cb.sm_edge[d][i] - sm.y;
    dst[i] = store ? acc : dst[i] + acc;
  }
}

// The destinations of a CombineBatch that are gradients of a coarser scale's map ([B, H >> ds, W >> ds]): the backward
// of the nearest up-sampling (loss_functions.py:77-82) is the sum over each 2^ds x 2^ds block of the full-resolution
// planes.  Launched only when a call has such destinations.
template <typename T>
__global__ __launch_bounds__(kThreads) void pairs_combine_pooled_kernel(CombineBatch<T> cb, size_t n, int W,
                                                                        const T* __restrict__ g_photo,
                                                                        const T* __restrict__ g_geom) {
  const int d = (int)blockIdx.y;
  const int ds = cb.ds[d];
  if (!ds) return;
  T* __restrict__ dst = cb.dst[d];
  const bool store = cb.store[d] != 0;
  const T* src[2 * kMaxPairs];
  T scale[2 * kMaxPairs];
  (void)combine_sources(cb, d, g_photo, g_geom, src, scale);
  // row r of the map (over batch and rows: H is a multiple of 2^ds) covers the full-resolution rows from r << ds
  const unsigned wl = unsigned(W) >> ds, side = 1u << ds;
  const size_t nl = n >> (2 * ds);
  for (size_t i = (size_t)blockIdx.x * kThreads + threadIdx.x; i < nl; i += (size_t)gridDim.x * kThreads) {
    const size_t r = i / wl;
    const unsigned xl = unsigned(i - r * wl);
    const size_t base = (r << ds) * unsigned(W) + (size_t(xl) << ds);
    T acc = T(0);
#pragma unroll
    for (int k = 0; k < 2 * kMaxPairs; ++k) {
      if (scale[k] != T(0)) {
        T sum = T(0);
        for (unsigned dy = 0; dy < side; ++dy)
          for (unsigned dx = 0; dx < side; ++dx) sum += src[k][base + (size_t)dy * unsigned(W) + dx];
        acc += scale[k] * sum;
      }
    }
    dst[i] = store ? acc : dst[i] + acc;
  }
}

// Before a speculative forward: clears the scatter plane of every pair of a batch and, in its first workgroup,
// does the work of pairs_prep_kernel (one launch instead of two in front of the dominant kernel).
template <typename T>
__global__ __launch_bounds__(kThreads) void pairs_zero_prep_kernel(PairBatch<T> pb, size_t n, int npairs, int B,
                                                                   const T* __restrict__ K) {
  // the last column of workgroups does not clear anything: its first one derives the per-image constants (a chain of
  // sincos / reciprocal latencies that would otherwise sit in front of one workgroup's share of the stream)
  if (blockIdx.x == gridDim.x - 1) {
    if (blockIdx.y == 0) {
      if (threadIdx.x == 0) *finalize_counter(pb) = 0u;
      for (int i = threadIdx.x; i < npairs * B; i += kThreads) {
        const int pair = i / B, b = i - pair * B;
        if (b == 0) *window_overflow_counter(pb.p[pair]) = 0u;
        prep_one(b, pb.p[pair].pose, K, pb.p[pair].consts);
      }
    }
    return;
  }
  T* __restrict__ p = pb.p[blockIdx.y].gbuf + kPlaneScatter * n;
  constexpr int Q = 16 / sizeof(T);
  const size_t stride = (size_t)(gridDim.x - 1) * kThreads, t0 = (size_t)blockIdx.x * kThreads + threadIdx.x;
  const size_t nq = (reinterpret_cast<size_t>(p) & 15) == 0 ? n / Q : 0;  // 16-byte stores where the plane allows
  Quad<T> z;
#pragma unroll
  for (int j = 0; j < Q; ++j) z.v[j] = T(0);
  for (size_t i = t0; i < nq; i += stride) reinterpret_cast<Quad<T>*>(p)[i] = z;
  for (size_t i = nq * Q + t0; i < n; i += stride) p[i] = T(0);
}

// ------------------------------------------------------------------------------------------
// Host side of the C ABI.
// ------------------------------------------------------------------------------------------
// scsfm_profile_begin / _end: event pairs around launches of the speculative forward.
struct ProfileState {
  hipEvent_t* start = nullptr;
  hipEvent_t* stop = nullptr;
  int n = 0, used = 0;
};
static ProfileState g_profile;

static void profile_release() {
  for (int i = 0; i < g_profile.n; ++i) { (void)hipEventDestroy(g_profile.start[i]); (void)hipEventDestroy(g_profile.stop[i]); }
  delete[] g_profile.start; delete[] g_profile.stop;
  g_profile = ProfileState();
}

template <typename T>
static PairArgs<T> make_pair_args(const scsfm_pair_desc& d, int B, int H, int W, void* shared_scratch, int idx) {
  const PairWs l = pair_ws_layout(B, H, W);
  char* base = reinterpret_cast<char*>(d.ws);
  PairArgs<T> a;
  a.tgt_img = (const T*)d.tgt_img; a.ref_img = (const T*)d.ref_img;
  a.tgt_depth = (const T*)d.tgt_depth; a.ref_depth = (const T*)d.ref_depth; a.pose = (const T*)d.pose;
  a.consts = reinterpret_cast<BatchConsts<T>*>(base);
  a.sums = reinterpret_cast<double*>(base + l.off_sums);
  a.partials = reinterpret_cast<double*>(base + l.off_partials);
  a.gPp = reinterpret_cast<double*>(base + l.off_gP);
  a.out = (T*)d.out;
  a.gbuf = d.gbuf ? (T*)d.gbuf
                  : (shared_scratch ? (T*)((char*)shared_scratch + (size_t)idx * scsfm_pair_bwd_scratch_bytes(B, H, W))
                                    : (T*)nullptr);
  a.g_ref_depth = (T*)d.g_ref_depth;
  a.g_pose = (T*)d.g_pose;
  a.ds = d.depth_shift;
  a.sm_partials = nullptr; a.sm_edge = nullptr; a.sm_img = nullptr; a.sm_out = nullptr;
  if (d.smooth_ws) {  // (only the speculative forward acts on it: pairs_fwd checks the descriptor)
    a.sm_partials = reinterpret_cast<double*>(base + l.off_smooth);
    a.sm_edge = (T*)d.smooth_edge;
    a.sm_img = reinterpret_cast<double*>(d.smooth_ws);
    a.sm_out = (T*)d.smooth_out;
  }
  return a;
}

// depth_shift: the depth maps are [B, 1, H >> s, W >> s]; H and W must be multiples of 2^s
static bool desc_inputs_ok(const scsfm_pair_desc& d, int H, int W) {
  const int s = d.depth_shift;
  if (s < 0 || s > 8 || ((H >> s) << s) != H || ((W >> s) << s) != W) return false;
  return d.tgt_img && d.ref_img && d.tgt_depth && d.ref_depth && d.pose && d.ws;
}

// Occupancy experiments: extra dynamic LDS per workgroup of the speculative forward (SCSFM_DEBUG_EXTRA_LDS=bytes; it
// only lowers the number of workgroups a CU holds).  Read once per process; 0 in production.
static unsigned debug_extra_lds() {
  static const unsigned bytes = [] {
    const char* e = getenv("SCSFM_DEBUG_EXTRA_LDS");
    return e ? (unsigned)strtoul(e, nullptr, 10) : 0u;
  }();
  return bytes;
}
// Which kernel serves the speculative forward: the tile kernel, or -- in builds that carry them -- the column march
// (SCSFM_SPEC_KERNEL=march) or the tile kernel with LDS-staged forward taps (SCSFM_SPEC_KERNEL=stagefwd); for A/B
// measurements and the CPU simulation, read per launch.
static bool spec_uses_march() {
#ifdef SCSFM_WITH_MARCH
  const char* e = getenv("SCSFM_SPEC_KERNEL");
  return e && e[0] == 'm';
#else
  return false;
#endif
}
static bool spec_stages_fwd() {
#ifdef SCSFM_WITH_MARCH
  const char* e = getenv("SCSFM_SPEC_KERNEL");
  return e && e[0] == 's';
#else
  return false;
#endif
}
#ifdef SCSFM_WITH_MARCH
// Rows of a segment of the speculative forward's column march (variants/src/scsfm_march.h).  A segment costs 4 extra warped rows
// and a last, mostly idle chunk, so segments should be long; the launch should still be many times the 1024
// workgroups the chip holds (4 per CU), so they cannot be too long.  SCSFM_MARCH_ROWS overrides (tests, tuning).
static int march_seg_rows(int H, int chunk, int units) {
  const char* e = getenv("SCSFM_MARCH_ROWS");  // (read per launch: tools/march_sweep.py varies it inside one process)
  const int forced = e ? atoi(e) : 0;
  int rows = forced > 0 ? forced : 64;
  if (forced <= 0) {
    // short of four rounds of workgroups: halve the segments (down to two chunks)
    while (rows > 2 * chunk && (long)units * ceil_div(H, rows) < 4 * 1024) rows /= 2;
  }
  rows = ceil_div(rows, chunk) * chunk;
  return rows < H ? rows : ceil_div(H, chunk) * chunk;
}
#endif

// Forward of up to kMaxPairs pair-directions per launch.  `spec`: every pair has a gbuf and w_photo != 0.
template <typename T>
static int pairs_fwd_chunk(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags, bool spec,
                           double w_photo, double w_geom, T* total, bool first, const double* hint, hipStream_t stream,
                           const StepTotal<T>& st) {
  PairBatch<T> pb;
  for (int i = 0; i < n; ++i) {
    pb.p[i] = make_pair_args<T>(d[i], B, H, W, nullptr, i);
    if (!spec) pb.p[i].sm_partials = nullptr;  // (pairs_fwd rejected such descriptors already)
  }
  const bool kernel_only = (flags & SCSFM_DEBUG_KERNEL_ONLY) != 0;  // profiling: consts are in place already
  // a launch with a coarser scale's maps in it runs the kernels instantiated for the index map (DepthMap)
  bool full_res = true;
  for (int i = 0; i < n; ++i) full_res = full_res && d[i].depth_shift == 0;
  dim3 grid;
  if (spec) {
    const size_t npx = (size_t)B * H * W;
    if (!kernel_only)
      hipLaunchKernelGGL((pairs_zero_prep_kernel<T>), dim3(1024 + 1, n), dim3(kThreads), 0, stream, pb, npx, n, B, K);
    const T r_hint = w_photo != 0.0 ? T(3.0 * w_geom / w_photo) : T(0);
    // reciprocal edge counts of get_smooth_loss's two means (loss_functions.py:150-152), as smooth_fwd_kernel rounds them
    const T sm_icx = T(1.0 / ((double)B * H * (W - 1))), sm_icy = T(1.0 / ((double)B * (H - 1) * W));
    const bool timed = g_profile.used < g_profile.n;
    if (timed) (void)hipEventRecord(g_profile.start[g_profile.used], stream);
    if (!spec_uses_march()) {
      grid = dim3(ceil_div(W, kTileW - 2), ceil_div(H, Tile<T>::kH - 2), n * B);
#define SCSFM_LAUNCH_SPEC(...)                                                                                          \
  hipLaunchKernelGGL((pair_fwd_spec_kernel<T, __VA_ARGS__>), grid, dim3(kThreads), debug_extra_lds(), stream, pb, B, H, W, \
                     flags & ~SCSFM_DEBUG_KERNEL_ONLY, r_hint, hint, sm_icx, sm_icy)
      if (!full_res && (flags & SCSFM_WITH_SSIM)) SCSFM_LAUNCH_SPEC(true, kRuntimeFlags, true);
      else if (!full_res) SCSFM_LAUNCH_SPEC(false, kRuntimeFlags, true);
#ifdef SCSFM_WITH_MARCH
      else if (spec_stages_fwd() && sizeof(T) == 4 && (flags & ~SCSFM_DEBUG_KERNEL_ONLY) == kTrainFlags) SCSFM_LAUNCH_SPEC(true, kTrainFlags, false, true);
      else if (spec_stages_fwd() && (flags & SCSFM_WITH_SSIM)) SCSFM_LAUNCH_SPEC(true, kRuntimeFlags, false, true);
#endif
      else if (sizeof(T) == 4 && (flags & ~SCSFM_DEBUG_KERNEL_ONLY) == kTrainFlags) SCSFM_LAUNCH_SPEC(true, kTrainFlags);
      else if (flags & SCSFM_WITH_SSIM) SCSFM_LAUNCH_SPEC(true);
      else SCSFM_LAUNCH_SPEC(false);
#undef SCSFM_LAUNCH_SPEC
    } else {
#ifdef SCSFM_WITH_MARCH
      const int nbands = ceil_div(W, kBandOut);
      const int rows = march_seg_rows(H, March<T>::kStrip * March<T>::kWaves, nbands * n * B);
      grid = dim3(nbands, ceil_div(H, rows), n * B);
#define SCSFM_LAUNCH_SPEC(...)                                                                                          \
  hipLaunchKernelGGL((pair_march_kernel<T, __VA_ARGS__>), grid, dim3(March<T>::kWaves * kWave), debug_extra_lds(), stream, pb, B, H, W, \
                     flags & ~SCSFM_DEBUG_KERNEL_ONLY, r_hint, rows, hint)
      if (!full_res && (flags & SCSFM_WITH_SSIM)) SCSFM_LAUNCH_SPEC(true, kRuntimeFlags, true);
      else if (!full_res) SCSFM_LAUNCH_SPEC(false, kRuntimeFlags, true);
      else if (sizeof(T) == 4 && (flags & ~SCSFM_DEBUG_KERNEL_ONLY) == kTrainFlags) SCSFM_LAUNCH_SPEC(true, kTrainFlags);
      else if (flags & SCSFM_WITH_SSIM) SCSFM_LAUNCH_SPEC(true);
      else SCSFM_LAUNCH_SPEC(false);
#undef SCSFM_LAUNCH_SPEC
#endif
    }
    if (timed) (void)hipEventRecord(g_profile.stop[g_profile.used++], stream);
  } else {
    hipLaunchKernelGGL((pairs_prep_kernel<T>), dim3(ceil_div(n * B, 64)), dim3(64), 0, stream, pb, n, B, K);
    grid = dim3(ceil_div(W, kTileW), ceil_div(H, Tile<T>::kH), n * B);
    if (flags & SCSFM_WITH_SSIM) {
      if (full_res) hipLaunchKernelGGL((pair_fwd_kernel<T, true, false>), grid, dim3(kThreads), 0, stream, pb, B, H, W, flags);
      else hipLaunchKernelGGL((pair_fwd_kernel<T, true, true>), grid, dim3(kThreads), 0, stream, pb, B, H, W, flags);
    } else {
      if (full_res) hipLaunchKernelGGL((pair_fwd_kernel<T, false, false>), grid, dim3(kThreads), 0, stream, pb, B, H, W, flags);
      else hipLaunchKernelGGL((pair_fwd_kernel<T, false, true>), grid, dim3(kThreads), 0, stream, pb, B, H, W, flags);
    }
  }
  bool any_smooth = false;
  for (int i = 0; i < n; ++i) any_smooth = any_smooth || pb.p[i].sm_partials != nullptr;
  if (!kernel_only)
    hipLaunchKernelGGL((pair_finalize_kernel<T>), dim3(any_smooth ? 2 * n : n), dim3(kThreads), 0, stream, pb, n, (int)(grid.x * grid.y * B),
                       (int)(grid.x * grid.y), spec ? 1.0 : 0.0, spec ? w_photo : 0.0, spec ? w_geom : 0.0, total, first ? 1 : 0,
                       spec ? hint : nullptr, st, H, W);
  return launch_status();
}

template <typename T>
static int pairs_fwd(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags, double w_photo,
                     double w_geom, void* stream_, T* step_out = nullptr, double w_smooth = 0.0) {
  clear_status();
  if (n < 0 || (n > 0 && !d) || B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !K) return SCSFM_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!desc_inputs_ok(d[i], H, W) || !d[i].out) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  // maximal runs of descriptors with the same mode (speculative or plain), at most kMaxPairs each
  const double* hint = n > 0 ? (const double*)d[0].hint : nullptr;
  const bool may_spec = w_photo != 0.0 || hint != nullptr;
  // the smooth loss rides in the speculative tile only, on full-resolution maps (scsfm_pair_desc::smooth_ws); the
  // experimental forwards of tuning builds do not carry it
  for (int i = 0; i < n; ++i)
    if (d[i].smooth_ws && (!(d[i].gbuf && may_spec) || d[i].depth_shift != 0 || spec_uses_march() || spec_stages_fwd()))
      return SCSFM_ERR_ARG;
  StepTotal<T> st;
  st.smooth_total = n > 0 ? (T*)d[0].smooth_total : nullptr;
  st.out = step_out; st.w1 = T(w_photo); st.w2 = T(w_smooth); st.w3 = T(w_geom);
  if (step_out && (n <= 0 || n > kMaxPairs || !d[0].total || !d[0].smooth_total)) return SCSFM_ERR_ARG;
  if (step_out)  // (one finalize launch must see every pair: a single run of one mode)
    for (int i = 1; i < n; ++i)
      if ((d[i].gbuf != nullptr && may_spec) != (d[0].gbuf != nullptr && may_spec)) return SCSFM_ERR_ARG;
  int i = 0;
  while (i < n) {
    const bool spec = d[i].gbuf != nullptr && may_spec;
    int j = i + 1;
    while (j < n && j - i < kMaxPairs && ((d[j].gbuf != nullptr && may_spec) == spec)) ++j;
    int rc = pairs_fwd_chunk<T>(j - i, d + i, B, H, W, K, flags, spec, w_photo, w_geom, (T*)d[0].total, i == 0, hint, stream, st);
    if (rc) return rc;
    i = j;
  }
  return SCSFM_OK;
}

// `sm` (scsfm_pairs_bwd_smooth): the smooth loss's gradient of n_frames depth maps is added by the same combining pass
// that stores them -- frame j's gradient buffer must be one of the buffers the descriptors name (g_tgt_depth /
// g_ref_depth at depth_shift 0).
struct SmoothFrames {
  int n = 0;
  void* const* grads = nullptr;       // [n] depth-gradient buffers
  void* const* edges = nullptr;       // [n] edge planes left by scsfm_smooth_multi_fwd
  void* const* per_img = nullptr;     // [n] {den_b, L_b} records = the start of each frame's smooth workspace
  const void* g_smooth = nullptr;     // 1 element
};
template <typename T>
static int pairs_bwd(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags, void* scratch,
                     const T* g_photo, const T* g_geom, bool accumulate, void* stream_, const SmoothFrames& sm = SmoothFrames()) {
  clear_status();
  if (n < 0 || (n > 0 && !d) || B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !K || !g_photo || !g_geom) return SCSFM_ERR_ARG;
  if (sm.n < 0 || sm.n > 2 * kMaxPairs || (sm.n > 0 && (!sm.grads || !sm.edges || !sm.per_img || !sm.g_smooth || (flags & SCSFM_DEBUG_SKIP_GEOM))))
    return SCSFM_ERR_ARG;
  for (int j = 0; j < sm.n; ++j) {  // every frame's buffer must be a full-resolution destination of this call
    if (!sm.grads[j] || !sm.edges[j] || !sm.per_img[j]) return SCSFM_ERR_ARG;
    bool found = false;
    for (int i = 0; i < n && !found; ++i)
      found = d[i].depth_shift == 0 && (d[i].g_tgt_depth == sm.grads[j] || d[i].g_ref_depth == sm.grads[j]);
    if (!found) return SCSFM_ERR_ARG;
  }
  bool sm_done[2 * kMaxPairs] = {};
  for (int i = 0; i < n; ++i)
    if (!desc_inputs_ok(d[i], H, W) || !d[i].g_tgt_depth || !d[i].g_ref_depth || !d[i].g_pose || (!d[i].gbuf && !scratch))
      return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const size_t npx = (size_t)B * H * W;
  // store mode: the depth-gradient buffers this call has written so far (later chunks add to them)
  constexpr int kSeenMax = 256;
  const void* seen[kSeenMax];
  int nseen = 0;
  if (!accumulate && 2 * n > kSeenMax) return SCSFM_ERR_ARG;
  for (int i0 = 0; i0 < n; i0 += kMaxPairs) {
    const int m = n - i0 < kMaxPairs ? n - i0 : kMaxPairs;
    PairBatch<T> pb;
    bool full_res = true;
    for (int i = 0; i < m; ++i) {
      pb.p[i] = make_pair_args<T>(d[i0 + i], B, H, W, scratch, i0 + i);
      full_res = full_res && d[i0 + i].depth_shift == 0;
    }
    const int nax = ceil_div(W, kTileW - 2), nay = ceil_div(H, Tile<T>::kH - 2);
    const int nbx = ceil_div(W, kWave), nby = ceil_div(H, kGeomRows * (kThreads / kWave));
    if (!(flags & SCSFM_DEBUG_SKIP_PHOTO)) {
      const int g = nax * nay * m * B < kPersistentGrid ? nax * nay * m * B : kPersistentGrid;
#define SCSFM_LAUNCH_PHOTO(SSIM, SCALED)                                                                                   \
  hipLaunchKernelGGL((pair_bwd_photo_kernel<T, SSIM, SCALED>), dim3(g), dim3(kThreads), 0, stream, pb, nax, nay, m * B, B, H, \
                     W, flags, g_photo, g_geom)
      if (flags & SCSFM_WITH_SSIM) {
        if (full_res) SCSFM_LAUNCH_PHOTO(true, false); else SCSFM_LAUNCH_PHOTO(true, true);
      } else {
        if (full_res) SCSFM_LAUNCH_PHOTO(false, false); else SCSFM_LAUNCH_PHOTO(false, true);
      }
#undef SCSFM_LAUNCH_PHOTO
    }
    if (!(flags & SCSFM_DEBUG_SKIP_GEOM)) {
      const int g = nbx * nby * m * B < kPersistentGrid ? nbx * nby * m * B : kPersistentGrid;
      if (full_res)
        hipLaunchKernelGGL((pair_bwd_geom_kernel<T, false>), dim3(g), dim3(kThreads), 0, stream, pb, nbx, nby, m * B, B, H, W,
                           flags, g_photo, g_geom);
      else
        hipLaunchKernelGGL((pair_bwd_geom_kernel<T, true>), dim3(g), dim3(kThreads), 0, stream, pb, nbx, nby, m * B, B, H, W,
                           flags, g_photo, g_geom);
    }
    if (flags & SCSFM_DEBUG_SKIP_GEOM) {
      hipLaunchKernelGGL((pairs_pose_reduce_kernel<T>), dim3(m * B), dim3(kWave), 0, stream, pb, B, nbx * nby, K, g_photo,
                         g_geom, (double*)d[0].hint);
    } else {
      // group the private planes by the caller's destination buffer: a pair's dense plane belongs to its
      // target depth map, its scatter plane to its reference depth map
      CombineBatch<T> cb;
      cb.nd = 0; cb.nsrc = 0;
      cb.g_smooth = (const T*)sm.g_smooth;
      for (int k = 0; k < 2 * kMaxPairs; ++k) { cb.sm_edge[k] = nullptr; cb.sm_img[k] = nullptr; }
      for (int i = 0; i < m; ++i) {
        for (int which = 0; which < 2; ++which) {
          T* dst = (T*)(which == 0 ? d[i0 + i].g_tgt_depth : d[i0 + i].g_ref_depth);
          int k = 0;
          while (k < cb.nd && cb.dst[k] != dst) ++k;
          if (k == cb.nd) {
            cb.dst[k] = dst;
            cb.ds[k] = d[i0 + i].depth_shift;
            int q = 0;
            while (q < nseen && seen[q] != dst) ++q;
            cb.store[k] = (!accumulate && q == nseen && nseen < kSeenMax) ? 1 : 0;
            if (q == nseen && nseen < kSeenMax) seen[nseen++] = dst;
            for (int j = 0; j < sm.n; ++j)  // the smooth term joins the first combining pass that writes this buffer
              if (sm.grads[j] == (void*)dst && !sm_done[j] && d[i0 + i].depth_shift == 0) {
                cb.sm_edge[k] = (const T*)sm.edges[j]; cb.sm_img[k] = (const double*)sm.per_img[j];
                sm_done[j] = true;
                break;
              }
            ++cb.nd;
          }
          CombineSrc<T>& sc = cb.src[cb.nsrc++];
          sc.plane = pb.p[i].gbuf + (which == 0 ? kPlaneDense : kPlaneScatter) * npx;
          sc.sums = pb.p[i].sums;
          sc.dst = k;
        }
      }
      {  // the sources of one destination side by side (stable: pair order is kept within a destination)
        CombineSrc<T> sorted[2 * kMaxPairs];
        int ns = 0;
        for (int k = 0; k < cb.nd; ++k) {
          cb.first[k] = ns;
          for (int q = 0; q < cb.nsrc; ++q)
            if (cb.src[q].dst == k) sorted[ns++] = cb.src[q];
          cb.count[k] = ns - cb.first[k];
        }
        for (int q = 0; q < cb.nsrc; ++q) cb.src[q] = sorted[q];
        for (int k = cb.nd; k < 2 * kMaxPairs; ++k) { cb.first[k] = 0; cb.count[k] = 0; }
      }
      // one image of the batch per grid z (8 elements per thread); row 0 of y: dL/dpose, on z = 0
      const size_t plane = (size_t)H * W;
      int gx = (int)((plane + 8 * kThreads - 1) / (8 * kThreads));
      const int gpose = ceil_div(m * B, kThreads / kWave);
      gx = gx < gpose ? gpose : gx;
      hipLaunchKernelGGL((pairs_combine_kernel<T>), dim3(gx, cb.nd + 1, B), dim3(kThreads), 0, stream, cb, npx, pb, m, B,
                         nbx * nby, K, g_photo, g_geom, (double*)d[0].hint);
      if (!full_res)
        hipLaunchKernelGGL((pairs_combine_pooled_kernel<T>), dim3((int)((npx + 8 * kThreads - 1) / (8 * kThreads)) / 4 + 1, cb.nd), dim3(kThreads), 0, stream, cb, npx, W,
                           g_photo, g_geom);
    }
  }
  return launch_status();
}

// Gradients of the DATA inputs of the pair losses (images, intrinsics): the reference's autograd reaches them
// (inverse_warp.py:253-262, loss_functions.py:99-108) although train.py never asks.  After pairs_bwd, same arguments.
template <typename T>
static int pairs_bwd_inputs(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags,
                            const T* g_photo, const T* g_geom, T* g_K, void* stream_) {
  clear_status();
  if (n < 0 || (n > 0 && !d) || B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !K || !g_photo || !g_geom) return SCSFM_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!desc_inputs_ok(d[i], H, W)) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  for (int i0 = 0; i0 < n; i0 += kMaxPairs) {
    const int m = n - i0 < kMaxPairs ? n - i0 : kMaxPairs;
    PairBatch<T> pb;
    ImageGrads<T> ig;
    bool full_res = true, any_img = false;
    for (int i = 0; i < kMaxPairs; ++i) { ig.tgt[i] = nullptr; ig.ref[i] = nullptr; }
    for (int i = 0; i < m; ++i) {
      pb.p[i] = make_pair_args<T>(d[i0 + i], B, H, W, nullptr, i0 + i);
      full_res = full_res && d[i0 + i].depth_shift == 0;
      ig.tgt[i] = (T*)d[i0 + i].g_tgt_img; ig.ref[i] = (T*)d[i0 + i].g_ref_img;
      any_img = any_img || ig.tgt[i] || ig.ref[i];
    }
    if (any_img) {
      const dim3 grid(ceil_div(W, kTileW - 2), ceil_div(H, Tile<T>::kH - 2), m * B);
#define SCSFM_LAUNCH_IMAGES(SSIM, SCALED)                                                                             \
  hipLaunchKernelGGL((pair_bwd_images_kernel<T, SSIM, SCALED>), grid, dim3(kThreads), 0, stream, pb, ig, B, H, W, flags, \
                     g_photo, g_geom)
      if (flags & SCSFM_WITH_SSIM) {
        if (full_res) SCSFM_LAUNCH_IMAGES(true, false); else SCSFM_LAUNCH_IMAGES(true, true);
      } else {
        if (full_res) SCSFM_LAUNCH_IMAGES(false, false); else SCSFM_LAUNCH_IMAGES(false, true);
      }
#undef SCSFM_LAUNCH_IMAGES
    }
    if (g_K) {
      const int nbx = ceil_div(W, kWave), nby = ceil_div(H, kGeomRows * (kThreads / kWave));
      hipLaunchKernelGGL((pairs_intrinsics_kernel<T>), dim3(B), dim3(kWave), 0, stream, pb, m, B, nbx * nby, K, g_photo,
                         g_geom, g_K, i0 > 0 ? 1 : 0);
    }
  }
  return launch_status();
}

template <typename T>
static int pair_refinalize(int B, int H, int W, void* ws, T* out, void* stream) {
  clear_status();
  if (B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || !ws || !out) return SCSFM_ERR_ARG;
  const PairWs l = pair_ws_layout(B, H, W);
  double* sums = reinterpret_cast<double*>(reinterpret_cast<char*>(ws) + l.off_sums);
  hipLaunchKernelGGL((pair_refinalize_kernel<T>), dim3(1), dim3(kWave), 0, (hipStream_t)stream, sums, out);
  return launch_status();
}

template <typename T>
static scsfm_pair_desc one_desc(const T* tgt_img, const T* ref_img, const T* tgt_depth, const T* ref_depth,
                                const T* pose, void* ws, void* out, void* g_tgt, void* g_ref, void* g_pose, void* gbuf) {
  scsfm_pair_desc d;
  d.tgt_img = tgt_img; d.ref_img = ref_img; d.tgt_depth = tgt_depth; d.ref_depth = ref_depth; d.pose = pose;
  d.ws = ws; d.out = out; d.g_tgt_depth = g_tgt; d.g_ref_depth = g_ref; d.g_pose = g_pose; d.gbuf = gbuf;
  d.total = nullptr;
  d.hint = nullptr;
  d.depth_shift = 0;
  d.g_tgt_img = nullptr; d.g_ref_img = nullptr;
  d.smooth_ws = nullptr; d.smooth_edge = nullptr; d.smooth_out = nullptr; d.smooth_total = nullptr;
  return d;
}

}  // namespace scsfm

extern "C" {

int scsfm_profile_begin(int n) {
  scsfm::profile_release();
  if (n <= 0) return n == 0 ? SCSFM_OK : SCSFM_ERR_ARG;
  scsfm::g_profile.start = new hipEvent_t[n];
  scsfm::g_profile.stop = new hipEvent_t[n];
  for (int i = 0; i < n; ++i) {
    if (hipEventCreate(&scsfm::g_profile.start[i]) != hipSuccess || hipEventCreate(&scsfm::g_profile.stop[i]) != hipSuccess) {
      scsfm::g_profile.n = i;  // (what exists so far is released)
      scsfm::profile_release();
      return (int)hipGetLastError();
    }
    scsfm::g_profile.n = i + 1;
  }
  return SCSFM_OK;
}

int scsfm_profile_end(double* mean_us, double* min_us, int* count) {
  if (!mean_us || !min_us || !count) return SCSFM_ERR_ARG;
  double sum = 0.0, mn = 0.0;
  const int used = scsfm::g_profile.used;
  for (int i = 0; i < used; ++i) {
    float ms = 0.0f;
    (void)hipEventSynchronize(scsfm::g_profile.stop[i]);
    (void)hipEventElapsedTime(&ms, scsfm::g_profile.start[i], scsfm::g_profile.stop[i]);
    sum += 1e3 * ms;
    mn = (i == 0 || 1e3 * ms < mn) ? 1e3 * ms : mn;
  }
  *mean_us = used ? sum / used : 0.0;
  *min_us = mn;
  *count = used;
  scsfm::profile_release();
  return SCSFM_OK;
}

size_t scsfm_pair_bwd_scratch_bytes(int B, int H, int W) {
  if (B <= 0 || H < 2 || W < 2) return 0;
  // six planes per pair-direction (dL/dI_w x3, dL/d diff_depth, dense dL/d tgt_depth, scattered dL/d ref_depth);
  // sized for fp64
  return (((size_t)scsfm::kNumPlanes * B * H * W * sizeof(double)) + 255) & ~(size_t)255;
}

size_t scsfm_pair_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H < 2 || W < 2) return 0;
  return scsfm::pair_ws_layout(B, H, W).total;  // sized for both precisions and both forward kernels
}

#define SCSFM_PAIR_API(SUF, T)                                                                                        \
  int scsfm_pairs_fwd_##SUF(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags,          \
                            double w_photo, double w_geom, void* stream) {                                            \
    return scsfm::pairs_fwd<T>(n, d, B, H, W, K, flags, w_photo, w_geom, stream);                                     \
  }                                                                                                                   \
  int scsfm_pairs_fwd_step_##SUF(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags,     \
                                 double w_photo, double w_smooth, double w_geom, T* step_out, void* stream) {         \
    if (!step_out) return SCSFM_ERR_ARG;                                                                              \
    return scsfm::pairs_fwd<T>(n, d, B, H, W, K, flags, w_photo, w_geom, stream, step_out, w_smooth);                 \
  }                                                                                                                   \
  int scsfm_pairs_bwd_##SUF(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags,          \
                            void* scratch, const T* g_photo, const T* g_geom, void* stream) {                         \
    return scsfm::pairs_bwd<T>(n, d, B, H, W, K, flags, scratch, g_photo, g_geom, false, stream);                     \
  }                                                                                                                   \
  int scsfm_pairs_bwd_smooth_##SUF(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags,   \
                                   void* scratch, const T* g_photo, const T* g_geom, int n_frames,                    \
                                   void* const* frame_grads, void* const* frame_edges, void* const* frame_stats,      \
                                   const T* g_smooth, void* stream) {                                                 \
    scsfm::SmoothFrames sm;                                                                                           \
    sm.n = n_frames; sm.grads = frame_grads; sm.edges = frame_edges; sm.per_img = frame_stats; sm.g_smooth = g_smooth; \
    return scsfm::pairs_bwd<T>(n, d, B, H, W, K, flags, scratch, g_photo, g_geom, false, stream, sm);                 \
  }                                                                                                                   \
  int scsfm_pairs_bwd_inputs_##SUF(int n, const scsfm_pair_desc* d, int B, int H, int W, const T* K, unsigned flags,   \
                                   const T* g_photo, const T* g_geom, T* g_intrinsics, void* stream) {                \
    return scsfm::pairs_bwd_inputs<T>(n, d, B, H, W, K, flags, g_photo, g_geom, g_intrinsics, stream);                \
  }                                                                                                                   \
  int scsfm_pair_fwd_##SUF(int B, int H, int W, const T* tgt_img, const T* ref_img, const T* tgt_depth,               \
                           const T* ref_depth, const T* pose, const T* K, unsigned flags, void* ws, T* out,           \
                           void* stream) {                                                                            \
    scsfm_pair_desc d = scsfm::one_desc<T>(tgt_img, ref_img, tgt_depth, ref_depth, pose, ws, out, 0, 0, 0, 0);         \
    return scsfm::pairs_fwd<T>(1, &d, B, H, W, K, flags, 0.0, 0.0, stream);                                           \
  }                                                                                                                   \
  int scsfm_pair_fwd_spec_##SUF(int B, int H, int W, const T* tgt_img, const T* ref_img, const T* tgt_depth,          \
                                const T* ref_depth, const T* pose, const T* K, unsigned flags, void* ws, void* gbuf,  \
                                double w_photo, double w_geom, T* out, void* stream) {                                \
    if (!gbuf || w_photo == 0.0) return SCSFM_ERR_ARG;                                                                \
    scsfm_pair_desc d = scsfm::one_desc<T>(tgt_img, ref_img, tgt_depth, ref_depth, pose, ws, out, 0, 0, 0, gbuf);      \
    return scsfm::pairs_fwd<T>(1, &d, B, H, W, K, flags, w_photo, w_geom, stream);                                    \
  }                                                                                                                   \
  int scsfm_pair_bwd_##SUF(int B, int H, int W, const T* tgt_img, const T* ref_img, const T* tgt_depth,               \
                           const T* ref_depth, const T* pose, const T* K, unsigned flags, void* ws, void* scratch,    \
                           const T* g_photo, const T* g_geom, T* g_tgt_depth, T* g_ref_depth, T* g_pose,              \
                           void* stream) {                                                                            \
    if (!scratch) return SCSFM_ERR_ARG;                                                                               \
    scsfm_pair_desc d = scsfm::one_desc<T>(tgt_img, ref_img, tgt_depth, ref_depth, pose, ws, 0, g_tgt_depth,           \
                                           g_ref_depth, g_pose, scratch);                                             \
    return scsfm::pairs_bwd<T>(1, &d, B, H, W, K, flags, scratch, g_photo, g_geom, true, stream);                     \
  }                                                                                                                   \
  int scsfm_pair_refinalize_##SUF(int B, int H, int W, void* ws, T* out, void* stream) {                              \
    return scsfm::pair_refinalize<T>(B, H, W, ws, out, stream);                                                       \
  }

#ifdef PROBE_TIMING
int scsfm_probe_read(unsigned long long* host, size_t n) {
  (void)hipDeviceSynchronize();
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(scsfm::g_probe), n * sizeof(unsigned long long));
}
#endif

SCSFM_PAIR_API(f32, float)
SCSFM_PAIR_API(f64, double)

}  // extern "C"
