// Edge-aware smoothness of the mean-normalised depth (get_smooth_loss, loss_functions.py:133-152):
//     d = D / (mean_HW(D) + 1e-7);  loss = mean(|dx d| * exp(-mean_c |dx I|)) + mean(|dy d| * exp(-mean_c |dy I|))
// Because the per-image normaliser is positive, |dx d| = |dx D| / den, so one pass over D and I
// yields, per image, sum(D) and the two weighted edge sums; the loss and the mean's contribution
// to the gradient follow from those three numbers:
//     loss   = sum_b L_b / den_b,        L_b = Sx_b / cnt_x + Sy_b / cnt_y
//     dL/dD  = g * [ (1/den_b) * dL_b/dD(p)  -  L_b / (den_b^2 * H * W) ]
//
// All frames of compute_smooth_loss (target + every reference, loss_functions.py:154-159) run in one
// launch per stage (blockIdx.z = frame * B + b).  A thread walks a 4-row column strip: the row below
// is loaded once and becomes the next row's centre, the right neighbour comes from the next lane by
// shuffle (lane 63 loads it), and every edge weight -- one exp each -- is evaluated exactly once in
// the forward and once per direction in the backward (the left / upper edge of a pixel is the
// previous lane's / row's right / lower edge).
#include "scsfm_common.h"
#include "scsfm_smooth_math.h"

namespace scsfm {

#ifndef SCSFM_SMOOTH_ROWS  // tuning knob (tools/build_variants.sh)
#define SCSFM_SMOOTH_ROWS 4
#endif
constexpr int kSmRows = SCSFM_SMOOTH_ROWS;  // rows per thread (backward)
#ifndef SCSFM_SMOOTH_FWD_ROWS  // tuning knob
#define SCSFM_SMOOTH_FWD_ROWS 8
#endif
constexpr int kSmFwdRows = SCSFM_SMOOTH_FWD_ROWS;  // rows per thread of the forward
constexpr int kSmFwdCols = kWave - 2;        // owned columns per wave of the forward (lanes 1 .. 62; 0 and 63 are halo)
constexpr int kMaxFrames = 8;

struct SmoothWs {
  size_t off_img, off_partials, off_counter, total;
  int nbx, nby;    // tiling of the backward (64 x 4 kSmRows pixels per workgroup)
  int fbx, fby;    // tiling of the forward (62 x 4 kSmFwdRows): the partial sums it leaves
};
inline SmoothWs smooth_ws_layout(int B, int H, int W) {
  SmoothWs l;
  l.nbx = ceil_div(W, kWave);
  l.nby = ceil_div(H, kSmRows * (kThreads / kWave));
  l.fbx = ceil_div(W, kSmFwdCols);
  l.fby = ceil_div(H, kSmFwdRows * (kThreads / kWave));
  l.off_img = 0;                                   // double[B][2] = {den_b, L_b}
  l.off_partials = (size_t)B * 2 * sizeof(double); // double[B][nby*nbx][3]
  // (+ 256 bytes whose first word is the "finalize blocks done" counter of a multi-frame call)
  l.off_counter = (l.off_partials + (size_t)B * l.fbx * l.fby * 3 * sizeof(double) + 255) & ~(size_t)255;
  l.total = l.off_counter + 256;
  return l;
}

template <typename T>
struct SmoothFrame {
  const T* depth; const T* img; double* per_img; double* partials; T* out; T* g_depth;
  T* edge;  // optional plane [B,H,W]: sum over the pixel's four edges of +-sgn(dD) w / cnt, written by the
            // forward so that the backward is a pure stream (no image reads, no exp)
};
template <typename T>
struct SmoothBatch {
  SmoothFrame<T> f[kMaxFrames];
  // optional sum over the frames of a call, finished by whichever finalize block comes last (no extra launch):
  unsigned* counter;  // in the first frame's workspace; zeroed by the forward kernel
  T* total;           // nullptr: not wanted
  int first;          // 1: store (first launch of the call), 0: add
  // scsfm_smooth_multi_fwd_step: the block that finishes the frames' total also forms the step's objective
  // (train.py:268) from it and the pair losses the caller computed before -- step_out[4] = {w1 photo + w2 smooth + w3
  // geometry, photo, smooth, geometry} -- instead of a launch of its own (scsfm_step_total)
  const T* step_pg;   // {photo, geometry} (device), or nullptr
  T* step_out;
  T w1, w2, w3;
};

template <typename T>
__device__ __forceinline__ Px<T> load_px(const T* __restrict__ depth, const T* __restrict__ img, unsigned plane, unsigned p) {
  Px<T> r;
  const unsigned off = p * unsigned(sizeof(T));
  r.d = ld_at(depth, off); r.c0 = ld_at(img, off); r.c1 = ld_at(img + plane, off); r.c2 = ld_at(img + 2 * plane, off);
  return r;
}
template <typename T>
__device__ __forceinline__ Px<T> shfl_down_px(const Px<T>& v) {
  Px<T> r;
  r.d = __shfl_down(v.d, 1); r.c0 = __shfl_down(v.c0, 1); r.c1 = __shfl_down(v.c1, 1); r.c2 = __shfl_down(v.c2, 1);
  return r;
}
// Forward.  A wave covers 64 columns of which it owns 62 (lanes 0 and 63 are halo: they only supply the left / right
// neighbour) and a thread walks a column strip of kSmFwdRows rows plus the row above and the row below it, so every
// pixel value is loaded ~1.3 times (round 1: 2.5 times -- the right neighbours were loaded instead of taken from
// the adjacent lane, strips were 4 rows) and no load sits behind a divergent branch.  Horizontal neighbours come from
// the adjacent lanes through DPP.
template <typename T>
__global__ __launch_bounds__(kThreads) void smooth_fwd_kernel(SmoothBatch<T> sb, int B, int H, int W) {
  __shared__ double red[3 * (kThreads / kWave)];
  const int frame = blockIdx.z / B, b = blockIdx.z - frame * B;
  const SmoothFrame<T>& fr = sb.f[frame];
  const int lane = threadIdx.x & (kWave - 1);
  const int x = blockIdx.x * kSmFwdCols - 1 + lane;
  const int y0 = (blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave) * kSmFwdRows;
  const unsigned plane = unsigned(H) * unsigned(W);
  const T* __restrict__ depth = fr.depth + (size_t)b * plane;
  const T* __restrict__ img = fr.img + (size_t)b * 3 * plane;
  T* __restrict__ edge = fr.edge ? fr.edge + (size_t)b * plane : nullptr;
  const bool own_x = lane >= 1 && lane <= kSmFwdCols && x < W;
  const bool has_right = x >= 0 && x + 1 < W;  // the edge (x, x + 1) exists (also evaluated by the left halo lane)
  const int xc = x < 0 ? 0 : (x < W ? x : W - 1);
  const T icx = T(1.0 / ((double)B * H * (W - 1))), icy = T(1.0 / ((double)B * (H - 1) * W));
  T sd = T(0), sx = T(0), sy = T(0);
  // every load of the strip first: rows y0 - 1 .. y0 + kSmFwdRows (clamped into the image)
  Px<T> row[kSmFwdRows + 2];
#pragma unroll
  for (int r = 0; r < kSmFwdRows + 2; ++r) {
    const int y = y0 - 1 + r;
    row[r] = load_px(depth, img, plane, unsigned(y < 0 ? 0 : (y < H ? y : H - 1)) * unsigned(W) + unsigned(xc));
  }
  // lower edge of the row above the strip (only the stored gradient terms need it)
  T ty_prev = (y0 > 0 && y0 < H) ? t_sgn(row[0].d - row[1].d) * edge_weight(row[0], row[1]) * icy : T(0);
#pragma unroll
  for (int r = 0; r < kSmFwdRows; ++r) {
    const int y = y0 + r;
    const Px<T> cur = row[r + 1], down = row[r + 2], right = lane_right_px(cur);
    const bool ex = has_right && y < H, ey = y < H && y + 1 < H;
    const T wx = ex ? edge_weight(cur, right) : T(0), wy = ey ? edge_weight(cur, down) : T(0);
    const T dx = cur.d - right.d, dy = cur.d - down.d;
    if (own_x && y < H) { sd += cur.d; sx += t_abs(dx) * wx; sy += t_abs(dy) * wy; }
    if (edge) {  // workgroup-uniform
      const T tx = t_sgn(dx) * wx * icx, ty = t_sgn(dy) * wy * icy;
      const T tx_left = lane_left(tx);  // the pixel's left edge is its left neighbour's right edge
      if (own_x && y < H) st_at(edge, (unsigned(y) * unsigned(W) + unsigned(x)) * unsigned(sizeof(T)), tx - tx_left + ty - ty_prev);
      ty_prev = ty;
    }
  }
  if (sb.total && blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && threadIdx.x == 0) *sb.counter = 0u;
  T v[3] = {sd, sx, sy};
  block_sum<3>(v, red);
  if (threadIdx.x == 0) {
    double* o = fr.partials + 3 * ((size_t)(b * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
    o[0] = double(v[0]); o[1] = double(v[1]); o[2] = double(v[2]);
  }
}

// One block per frame; each wave reduces whole images (b = wave, wave + 4, ...) with shuffles only,
// then the per-wave loss contributions meet in LDS.
template <typename T>
__global__ __launch_bounds__(kThreads) void smooth_finalize_kernel(SmoothBatch<T> sb, int B, int H, int W, int nblk) {
  __shared__ double red[kThreads / kWave];
  const SmoothFrame<T>& fr = sb.f[blockIdx.x];
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const double cnt_x = (double)B * H * (W - 1), cnt_y = (double)B * (H - 1) * W;
  double loss = 0.0;
  for (int b = wave; b < B; b += kThreads / kWave) {
    double v0 = 0, v1 = 0, v2 = 0;
    // (two records per lane in flight: the 112 records of a 256 x 832 image are one round trip per image, not two)
    for (int i = lane; i < nblk; i += 2 * kWave) {
      const bool two = i + kWave < nblk;
      const double* q = fr.partials + 3 * ((size_t)b * nblk + i);
      const double* r = fr.partials + 3 * ((size_t)b * nblk + (two ? i + kWave : i));
      const double a0 = q[0], a1 = q[1], a2 = q[2], b0 = r[0], b1 = r[1], b2 = r[2];
      v0 += a0; v1 += a1; v2 += a2;
      if (two) { v0 += b0; v1 += b1; v2 += b2; }
    }
    v0 = wave_sum(v0); v1 = wave_sum(v1); v2 = wave_sum(v2);
    const double den = v0 / ((double)H * W) + 1e-7;  // mean_HW(D) + 1e-7, loss_functions.py:139-140
    const double L = v1 / cnt_x + v2 / cnt_y;
    if (lane == 0) { fr.per_img[2 * b] = den; fr.per_img[2 * b + 1] = L; }
    loss += L / den;
  }
  if (lane == 0) red[wave] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0;
    for (int w = 0; w < kThreads / kWave; ++w) t += red[w];
    fr.out[0] = T(t);
    if (sb.total) {  // the last block to get here adds the frames up, in frame order
      __threadfence();
      if (atomicAdd(sb.counter, 1u) == gridDim.x - 1) {
        __threadfence();
        T sum = T(0);
        for (unsigned i = 0; i < gridDim.x; ++i) sum += *const_cast<const volatile T*>(sb.f[i].out);
        sb.total[0] = sb.first ? sum : sb.total[0] + sum;
        if (sb.step_pg) {
          const T photo = sb.step_pg[0], geom = sb.step_pg[1], smooth = sb.total[0];
          sb.step_out[0] = sb.w1 * photo + sb.w2 * smooth + sb.w3 * geom;
          sb.step_out[1] = photo; sb.step_out[2] = smooth; sb.step_out[3] = geom;
        }
      }
    }
  }
}

// kAccumulate: g_depth += (the single-frame entry point's contract) instead of a plain store.
template <typename T, bool kAccumulate>
__global__ __launch_bounds__(kThreads) void smooth_bwd_kernel(SmoothBatch<T> sb, int B, int H, int W,
                                                              const T* __restrict__ g_loss) {
  const int frame = blockIdx.z / B, b = blockIdx.z - frame * B;
  const SmoothFrame<T>& fr = sb.f[frame];
  if (!fr.g_depth) return;  // this frame's gradient is not wanted (workgroup-uniform)
  const int lane = threadIdx.x & (kWave - 1);
  const int x = blockIdx.x * kWave + lane;
  const int y0 = (blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave) * kSmRows;
  const unsigned plane = unsigned(H) * unsigned(W);
  const T* __restrict__ depth = fr.depth + (size_t)b * plane;
  const T* __restrict__ img = fr.img + (size_t)b * 3 * plane;
  T* __restrict__ g_depth = fr.g_depth + (size_t)b * plane;
  const bool in_x = x < W;
  const int xc = in_x ? x : W - 1;
  const T g = g_loss[0];
  const T iden = T(1.0 / fr.per_img[2 * b]);
  const T icx = T(1.0 / ((double)B * H * (W - 1))), icy = T(1.0 / ((double)B * (H - 1) * W));
  const T mean_term = T(fr.per_img[2 * b + 1] / (fr.per_img[2 * b] * fr.per_img[2 * b] * (double)H * W));
  if (fr.edge) {  // the forward left the per-pixel edge terms: 4 B read + 4 B written per pixel, as a linear 16-byte stream
    const T* __restrict__ edge = fr.edge + (size_t)b * plane;
    constexpr int Q = 16 / sizeof(T);
    struct alignas(16) Quad { T v[Q]; };
    const unsigned nthreads = gridDim.x * gridDim.y * kThreads;
    const unsigned tid = (blockIdx.y * gridDim.x + blockIdx.x) * kThreads + threadIdx.x;
    const bool aligned = ((reinterpret_cast<size_t>(edge) | reinterpret_cast<size_t>(g_depth)) & 15) == 0;
    const unsigned nq = aligned ? plane / Q : 0;
    for (unsigned q = tid; q < nq; q += nthreads) {
      const Quad e = reinterpret_cast<const Quad*>(edge)[q];
      Quad o;
      if (kAccumulate) o = reinterpret_cast<const Quad*>(g_depth)[q];
#pragma unroll
      for (int j = 0; j < Q; ++j) o.v[j] = (kAccumulate ? o.v[j] : T(0)) + g * (e.v[j] * iden - mean_term);
      reinterpret_cast<Quad*>(g_depth)[q] = o;
    }
    for (unsigned i = nq * Q + tid; i < plane; i += nthreads) {
      const T v = g * (edge[i] * iden - mean_term);
      g_depth[i] = kAccumulate ? g_depth[i] + v : v;
    }
    return;
  }
  Px<T> cur = load_px(depth, img, plane, unsigned(y0 < H ? y0 : H - 1) * unsigned(W) + unsigned(xc));
  // lower edge of the row above the strip
  T ty_prev = T(0);
  if (y0 > 0 && y0 < H) {
    const Px<T> up = load_px(depth, img, plane, unsigned(y0 - 1) * unsigned(W) + unsigned(xc));
    ty_prev = t_sgn(up.d - cur.d) * edge_weight(up, cur) * icy;
  }
#pragma unroll
  for (int r = 0; r < kSmRows; ++r) {
    const int y = y0 + r;
    const unsigned p = unsigned(y < H ? y : H - 1) * unsigned(W) + unsigned(xc);
    Px<T> right = shfl_down_px(cur);
    if (lane == kWave - 1 && x + 1 < W) right = load_px(depth, img, plane, p + 1);
    const Px<T> down = load_px(depth, img, plane, unsigned(y + 1 < H ? y + 1 : H - 1) * unsigned(W) + unsigned(xc));
    // d/dD(p) of |D(p) - D(q)| w(p,q) / cnt : +sgn for the first pixel of an edge, -sgn for the second
    const T tx = (in_x && x + 1 < W) ? t_sgn(cur.d - right.d) * edge_weight(cur, right) * icx : T(0);
    const T ty = (y + 1 < H) ? t_sgn(cur.d - down.d) * edge_weight(cur, down) * icy : T(0);
    T tx_left = __shfl_up(tx, 1);
    if (lane == 0) {
      tx_left = T(0);
      if (x > 0 && in_x && y < H) {
        const Px<T> left = load_px(depth, img, plane, p - 1);
        tx_left = t_sgn(left.d - cur.d) * edge_weight(left, cur) * icx;
      }
    }
    if (in_x && y < H) {
      const T v = g * ((tx - tx_left + ty - ty_prev) * iden - mean_term);
      g_depth[p] = kAccumulate ? g_depth[p] + v : v;
    }
    ty_prev = ty;
    cur = down;
  }
}

// dL/d img of get_smooth_loss (the reference's autograd reaches the image through the edge weights; train.py never
// asks): with t(p, q) = |D(p) - D(q)| exp(-mean_c |I_c(p) - I_c(q)|) / (den_b cnt) for the edge (p, q),
//   dL/dI_c(p) = g * sum over the (up to four) edges of p of  -+ t / 3 * sgn(I_c(first) - I_c(second))
// (- for the edge's first pixel, + for its second; abs has the sub-gradient 0 at 0).  One thread per pixel, plain
// loads, stores (or accumulates): not a hot path.
template <typename T>
__global__ __launch_bounds__(kThreads) void smooth_bwd_images_kernel(SmoothBatch<T> sb, int B, int H, int W,
                                                                     const T* __restrict__ g_loss, int accumulate) {
  const int frame = blockIdx.z / B, b = blockIdx.z - frame * B;
  const SmoothFrame<T>& fr = sb.f[frame];
  if (!fr.g_depth) return;  // (g_depth carries this frame's IMAGE gradient buffer in this launch; workgroup-uniform)
  const int x = blockIdx.x * kWave + (threadIdx.x & (kWave - 1));
  const int y = blockIdx.y * (kThreads / kWave) + threadIdx.x / kWave;
  if (x >= W || y >= H) return;
  const unsigned plane = unsigned(H) * unsigned(W);
  const T* __restrict__ depth = fr.depth + (size_t)b * plane;
  const T* __restrict__ img = fr.img + (size_t)b * 3 * plane;
  T* __restrict__ g_img = fr.g_depth + (size_t)b * 3 * plane;
  const T third = T(1.0 / 3.0);
  const T sx = g_loss[0] * T(1.0 / (fr.per_img[2 * b] * (double)B * H * (W - 1))) * third;
  const T sy = g_loss[0] * T(1.0 / (fr.per_img[2 * b] * (double)B * (H - 1) * W)) * third;
  const unsigned p = unsigned(y) * unsigned(W) + unsigned(x);
  const Px<T> cur = load_px(depth, img, plane, p);
  T acc[3] = {T(0), T(0), T(0)};
  auto edge = [&](const Px<T>& first, const Px<T>& second, T scale, T side) {  // side: -1 = cur is `first`, +1 = `second`
    const T t = t_abs(first.d - second.d) * edge_weight(first, second) * scale * side;
    acc[0] += t * t_sgn(first.c0 - second.c0); acc[1] += t * t_sgn(first.c1 - second.c1); acc[2] += t * t_sgn(first.c2 - second.c2);
  };
  if (x + 1 < W) edge(cur, load_px(depth, img, plane, p + 1), sx, T(-1));
  if (x > 0) edge(load_px(depth, img, plane, p - 1), cur, sx, T(1));
  if (y + 1 < H) edge(cur, load_px(depth, img, plane, p + unsigned(W)), sy, T(-1));
  if (y > 0) edge(load_px(depth, img, plane, p - unsigned(W)), cur, sy, T(1));
#pragma unroll
  for (int c = 0; c < 3; ++c) g_img[c * plane + p] = accumulate ? g_img[c * plane + p] + acc[c] : acc[c];
}

template <typename T>
static SmoothFrame<T> make_frame(int B, int H, int W, const void* depth, const void* img, void* ws, T* out, void* g_depth,
                                 void* edge) {
  const SmoothWs l = smooth_ws_layout(B, H, W);
  SmoothFrame<T> f;
  f.depth = (const T*)depth; f.img = (const T*)img;
  f.per_img = reinterpret_cast<double*>((char*)ws + l.off_img);
  f.partials = reinterpret_cast<double*>((char*)ws + l.off_partials);
  f.out = out; f.g_depth = (T*)g_depth; f.edge = (T*)edge;
  return f;
}

template <typename T>
static int smooth_multi_fwd(int n, const void* const* depths, const void* const* imgs, int B, int H, int W, void* ws,
                            void* const* edges, T* out, T* total, void* stream_, const T* step_pg = nullptr,
                            double w1 = 0, double w2 = 0, double w3 = 0, T* step_out = nullptr) {
  clear_status();
  if (step_pg && (!step_out || !total || n > kMaxFrames)) return SCSFM_ERR_ARG;  // (one launch: the total is complete)
  if (n < 0 || B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || (n > 0 && (!depths || !imgs || !ws || !out))) return SCSFM_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!depths[i] || !imgs[i]) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const SmoothWs l = smooth_ws_layout(B, H, W);
  for (int i0 = 0; i0 < n; i0 += kMaxFrames) {
    const int m = n - i0 < kMaxFrames ? n - i0 : kMaxFrames;
    SmoothBatch<T> sb;
    for (int i = 0; i < m; ++i)
      sb.f[i] = make_frame<T>(B, H, W, depths[i0 + i], imgs[i0 + i], (char*)ws + (size_t)(i0 + i) * l.total, out + i0 + i,
                              nullptr, edges ? edges[i0 + i] : nullptr);
    sb.counter = reinterpret_cast<unsigned*>((char*)ws + (size_t)i0 * l.total + l.off_counter);
    sb.total = total;
    sb.first = i0 == 0 ? 1 : 0;
    sb.step_pg = step_pg; sb.step_out = step_out; sb.w1 = T(w1); sb.w2 = T(w2); sb.w3 = T(w3);
    hipLaunchKernelGGL((smooth_fwd_kernel<T>), dim3(l.fbx, l.fby, m * B), dim3(kThreads), 0, stream, sb, B, H, W);
    hipLaunchKernelGGL((smooth_finalize_kernel<T>), dim3(m), dim3(kThreads), 0, stream, sb, B, H, W, l.fbx * l.fby);
  }
  return launch_status();
}

template <typename T>
static int smooth_multi_bwd(int n, const void* const* depths, const void* const* imgs, int B, int H, int W, void* ws,
                            void* const* edges, const T* g_loss, void* const* g_depths, bool accumulate, void* stream_) {
  clear_status();
  if (n < 0 || B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || (n > 0 && (!depths || !imgs || !ws || !g_loss || !g_depths)))
    return SCSFM_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!depths[i] || !imgs[i]) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const SmoothWs l = smooth_ws_layout(B, H, W);
  for (int i0 = 0; i0 < n; i0 += kMaxFrames) {
    const int m = n - i0 < kMaxFrames ? n - i0 : kMaxFrames;
    SmoothBatch<T> sb;
    for (int i = 0; i < m; ++i)
      sb.f[i] = make_frame<T>(B, H, W, depths[i0 + i], imgs[i0 + i], (char*)ws + (size_t)(i0 + i) * l.total, nullptr,
                              g_depths[i0 + i], edges ? edges[i0 + i] : nullptr);
    sb.counter = nullptr; sb.total = nullptr; sb.first = 0; sb.step_pg = nullptr; sb.step_out = nullptr;
    if (accumulate)
      hipLaunchKernelGGL((smooth_bwd_kernel<T, true>), dim3(l.nbx, l.nby, m * B), dim3(kThreads), 0, stream, sb, B, H, W,
                         g_loss);
    else
      hipLaunchKernelGGL((smooth_bwd_kernel<T, false>), dim3(l.nbx, l.nby, m * B), dim3(kThreads), 0, stream, sb, B, H, W,
                         g_loss);
  }
  return launch_status();
}

template <typename T>
static int smooth_multi_bwd_images(int n, const void* const* depths, const void* const* imgs, int B, int H, int W, void* ws,
                                   const T* g_loss, void* const* g_imgs, bool accumulate, void* stream_) {
  clear_status();
  if (n < 0 || B <= 0 || H < 2 || W < 2 || !dims_ok<T>(B, H, W) || (n > 0 && (!depths || !imgs || !ws || !g_loss || !g_imgs))) return SCSFM_ERR_ARG;
  for (int i = 0; i < n; ++i)
    if (!depths[i] || !imgs[i]) return SCSFM_ERR_ARG;
  hipStream_t stream = (hipStream_t)stream_;
  const SmoothWs l = smooth_ws_layout(B, H, W);
  for (int i0 = 0; i0 < n; i0 += kMaxFrames) {
    const int m = n - i0 < kMaxFrames ? n - i0 : kMaxFrames;
    SmoothBatch<T> sb;
    for (int i = 0; i < m; ++i)
      sb.f[i] = make_frame<T>(B, H, W, depths[i0 + i], imgs[i0 + i], (char*)ws + (size_t)(i0 + i) * l.total, nullptr,
                              g_imgs[i0 + i], nullptr);
    sb.counter = nullptr; sb.total = nullptr; sb.first = 0; sb.step_pg = nullptr; sb.step_out = nullptr;
    hipLaunchKernelGGL((smooth_bwd_images_kernel<T>), dim3(ceil_div(W, kWave), ceil_div(H, kThreads / kWave), m * B),
                       dim3(kThreads), 0, stream, sb, B, H, W, g_loss, accumulate ? 1 : 0);
  }
  return launch_status();
}

}  // namespace scsfm

extern "C" {

size_t scsfm_smooth_ws_bytes(int B, int H, int W) {
  if (B <= 0 || H < 2 || W < 2) return 0;
  return scsfm::smooth_ws_layout(B, H, W).total;
}

#define SCSFM_SMOOTH_API(SUF, T)                                                                                     \
  int scsfm_smooth_multi_fwd_##SUF(int n, const void* const* depths, const void* const* imgs, int B, int H, int W,   \
                                   void* ws, void* const* edges, T* out, void* stream) {                             \
    return scsfm::smooth_multi_fwd<T>(n, depths, imgs, B, H, W, ws, edges, out, n > 0 ? out + n : nullptr, stream);   \
  }                                                                                                                  \
  int scsfm_smooth_multi_fwd_step_##SUF(int n, const void* const* depths, const void* const* imgs, int B, int H,     \
                                        int W, void* ws, void* const* edges, T* out, const T* photo_geom,           \
                                        double w_photo, double w_smooth, double w_geom, T* step_out, void* stream) { \
    if (n <= 0 || !photo_geom || !step_out) return SCSFM_ERR_ARG;                                                    \
    return scsfm::smooth_multi_fwd<T>(n, depths, imgs, B, H, W, ws, edges, out, out + n, stream, photo_geom, w_photo, \
                                      w_smooth, w_geom, step_out);                                                   \
  }                                                                                                                  \
  int scsfm_smooth_multi_bwd_##SUF(int n, const void* const* depths, const void* const* imgs, int B, int H, int W,   \
                                   void* ws, void* const* edges, const T* g_loss, void* const* g_depths,             \
                                   int accumulate, void* stream) {                                                   \
    return scsfm::smooth_multi_bwd<T>(n, depths, imgs, B, H, W, ws, edges, g_loss, g_depths, accumulate != 0,        \
                                      stream);                                                                       \
  }                                                                                                                  \
  int scsfm_smooth_multi_bwd_images_##SUF(int n, const void* const* depths, const void* const* imgs, int B, int H,   \
                                          int W, void* ws, const T* g_loss, void* const* g_imgs, int accumulate,     \
                                          void* stream) {                                                            \
    return scsfm::smooth_multi_bwd_images<T>(n, depths, imgs, B, H, W, ws, g_loss, g_imgs, accumulate != 0, stream);  \
  }                                                                                                                  \
  int scsfm_smooth_fwd_##SUF(int B, int H, int W, const T* depth, const T* img, void* ws, T* out, void* stream) {    \
    const void* d = depth; const void* im = img;                                                                     \
    if (!depth || !img) return SCSFM_ERR_ARG;                                                                        \
    return scsfm::smooth_multi_fwd<T>(1, &d, &im, B, H, W, ws, nullptr, out, nullptr, stream);                       \
  }                                                                                                                  \
  int scsfm_smooth_bwd_##SUF(int B, int H, int W, const T* depth, const T* img, void* ws, const T* g_loss,           \
                             T* g_depth, void* stream) {                                                             \
    const void* d = depth; const void* im = img; void* g = g_depth;                                                  \
    if (!depth || !img || !g_depth) return SCSFM_ERR_ARG;                                                            \
    return scsfm::smooth_multi_bwd<T>(1, &d, &im, B, H, W, ws, nullptr, g_loss, &g, true, stream);                   \
  }

SCSFM_SMOOTH_API(f32, float)
SCSFM_SMOOTH_API(f64, double)

}  // extern "C"
