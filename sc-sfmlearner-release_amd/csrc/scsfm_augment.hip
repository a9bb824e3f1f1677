// Device-side input transform: the training transform of the reference
// (train.py:95-100: RandomHorizontalFlip -> RandomScaleCrop -> ArrayToTensor -> Normalize,
// custom_transforms.py:33-84) applied to decoded uint8 frames that are already in HBM, instead of
// per-sample PIL + numpy work on host cores.  Byte-exact with the reference: RandomScaleCrop resizes
// with Pillow's default filter (bicubic, a = -0.5, 8-bit fixed-point coefficients with 22 fraction
// bits, horizontal pass rounded to uint8 before the vertical pass -- Pillow's ImagingResample, a
// third-party dependency of the reference, unpinned in requirements.txt); the random draws and the
// coefficient tables are prepared on the host (scsfm_hip/augment.py) exactly as Pillow computes them,
// this kernel evaluates both passes for the cropped window only and maps the byte through a
// 256-entry table holding fl32(fl32(fl32(v / 255) - mean) / std).
//
// Per sample a parameter record and two coefficient tables:
//   htab[s][x] = {xmin, count, k0..k4, -} for output column ox + x of the scaled image (source columns
//   of the flipped frame), vtab[s][y] likewise for rows.  An axis that is not resized has the
//   identity entry {x, 1, 1 << 22}.
#include <stdint.h>

#include "scsfm_common.h"

namespace scsfm {

constexpr int kAugPrec = 22;  // Pillow: PRECISION_BITS = 32 - 8 - 2

__device__ __forceinline__ int aug_clip8(int v) {
  v >>= kAugPrec;  // arithmetic shift, as Pillow's clip8 lookup index
  return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Workgroup = 64 output columns x kAugRows output rows of one frame.  RandomScaleCrop only zooms in, so the source
// window of such a tile is at most (64 + 4) x (kAugRows + 4) pixels:
//   1. the window's bytes go to LDS with coalesced dword loads (the HWC rows of the window are contiguous runs);
//   2. the horizontal pass is evaluated ONCE per (source row, output column) -- not once per output pixel and
//      vertical tap -- rounded to uint8 exactly as Pillow does between its passes, and parked in LDS;
//   3. every output pixel combines its (at most 5) rows from there, maps the byte through the table and stores three
//      coalesced planes.
// (Round 2's kernel read up to 75 single bytes per output pixel from global memory: 0.70-0.75 TB/s on its 15 B per
// output pixel, profiles/r03a_input_pipeline.json.)
constexpr int kAugRows = 16;                  // output rows per workgroup: 4 per thread
constexpr int kAugSrcRows = kAugRows + 5;     // source rows a tile can touch (zoom >= 1: span <= rows + taps - 1)
constexpr int kAugSrcCols = kWave + 5;        // ... and source columns
constexpr int kAugRowBytes = ((kAugSrcCols * 3 + 3 + 3) / 4) * 4;  // window row in LDS (+ up to 3 bytes of alignment slack)

// First byte of a tile's source window in frame row `row`, column `col`.  Under the precondition both lie inside the
// frame; tables outside the documented class (a negative first row, h1 > W under a flip) are clamped into it, so that
// no address below `frames` or beyond the buffer is ever formed (round-4 advisor finding: only the upper end was guarded).
__device__ __forceinline__ const uint8_t* aug_row_first(const uint8_t* __restrict__ src, int row, int col, int H, int W) {
  row = row < 0 ? 0 : (row > H - 1 ? H - 1 : row);
  col = col < 0 ? 0 : (col > W - 1 ? W - 1 : col);
  return src + ((size_t)row * W + col) * 3;
}

__global__ __launch_bounds__(kThreads) void augment_kernel(int T, int H, int W, const uint8_t* __restrict__ frames,
                                                           const int* __restrict__ params,
                                                           const int* __restrict__ htab, const int* __restrict__ vtab,
                                                           const float* __restrict__ lut, float* __restrict__ out) {
  __shared__ uint32_t sSrc[kAugSrcRows][kAugRowBytes / 4];     // the source window, bytes as loaded
  __shared__ uint32_t sH[kAugSrcRows][kWave];                  // horizontal pass: r | g << 8 | b << 16 per source row and output column
  const int f = blockIdx.z, s = f / T;
  const int lane = threadIdx.x & (kWave - 1), wave = threadIdx.x / kWave;
  const int x0 = blockIdx.x * kWave, y0 = blockIdx.y * kAugRows;
  const int x = x0 + lane;
  const bool flip = params[8 * s] != 0;
  const int xl = x0 + kWave - 1 < W ? x0 + kWave - 1 : W - 1, yl = y0 + kAugRows - 1 < H ? y0 + kAugRows - 1 : H - 1;
  // source rows [v0, v1) and (unflipped) source columns [c0, c1) of the tile: the tables are monotone
  const int* __restrict__ vt = vtab + (size_t)s * H * 8;
  const int* __restrict__ ht = htab + (size_t)s * W * 8;
  const int v0 = vt[8 * y0], v1 = vt[8 * yl] + vt[8 * yl + 1];
  const int h0 = ht[8 * x0], h1 = ht[8 * xl] + ht[8 * xl + 1];
  // in frame coordinates a flipped tile reads columns W - h1 .. W - 1 - h0
  // PRECONDITION (include/scsfm_hip.h): tables of a zoom-in resize with at most 5 taps, so that a tile's window spans
  // at most kAugSrcRows x kAugSrcCols source pixels.  Tables that break it (the C entry point takes any) must not
  // reach beyond the two LDS arrays nor outside [frames, frames + bytes): the row count, every window column, every tap
  // row and the window's first row / column are clamped, and the dword loads are guarded at both ends of the buffer --
  // such a call returns meaningless pixels, not a fault.
  const int c0 = flip ? W - h1 : h0;  // (columns: h1 - h0 <= 64 + 5)
  const int nrows = v1 - v0 < 0 ? 0 : (v1 - v0 > kAugSrcRows ? kAugSrcRows : v1 - v0);
  const uint8_t* __restrict__ src = frames + (size_t)f * H * W * 3;
  // ---- 1. window -> LDS (dwords from the 4-byte aligned address at or before the window's first byte of each row)
  const uint8_t* const buf_end = frames + (size_t)gridDim.z * H * W * 3;
  {
    const int nd = kAugRowBytes / 4;
    for (int i = threadIdx.x; i < nrows * nd; i += kThreads) {
      const int r = i / nd, d = i - r * nd;
      const uint8_t* first = aug_row_first(src, v0 + r, c0, H, W);       // first byte of the window in this row
      const uint8_t* a = first - (reinterpret_cast<size_t>(first) & 3) + 4 * (size_t)d;  // aligned dword d of the row
      uint32_t w = 0;
      if (a >= frames && a + 4 <= buf_end) w = *reinterpret_cast<const uint32_t*>(a);
      else for (int k = 0; k < 4; ++k) if (a + k >= frames && a + k < buf_end) w |= uint32_t(a[k]) << (8 * k);  // the buffer's first / last bytes
      sSrc[r][d] = w;
    }
  }
  __syncthreads();
  // ---- 2. horizontal pass per (source row, output column)
  if (x < W) {
    const int* __restrict__ hb = ht + 8 * x;
    const int hmin = hb[0], hcnt = hb[1];
    int k[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) k[i] = i < hcnt ? hb[2 + i] : 0;
    for (int r = wave; r < nrows; r += kThreads / kWave) {
      const unsigned skew = unsigned(reinterpret_cast<size_t>(aug_row_first(src, v0 + r, c0, H, W)) & 3);  // the window's first byte inside its first dword
      const uint8_t* __restrict__ row = reinterpret_cast<const uint8_t*>(&sSrc[r][0]) + skew;
      int h[3] = {1 << (kAugPrec - 1), 1 << (kAugPrec - 1), 1 << (kAugPrec - 1)};
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        if (i < hcnt) {
          const int col = hmin + i;                                        // column of the (flipped) frame
          int wc = flip ? (W - 1 - col) - c0 : col - c0;                   // ... inside the window
          wc = wc < 0 ? 0 : (wc > kAugSrcCols - 1 ? kAugSrcCols - 1 : wc);  // (no-op under the precondition)
          const uint8_t* __restrict__ px = row + 3 * wc;
          h[0] += int(px[0]) * k[i]; h[1] += int(px[1]) * k[i]; h[2] += int(px[2]) * k[i];
        }
      }
      sH[r][lane] = uint32_t(aug_clip8(h[0])) | uint32_t(aug_clip8(h[1])) << 8 | uint32_t(aug_clip8(h[2])) << 16;
    }
  }
  __syncthreads();
  // ---- 3. vertical pass, table, store
  const int t = f - s * T, S = gridDim.z / T;
  const size_t plane = (size_t)H * W;
#pragma unroll
  for (int j = 0; j < kAugRows / (kThreads / kWave); ++j) {
    const int y = y0 + wave + j * (kThreads / kWave);
    if (x >= W || y >= H) continue;
    const int* __restrict__ vb = vt + 8 * y;
    const int vmin = vb[0], vcnt = vb[1];
    int acc[3] = {1 << (kAugPrec - 1), 1 << (kAugPrec - 1), 1 << (kAugPrec - 1)};
    for (int i = 0; i < (vcnt < 5 ? vcnt : 5); ++i) {
      int r = vmin - v0 + i;
      r = r < 0 ? 0 : (r > kAugSrcRows - 1 ? kAugSrcRows - 1 : r);  // (no-op under the precondition)
      const uint32_t p = sH[r][lane];
      const int kk = vb[2 + i];
      acc[0] += int(p & 255u) * kk; acc[1] += int((p >> 8) & 255u) * kk; acc[2] += int(p >> 16) * kk;
    }
    // frame-major output [T][S][3][H][W]: out[t] is the contiguous batch of frame t, as the nets want it
    const size_t o = ((size_t)t * S + s) * 3 * plane + (size_t)y * W + x;
    out[o] = lut[aug_clip8(acc[0])];
    out[o + plane] = lut[aug_clip8(acc[1])];
    out[o + 2 * plane] = lut[aug_clip8(acc[2])];
  }
}

}  // namespace scsfm

extern "C" int scsfm_augment_u8_f32(int n_frames, int frames_per_sample, int H, int W, const unsigned char* frames,
                                    const int* params, const int* htab, const int* vtab, const float* lut, float* out,
                                    void* stream) {
  using namespace scsfm;
  clear_status();
  if (n_frames <= 0 || frames_per_sample <= 0 || n_frames % frames_per_sample || H < 1 || W < 1 || !frames || !params ||
      !htab || !vtab || !lut || !out)
    return SCSFM_ERR_ARG;
  hipLaunchKernelGGL(augment_kernel, dim3(ceil_div(W, kWave), ceil_div(H, kAugRows), n_frames), dim3(kThreads), 0,
                     (hipStream_t)stream, frames_per_sample, H, W, (const uint8_t*)frames, params, htab, vtab, lut, out);
  return launch_status();
}
