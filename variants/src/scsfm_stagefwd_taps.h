// fwd_taps(): the staged 2 x 2 blocks of the forward warp (variant builds only; see scsfm_spec_stagefwd.inc)
#pragma once
namespace scsfm {
// The forward warp's staged texels (SCSFM_STAGE_FWD, scsfm_spec_tile.h): the reference view's three colours and its
// depth in a kFwdStageW x kFwdStageH window that covers where the tile's 66 x 18 pixels land.  Four LDS planes (they
// live in regions of the tile kernel that are not in use during the warp); a 2 x 2 block inside the window is four
// 2-dword LDS reads, any other block the eight global gathers of before.  kDepth = false: colours only (ring pixels).
constexpr int kFwdStageW = 72, kFwdStageH = 20;
template <bool kDepth, typename T, typename Map>
__device__ __forceinline__ void fwd_taps(const Sample<T>& s, bool staged, int fx0, int fy0, const T* l0, const T* l1,
                                         const T* l2, const T* ld, const T* __restrict__ ref_img, unsigned plane,
                                         const Map& ref_depth, TapRows<T> (&tc)[3], TapRows<T>& td) {
  const int lx = s.xa - fx0, ly = s.ya - fy0;
  if (staged && unsigned(lx) <= unsigned(kFwdStageW - 2) && unsigned(ly) <= unsigned(kFwdStageH - 2)) {
    const int o = ly * kFwdStageW + lx;
    const T* const lp[3] = {l0 + o, l1 + o, l2 + o};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      tc[c].n.a = lds_ld(lp[c]); tc[c].n.b = lds_ld(lp[c] + 1);
      tc[c].s.a = lds_ld(lp[c] + kFwdStageW); tc[c].s.b = lds_ld(lp[c] + kFwdStageW + 1);
    }
    if (kDepth) {
      const T* q = ld + o;
      td.n.a = lds_ld(q); td.n.b = lds_ld(q + 1); td.s.a = lds_ld(q + kFwdStageW); td.s.b = lds_ld(q + kFwdStageW + 1);
    }
  } else {
#pragma unroll
    for (int c = 0; c < 3; ++c) tc[c] = load_tap_rows(ref_img + c * plane, s);
    if (kDepth) td = ref_depth.taps(s);
  }
}

}  // namespace scsfm
