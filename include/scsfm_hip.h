/*
 * scsfm_hip.h -- C ABI of libscsfm_hip.so: the SC-SfMLearner warp + loss hot path as hand-written
 * HIP kernels for gfx950 (MI355X).
 *
 * The reference (JiawangBian/SC-SfMLearner-Release) has no FFI layer: its operator API for this
 * path is a set of Python callables in inverse_warp.py / loss_functions.py.  Each
 * entry point below replaces the ATen op chain behind one of those callables and is what the
 * Python shim in sc-sfmlearner-release_amd/{inverse_warp,loss_functions}.py binds through ctypes
 * (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *  - All pointers are DEVICE pointers to contiguous NCHW fp32 ("_f32") or fp64 ("_f64", used by the
 *    gradient-check tests) arrays; the caller owns every buffer; nothing is retained after return.
 *  - `stream` is a hipStream_t passed as void*; all work is enqueued on it, no call synchronises.
 *  - Return value: 0 on success, SCSFM_ERR_ARG for a rejected argument, otherwise the hipError_t of
 *    the failed launch.  No exceptions cross the ABI.
 *  - "accumulate" outputs are added to (the caller zeroes them once per step); "store" outputs are
 *    overwritten.
 *  - `flags` is a bit set of SCSFM_* below; `with_ssim / with_mask / with_auto_mask / padding_mode`
 *    of loss_functions.py:50,95 map onto it one to one.
 */
#ifndef SCSFM_HIP_H_
#define SCSFM_HIP_H_

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCSFM_WITH_SSIM 1u      /* loss_functions.py:107 */
#define SCSFM_WITH_MASK 2u      /* loss_functions.py:111 */
#define SCSFM_WITH_AUTO_MASK 4u /* loss_functions.py:103 */
#define SCSFM_PAD_BORDER 8u     /* padding_mode == 'border' (default 'zeros'), inverse_warp.py:219-224,262 */
#define SCSFM_ROT_QUAT_FLAG 32u /* warp entry points only: the pose's rotation is a quaternion (legacy inverse_warp's
                                   rotation_mode='quat', inverse_warp.py:139-154,157-191); default euler */
#define SCSFM_C2P_OVERWRITE 64u /* scsfm_cam2pixel_* only: cam2pixel2's zeros-mode overwrite (inverse_warp.py:219-224) */
#define SCSFM_LEGACY_GRID 16u   /* warp entry points only: no zeros-mode coordinate overwrite, as the
                                   legacy inverse_warp / cam2pixel (inverse_warp.py:47-74,157-191) */

#define SCSFM_DEBUG_SKIP_PHOTO 256u /* scsfm_pair_bwd only, for per-kernel timing: skip the tiled pass */
#define SCSFM_DEBUG_SKIP_GEOM 512u  /* scsfm_pair_bwd only, for per-kernel timing: skip the per-pixel pass */

/* profiling only (results are WRONG): ablate one stage of the geometry pass */
#define SCSFM_DEBUG_X1 1024u  /* no scatter into g_ref_depth (no LDS window, no atomics) */
#define SCSFM_DEBUG_X2 2048u  /* no dense accumulate into g_tgt_depth */
#define SCSFM_DEBUG_X3 4096u  /* no 12-value block reduction / gP atomics */
#define SCSFM_DEBUG_X4 8192u  /* speculative forward: skip the per-pixel work of the geometry tail */
#define SCSFM_DEBUG_X5 32768u /* scatter into the LDS window but never flush it */
/* Debugging (results unchanged, runtime-flag instantiation, slower): scsfm_pairs_fwd (speculative forward) and
 * scsfm_pairs_bwd (the fallback geometry pass) count every signed wrap of a fixed-point cell of their scatter windows
 * (csrc/scsfm_geom.h: a cell holds +-2048 units; more than 32 near-cap pixels of one tile on one reference texel wrap
 * it) in a word of pair i's workspace (byte 256 * B + 104 of d[i].ws; cleared by every scsfm_pairs_fwd); the forward
 * also reports its count in out_i[7].  Intermediate wraps that cancel again are counted too: a count of 0 proves the
 * cells held, a non-zero count means the depth gradients of the step may be off by multiples of 4096 units. */
#define SCSFM_DEBUG_CHECK_WINDOW 65536u

#define SCSFM_DEBUG_KERNEL_ONLY 16384u /* scsfm_pairs_fwd only, for timing: launch the main kernel alone (the
                                          constants of an earlier identical call are still in `ws`) */

#define SCSFM_ROT_EULER 0 /* inverse_warp.py:77-112  */
#define SCSFM_ROT_QUAT 1  /* inverse_warp.py:115-136 */

#define SCSFM_OK 0
#define SCSFM_ERR_ARG (-1)

/* ABI version of this header; bumped on any change of a signature or of what an entry point does with its
 * buffers (2: the batched backwards store their depth gradients; scratch holds six planes.  3: scsfm_smooth_multi_bwd
 * takes `accumulate`; scsfm_step_total / scsfm_step_weights; scsfm_pair_desc::total.  4: scsfm_pixel2cam_*, scsfm_cam2pixel_*,
 * SCSFM_ROT_QUAT_FLAG for the warp entry points.  5: scsfm_pair_desc::depth_shift.  6: scsfm_pair_desc::hint,
 * scsfm_source_id.  8: scsfm_pairs_bwd_smooth, scsfm_smooth_multi_fwd_step.  7: gradients of the data inputs -- scsfm_pair_desc::g_tgt_img / g_ref_img, scsfm_pairs_bwd_inputs,
 * scsfm_warp_bwd_inputs, scsfm_pixel2cam_bwd_intrinsics, scsfm_masked_mean_bwd_mask, scsfm_smooth_multi_bwd_images). */
int scsfm_abi_version(void);
/* Identity of what this binary was built from: the first 16 hex digits of the sha256 over csrc/, this header AND the
 * compiler flags, any extra -D tuning knobs included (scsfm_hip/build.py: source_id(extra); "unknown" for a build that
 * did not record it), NUL-terminated into buf[n].  A tuning variant therefore never carries the default library's id. */
int scsfm_source_id(char* buf, size_t n);

/* ---------------------------------------------------------------------------------------------
 * compute_pairwise_loss (loss_functions.py:95-119) incl. inverse_warp2 (inverse_warp.py:230-269),
 * SSIM (loss_functions.py:11-42) and mean_on_mask (loss_functions.py:123-129), one (tgt, ref)
 * pair-direction.
 *
 * scsfm_pair_ws_bytes : size of the per-call device workspace `ws`.  The same `ws` must be handed,
 *                       untouched, from scsfm_pair_fwd to the matching scsfm_pair_bwd.
 * scsfm_pair_fwd      : out[8] (device, store) = { photo_loss, geometry_loss, S_photo, S_geom, S_mask,
 *                       0, 0, 0 } with S_photo = sum_{c,p} diff_img*m, S_geom = sum_p diff_depth*m,
 *                       S_mask = sum_p m.  The 10000-pixel gate of mean_on_mask is evaluated on the
 *                       device: a gated-off term is 0 and produces zero gradients (no host sync,
 *                       unlike the reference).
 * scsfm_pair_refinalize : data-parallel exact mode -- after the caller has all-reduced out[2..4]
 *                       over the ranks, recompute out[0..1] and the backward coefficients in `ws`
 *                       from the global sums (the masked means are ratios of whole-batch sums).
 * scsfm_pair_bwd      : `scratch` = scsfm_pair_bwd_scratch_bytes(B,H,W) bytes of device memory, contents
 *                       irrelevant before and after the call (six planes: dL/d warped colours and
 *                       dL/d diff_depth between the two backward kernels, then this pair's dense
 *                       dL/d tgt_depth and scattered dL/d ref_depth before they are added to the
 *                       caller's buffers); may be shared by consecutive calls on one stream.
 *                       g_photo / g_geom are device scalars (upstream gradients of the two losses).
 *                       g_tgt_depth [B,1,H,W] accumulate (dense), g_ref_depth [B,1,H,W] accumulate
 *                       (atomic scatter of the bilinear taps), g_pose [B,6] store.
 * --------------------------------------------------------------------------------------------- */
size_t scsfm_pair_ws_bytes(int B, int H, int W);
size_t scsfm_pair_bwd_scratch_bytes(int B, int H, int W);

/* Speculative forward: same results in `out` / `ws` as scsfm_pair_fwd, but computed by the backward's
 * tiled pass followed, in the same kernel, by the geometry pass: `gbuf` receives this pair's dense
 * dL/d tgt_depth and scattered dL/d ref_depth planes (and `ws` the partials of dL/d pose) up to the
 * factor g_photo / (3 S_mask), which the reduction only supplies afterwards -- valid if the upstream
 * gradients of (photo, geom) later stand in the ratio w_photo : w_geom (the loss weights,
 * train.py:268; w_photo != 0).  scsfm_pair_bwd must then be given the same `gbuf` as its `scratch`;
 * it verifies the ratio on the device, scales and adds the planes, and silently falls back to
 * running both passes when the ratio does not hold.  The whole pair-direction is then one kernel in
 * the forward and a streaming add in the backward. */
int scsfm_pair_fwd_spec_f32(int B, int H, int W, const float* tgt_img, const float* ref_img,
                            const float* tgt_depth, const float* ref_depth, const float* pose,
                            const float* intrinsics, unsigned flags, void* ws, void* gbuf, double w_photo,
                            double w_geom, float* out, void* stream);
int scsfm_pair_fwd_spec_f64(int B, int H, int W, const double* tgt_img, const double* ref_img,
                            const double* tgt_depth, const double* ref_depth, const double* pose,
                            const double* intrinsics, unsigned flags, void* ws, void* gbuf, double w_photo,
                            double w_geom, double* out, void* stream);

int scsfm_pair_fwd_f32(int B, int H, int W, const float* tgt_img, const float* ref_img,
                       const float* tgt_depth, const float* ref_depth, const float* pose,
                       const float* intrinsics, unsigned flags, void* ws, float* out, void* stream);
int scsfm_pair_bwd_f32(int B, int H, int W, const float* tgt_img, const float* ref_img,
                       const float* tgt_depth, const float* ref_depth, const float* pose,
                       const float* intrinsics, unsigned flags, void* ws, void* scratch,
                       const float* g_photo, const float* g_geom, float* g_tgt_depth,
                       float* g_ref_depth, float* g_pose, void* stream);
int scsfm_pair_refinalize_f32(int B, int H, int W, void* ws, float* out, void* stream);
int scsfm_pair_refinalize_f64(int B, int H, int W, void* ws, double* out, void* stream);
int scsfm_pair_fwd_f64(int B, int H, int W, const double* tgt_img, const double* ref_img,
                       const double* tgt_depth, const double* ref_depth, const double* pose,
                       const double* intrinsics, unsigned flags, void* ws, double* out, void* stream);
int scsfm_pair_bwd_f64(int B, int H, int W, const double* tgt_img, const double* ref_img,
                       const double* tgt_depth, const double* ref_depth, const double* pose,
                       const double* intrinsics, unsigned flags, void* ws, void* scratch,
                       const double* g_photo, const double* g_geom, double* g_tgt_depth,
                       double* g_ref_depth, double* g_pose, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Several pair-directions per call -- what compute_photo_and_geometry_loss (loss_functions.py:56-90)
 * needs per step: refs x scales x 2 directions.  `d` is a HOST array of n descriptors holding DEVICE
 * pointers; every pair shares B, H, W, the intrinsics, the flags and (backward) the scratch buffer
 * and the upstream gradients.  Field use: forward reads tgt_img..pose, ws, out; backward reads
 * tgt_img..pose, ws and STORES g_tgt_depth, g_ref_depth, g_pose: a depth-gradient buffer named by several
 * descriptors of one call (the same map is the target of one pair and the reference of another) receives
 * the sum of their contributions, and needs no zero-fill beforehand (n <= 128 per call).  Otherwise the
 * semantics per pair are those of scsfm_pair_fwd / scsfm_pair_bwd; the descriptors are consumed before
 * return.
 * --------------------------------------------------------------------------------------------- */
typedef struct scsfm_pair_desc {
  const void* tgt_img;
  const void* ref_img;
  const void* tgt_depth;
  const void* ref_depth;
  const void* pose;
  void* ws;
  void* out;
  void* g_tgt_depth;
  void* g_ref_depth;
  void* g_pose;
  void* gbuf; /* NULL, or scsfm_pair_bwd_scratch_bytes(B,H,W) bytes private to this pair, alive from the
                 forward to the backward: enables the speculative forward (see scsfm_pair_fwd_spec) */
  void* total; /* read from d[0] only, may be NULL: 2 elements (device, store) that the forward fills with the
                  sums over all n pair-directions of out[0] (photo) and out[1] (geometry) -- what
                  compute_photo_and_geometry_loss returns (loss_functions.py:89-92) -- without a launch of its own */
  void* hint; /* read from d[0] only, may be NULL: 2 doubles on the device = the (w_photo, w_geom) the speculative forward
                 assumes the upstream gradients will stand in.  With it the forward ignores the host values of
                 scsfm_pairs_fwd (beyond "speculate at all": a non-NULL gbuf) and reads the pair from here, and every
                 scsfm_pairs_bwd leaves the upstream gradients it actually saw in it: whatever weights a training loop
                 uses (train.py:268), the step after the first one speculates on the right ratio -- no host
                 round trip, capturable in a HIP graph.  The caller owns the two doubles and initialises them */
  int depth_shift; /* 0: tgt_depth / ref_depth (and their gradient buffers) are [B,1,H,W].  s > 0: they are the maps of
                      a coarser scale, [B,1,H>>s,W>>s] with H, W multiples of 2^s, and the kernels read them through
                      the index map of F.interpolate(..., (H, W), mode='nearest') (loss_functions.py:77-82) instead of a
                      materialised up-sampled copy; g_tgt_depth / g_ref_depth receive the sum-pooled gradients */
  void* g_tgt_img; /* read by scsfm_pairs_bwd_inputs only; NULL or [B,3,H,W], ACCUMULATE (atomic adds): dL/d tgt_img of this */
  void* g_ref_img; /* pair-direction, dL/d ref_img likewise -- zero the buffers before the first call that names them */
  /* ABI 9 -- get_smooth_loss (loss_functions.py:133-152) of this pair's TARGET frame evaluated by the speculative
     forward, in the tile that holds the frame's depth and colours anyway (train.py:262-266 calls both losses on the same
     frames; a separate pass over every frame costs 36 + 7 us per step at configs[1]).  Read by scsfm_pairs_fwd /
     scsfm_pairs_fwd_step only, and only from descriptors with a gbuf and depth_shift 0 (anything else: argument error): */
  void* smooth_ws;   /* NULL, or the frame's smooth workspace (scsfm_smooth_ws_bytes): receives {mean_HW(D) + 1e-7, L} per
                        image exactly as scsfm_smooth_multi_fwd leaves them -- scsfm_smooth_multi_bwd and
                        scsfm_pairs_bwd_smooth take it from there */
  void* smooth_edge; /* with smooth_ws: NULL or [B,1,H,W] (store): every pixel's summed edge terms, bit for bit the plane
                        scsfm_smooth_multi_fwd writes */
  void* smooth_out;  /* with smooth_ws: 1 element (store) = this frame's loss */
  void* smooth_total; /* read from d[0] only, may be NULL: 1 element (store) = the sum of smooth_out over the descriptors
                         that carry a smooth_ws, in descriptor order = compute_smooth_loss (loss_functions.py:154-159)
                         when each frame of the step is the target of exactly one of them */
} scsfm_pair_desc;

int scsfm_pairs_fwd_f32(int n, const scsfm_pair_desc* d, int B, int H, int W, const float* intrinsics,
                        unsigned flags, double w_photo, double w_geom, void* stream);
int scsfm_pairs_bwd_f32(int n, const scsfm_pair_desc* d, int B, int H, int W, const float* intrinsics,
                        unsigned flags, void* scratch, const float* g_photo, const float* g_geom,
                        void* stream);
int scsfm_pairs_fwd_f64(int n, const scsfm_pair_desc* d, int B, int H, int W, const double* intrinsics,
                        unsigned flags, double w_photo, double w_geom, void* stream);
/* ABI 9: scsfm_pairs_fwd that also forms the step's objective (train.py:268) in its finalize launch: step_out[4] (device,
 * store) = { w_photo * photo + w_smooth * smooth + w_geom * geometry, photo, smooth, geometry } with photo / geometry the
 * sums d[0].total receives and smooth the sum d[0].smooth_total receives (both must be non-NULL; n <= 8: one launch). */
int scsfm_pairs_fwd_step_f32(int n, const scsfm_pair_desc* d, int B, int H, int W, const float* intrinsics,
                             unsigned flags, double w_photo, double w_smooth, double w_geom, float* step_out,
                             void* stream);
int scsfm_pairs_fwd_step_f64(int n, const scsfm_pair_desc* d, int B, int H, int W, const double* intrinsics,
                             unsigned flags, double w_photo, double w_smooth, double w_geom, double* step_out,
                             void* stream);
int scsfm_pairs_bwd_f64(int n, const scsfm_pair_desc* d, int B, int H, int W, const double* intrinsics,
                        unsigned flags, void* scratch, const double* g_photo, const double* g_geom,
                        void* stream);

/* The same backward with the smooth loss's gradient riding along (ABI 8; loss_functions.py:132-159 next to :50-92, as
 * train.py:262-268 combines them): frame_grads[j] -- one of the full-resolution depth-gradient buffers the descriptors
 * name -- additionally receives g_smooth[0] * d smooth_j / d depth_j in the very pass that stores it, instead of being
 * re-read and re-written by scsfm_smooth_multi_bwd.  frame_edges[j]: the edge plane scsfm_smooth_multi_fwd left for
 * frame j; frame_stats[j]: the first 2 * B doubles of frame j's smooth workspace ({mean + 1e-7, loss} per image).
 * n_frames <= 16; a frame whose buffer no descriptor names is an argument error. */
int scsfm_pairs_bwd_smooth_f32(int n, const scsfm_pair_desc* d, int B, int H, int W, const float* intrinsics,
                               unsigned flags, void* scratch, const float* g_photo, const float* g_geom, int n_frames,
                               void* const* frame_grads, void* const* frame_edges, void* const* frame_stats,
                               const float* g_smooth, void* stream);
int scsfm_pairs_bwd_smooth_f64(int n, const scsfm_pair_desc* d, int B, int H, int W, const double* intrinsics,
                               unsigned flags, void* scratch, const double* g_photo, const double* g_geom, int n_frames,
                               void* const* frame_grads, void* const* frame_edges, void* const* frame_stats,
                               const double* g_smooth, void* stream);

/* Gradients of the DATA inputs of the pair losses.  The reference's autograd reaches the images (the target image
 * through the L1 and SSIM terms, loss_functions.py:99-108; the reference image through grid_sample's input,
 * inverse_warp.py:262) and the intrinsics (K^-1 of pixel2cam and K [R|t], inverse_warp.py:253-260) although train.py
 * never asks for them.  Call AFTER scsfm_pairs_bwd with the same descriptors, flags and upstream gradients (the pose
 * partials the backward left in each `ws` are read again): every non-NULL d[i].g_tgt_img / g_ref_img is ACCUMULATED
 * into (one extra tiled pass over the pairs that name one); g_intrinsics (NULL or [B,3,3]) is STORED = the sum over
 * all n pair-directions. */
int scsfm_pairs_bwd_inputs_f32(int n, const scsfm_pair_desc* d, int B, int H, int W, const float* intrinsics,
                               unsigned flags, const float* g_photo, const float* g_geom, float* g_intrinsics,
                               void* stream);
int scsfm_pairs_bwd_inputs_f64(int n, const scsfm_pair_desc* d, int B, int H, int W, const double* intrinsics,
                               unsigned flags, const double* g_photo, const double* g_geom, double* g_intrinsics,
                               void* stream);

/* ---------------------------------------------------------------------------------------------
 * inverse_warp2 (inverse_warp.py:230-269) as maps: projected_img [B,3,H,W], valid_mask [B,1,H,W]
 * (0/1), projected_depth [B,1,H,W], computed_depth [B,1,H,W] (all store).  The backward takes the
 * upstream gradients of the three differentiable maps (any may be NULL) and produces g_depth
 * (accumulate), g_ref_depth (accumulate, atomic scatter), g_pose [B,6] (store).  `ws` needs
 * scsfm_warp_ws_bytes(B) bytes.  Only SCSFM_PAD_BORDER, SCSFM_LEGACY_GRID and SCSFM_ROT_QUAT_FLAG are read from `flags`.
 * --------------------------------------------------------------------------------------------- */
size_t scsfm_warp_ws_bytes(int B);

int scsfm_warp_fwd_f32(int B, int H, int W, const float* img, const float* depth,
                       const float* ref_depth, const float* pose, const float* intrinsics,
                       unsigned flags, void* ws, float* projected_img, float* valid_mask,
                       float* projected_depth, float* computed_depth, void* stream);
int scsfm_warp_bwd_f32(int B, int H, int W, const float* img, const float* depth,
                       const float* ref_depth, const float* pose, const float* intrinsics,
                       unsigned flags, void* ws, const float* g_projected_img,
                       const float* g_projected_depth, const float* g_computed_depth,
                       float* g_depth, float* g_ref_depth, float* g_pose, void* stream);
int scsfm_warp_fwd_f64(int B, int H, int W, const double* img, const double* depth,
                       const double* ref_depth, const double* pose, const double* intrinsics,
                       unsigned flags, void* ws, double* projected_img, double* valid_mask,
                       double* projected_depth, double* computed_depth, void* stream);
int scsfm_warp_bwd_f64(int B, int H, int W, const double* img, const double* depth,
                       const double* ref_depth, const double* pose, const double* intrinsics,
                       unsigned flags, void* ws, const double* g_projected_img,
                       const double* g_projected_depth, const double* g_computed_depth,
                       double* g_depth, double* g_ref_depth, double* g_pose, void* stream);
/* ... and of its data inputs, AFTER scsfm_warp_bwd on the same `ws` (which keeps the sums of dL/d(K [R|t])):
 * g_img (NULL or [B,3,H,W], ACCUMULATE: the bilinear splat of g_projected_img, grid_sampler_2d_backward on its input)
 * and g_intrinsics (NULL or [B,3,3], STORE). */
int scsfm_warp_bwd_inputs_f32(int B, int H, int W, const float* depth, const float* pose, const float* intrinsics,
                              unsigned flags, void* ws, const float* g_projected_img, float* g_img,
                              float* g_intrinsics, void* stream);
int scsfm_warp_bwd_inputs_f64(int B, int H, int W, const double* depth, const double* pose, const double* intrinsics,
                              unsigned flags, void* ws, const double* g_projected_img, double* g_img,
                              double* g_intrinsics, void* stream);

/* ---------------------------------------------------------------------------------------------
 * pixel2cam (inverse_warp.py:29-44): depth [B,H,W], intrinsics_inv [B,3,3] -> cam [B,3,H,W] (store) =
 * K^-1 (u, v, 1) * depth.  Backward: g_cam [B,3,H,W] -> g_depth [B,H,W] (store); the gradient with respect to
 * intrinsics_inv has its own entry point (scsfm_pixel2cam_bwd_intrinsics: a caller that treats intrinsics as data
 * never pays for it).
 * cam2pixel (inverse_warp.py:47-74) and cam2pixel2 (:194-227; flags = SCSFM_C2P_OVERWRITE for padding_mode 'zeros',
 * z != NULL): cam [B,3,H,W], rot [B,3,3] or NULL, tr [B,3] or NULL -> grid [B,H,W,2] (store) and, optionally, the
 * clamped depth z [B,1,H,W] (store).  Backward: g_grid [B,H,W,2], g_z [B,1,H,W] or NULL -> g_cam [B,3,H,W] (store)
 * and g_rot_tr [B][12] fp64 (store: dL/d rot row-major, then dL/d tr).
 * --------------------------------------------------------------------------------------------- */
int scsfm_pixel2cam_fwd_f32(int B, int H, int W, const float* depth, const float* intrinsics_inv, float* cam,
                            void* stream);
int scsfm_pixel2cam_bwd_f32(int B, int H, int W, const float* intrinsics_inv, const float* g_cam, float* g_depth,
                            void* stream);
/* dL/d intrinsics_inv [B,3,3] (store) of pixel2cam = sum over the pixels of g_cam (x) (u, v, 1) depth. */
int scsfm_pixel2cam_bwd_intrinsics_f32(int B, int H, int W, const float* depth, const float* g_cam,
                                       float* g_intrinsics_inv, void* stream);
int scsfm_pixel2cam_bwd_intrinsics_f64(int B, int H, int W, const double* depth, const double* g_cam,
                                       double* g_intrinsics_inv, void* stream);
int scsfm_cam2pixel_fwd_f32(int B, int H, int W, const float* cam, const float* rot, const float* tr, unsigned flags,
                            float* grid, float* z, void* stream);
int scsfm_cam2pixel_bwd_f32(int B, int H, int W, const float* cam, const float* rot, const float* tr, unsigned flags,
                            const float* g_grid, const float* g_z, float* g_cam, double* g_rot_tr, void* stream);
int scsfm_pixel2cam_fwd_f64(int B, int H, int W, const double* depth, const double* intrinsics_inv, double* cam,
                            void* stream);
int scsfm_pixel2cam_bwd_f64(int B, int H, int W, const double* intrinsics_inv, const double* g_cam, double* g_depth,
                            void* stream);
int scsfm_cam2pixel_fwd_f64(int B, int H, int W, const double* cam, const double* rot, const double* tr, unsigned flags,
                            double* grid, double* z, void* stream);
int scsfm_cam2pixel_bwd_f64(int B, int H, int W, const double* cam, const double* rot, const double* tr, unsigned flags,
                            const double* g_grid, const double* g_z, double* g_cam, double* g_rot_tr, void* stream);

/* ---------------------------------------------------------------------------------------------
 * pose_vec2mat (inverse_warp.py:139-154): vec [B,6] = (tx,ty,tz,rx,ry,rz) -> mat [B,3,4] (store);
 * backward: g_mat [B,3,4] -> g_vec [B,6] (store).  mode = SCSFM_ROT_EULER | SCSFM_ROT_QUAT.
 * --------------------------------------------------------------------------------------------- */
int scsfm_pose_vec2mat_fwd_f32(int B, const float* vec, int mode, float* mat, void* stream);
int scsfm_pose_vec2mat_bwd_f32(int B, const float* vec, int mode, const float* g_mat, float* g_vec,
                               void* stream);
int scsfm_pose_vec2mat_fwd_f64(int B, const double* vec, int mode, double* mat, void* stream);
int scsfm_pose_vec2mat_bwd_f64(int B, const double* vec, int mode, const double* g_mat,
                               double* g_vec, void* stream);

/* ---------------------------------------------------------------------------------------------
 * get_smooth_loss of compute_smooth_loss (loss_functions.py:132-152), one frame:
 * depth [B,1,H,W], img [B,3,H,W] -> out[1] (device, store).  Backward: g_loss device scalar,
 * g_depth accumulate.  `ws` (scsfm_smooth_ws_bytes) carries the per-image sums fwd -> bwd.
 * --------------------------------------------------------------------------------------------- */
size_t scsfm_smooth_ws_bytes(int B, int H, int W);

int scsfm_smooth_fwd_f32(int B, int H, int W, const float* depth, const float* img, void* ws,
                         float* out, void* stream);
int scsfm_smooth_bwd_f32(int B, int H, int W, const float* depth, const float* img, void* ws,
                         const float* g_loss, float* g_depth, void* stream);
int scsfm_smooth_fwd_f64(int B, int H, int W, const double* depth, const double* img, void* ws,
                         double* out, void* stream);
int scsfm_smooth_bwd_f64(int B, int H, int W, const double* depth, const double* img, void* ws,
                         const double* g_loss, double* g_depth, void* stream);

/* ---------------------------------------------------------------------------------------------
 * SSIM module (loss_functions.py:11-42) stand-alone: x, y [N,H,W] planes (N = B*C) ->
 * out = clamp((1 - SSIM)/2, 0, 1) (store).  Backward: g_out -> g_x, g_y (store; either may be NULL).
 * --------------------------------------------------------------------------------------------- */
int scsfm_ssim_fwd_f32(int N, int H, int W, const float* x, const float* y, float* out, void* stream);
int scsfm_ssim_bwd_f32(int N, int H, int W, const float* x, const float* y, const float* g_out,
                       float* g_x, float* g_y, void* stream);
int scsfm_ssim_fwd_f64(int N, int H, int W, const double* x, const double* y, double* out, void* stream);
int scsfm_ssim_bwd_f64(int N, int H, int W, const double* x, const double* y, const double* g_out,
                       double* g_x, double* g_y, void* stream);

/* ---------------------------------------------------------------------------------------------
 * mean_on_mask (loss_functions.py:123-129): diff [B,C,HW], mask [B,Cm,HW] with Cm in {1, C}
 * (broadcast over channels) -> out[1] = sum(diff*mask)/sum(mask) if sum(mask expanded) > 10000
 * else 0 (store, gate on the device).  Backward: g (device scalar) -> g_diff [B,C,HW] (store).
 * `ws` (scsfm_masked_mean_ws_bytes) carries the sums fwd -> bwd.
 * --------------------------------------------------------------------------------------------- */
size_t scsfm_masked_mean_ws_bytes(void);
int scsfm_masked_mean_fwd_f32(int B, int C, int Cm, int HW, const float* diff, const float* mask,
                              void* ws, float* out, void* stream);
int scsfm_masked_mean_bwd_f32(int B, int C, int Cm, int HW, const float* mask, void* ws,
                              const float* g, float* g_diff, void* stream);
int scsfm_masked_mean_fwd_f64(int B, int C, int Cm, int HW, const double* diff, const double* mask,
                              void* ws, double* out, void* stream);
int scsfm_masked_mean_bwd_f64(int B, int C, int Cm, int HW, const double* mask, void* ws,
                              const double* g, double* g_diff, void* stream);
/* ... and with respect to a floating-point mask [B,Cm,HW] (store; diff as in the forward, `ws` from it): the
 * reference's autograd reaches the mask through both sums of mean_on_mask (loss_functions.py:123-129). */
int scsfm_masked_mean_bwd_mask_f32(int B, int C, int Cm, int HW, const float* diff, void* ws, const float* g,
                                   float* g_mask, void* stream);
int scsfm_masked_mean_bwd_mask_f64(int B, int C, int Cm, int HW, const double* diff, void* ws, const double* g,
                                   double* g_mask, void* stream);

/* compute_smooth_loss (loss_functions.py:154-159): n frames per call.  depths / imgs / g_depths / edges
 * are HOST arrays of n DEVICE pointers; ws = n * scsfm_smooth_ws_bytes(B,H,W) bytes; out[n + 1] (device,
 * store) holds one loss per frame and, in out[n], their sum (what compute_smooth_loss returns); g_depths[i] is STORED (every pixel is written, no zero-fill needed;
 * the single-frame scsfm_smooth_bwd accumulates) unless `accumulate` is non-zero; a NULL g_depths[i] skips
 * that frame's gradient.
 * edges (may be NULL, as may any entry): per frame a [B,H,W] plane in which the forward leaves each
 * pixel's summed edge terms; given the same plane, the backward is a pure stream (4 B read + 4 B
 * written per pixel) instead of re-evaluating the edge weights from the images. */
int scsfm_smooth_multi_fwd_f32(int n, const void* const* depths, const void* const* imgs, int B, int H,
                               int W, void* ws, void* const* edges, float* out, void* stream);
/* ... and, in the launch that finishes the frames' total, the step's objective (train.py:268) from it and the pair losses
 * computed before (ABI 8): photo_geom[2] = {photo, geometry} (device) -> step_out[4] = {w_photo photo + w_smooth smooth +
 * w_geom geometry, photo, smooth, geometry} (device, store), instead of a launch of scsfm_step_total.  n <= 8. */
int scsfm_smooth_multi_fwd_step_f32(int n, const void* const* depths, const void* const* imgs, int B, int H, int W,
                                    void* ws, void* const* edges, float* out, const float* photo_geom, double w_photo,
                                    double w_smooth, double w_geom, float* step_out, void* stream);
int scsfm_smooth_multi_fwd_step_f64(int n, const void* const* depths, const void* const* imgs, int B, int H, int W,
                                    void* ws, void* const* edges, double* out, const double* photo_geom, double w_photo,
                                    double w_smooth, double w_geom, double* step_out, void* stream);
int scsfm_smooth_multi_bwd_f32(int n, const void* const* depths, const void* const* imgs, int B, int H,
                               int W, void* ws, void* const* edges, const float* g_loss,
                               void* const* g_depths, int accumulate, void* stream);
int scsfm_smooth_multi_fwd_f64(int n, const void* const* depths, const void* const* imgs, int B, int H,
                               int W, void* ws, void* const* edges, double* out, void* stream);
int scsfm_smooth_multi_bwd_f64(int n, const void* const* depths, const void* const* imgs, int B, int H,
                               int W, void* ws, void* const* edges, const double* g_loss,
                               void* const* g_depths, int accumulate, void* stream);
/* dL/d img of the same frames (the reference's autograd reaches the images through the edge weights
 * exp(-mean_c |dI|), loss_functions.py:148-152; train.py never asks): g_imgs is a HOST array of n DEVICE pointers,
 * each NULL (skip) or [B,3,H,W], stored -- or added to when `accumulate` is non-zero.  `ws` as left by the forward. */
int scsfm_smooth_multi_bwd_images_f32(int n, const void* const* depths, const void* const* imgs, int B, int H,
                                      int W, void* ws, const float* g_loss, void* const* g_imgs, int accumulate,
                                      void* stream);
int scsfm_smooth_multi_bwd_images_f64(int n, const void* const* depths, const void* const* imgs, int B, int H,
                                      int W, void* ws, const double* g_loss, void* const* g_imgs, int accumulate,
                                      void* stream);

/* The weighted sum of a training step, loss = w_photo * photo + w_smooth * smooth + w_geom * geometry
 * (train.py:268), for callers that keep the three losses behind ONE autograd node:
 * scsfm_step_total  : photo_geom[2] = {photo, geometry} (the `total` of scsfm_pairs_fwd), smooth[1] (out[n] of
 *                     scsfm_smooth_multi_fwd) -> out[4] = {loss, photo, smooth, geometry} (device, store).
 * scsfm_step_weights: g_loss[1] -> out[3] = {w_photo g, w_geom g, w_smooth g}: the upstream gradients to hand
 *                     to scsfm_pairs_bwd (out, out + 1) and scsfm_smooth_multi_bwd (out + 2). */
int scsfm_step_total_f32(const float* photo_geom, const float* smooth, double w_photo, double w_smooth,
                         double w_geom, float* out, void* stream);
int scsfm_step_total_f64(const double* photo_geom, const double* smooth, double w_photo, double w_smooth,
                         double w_geom, double* out, void* stream);
int scsfm_step_weights_f32(const float* g_loss, double w_photo, double w_smooth, double w_geom, float* out,
                           void* stream);
int scsfm_step_weights_f64(const double* g_loss, double w_photo, double w_smooth, double w_geom, double* out,
                           void* stream);

/* ---------------------------------------------------------------------------------------------
 * The training input transform on the device (train.py:95-100; custom_transforms.py:33-84):
 * RandomHorizontalFlip -> RandomScaleCrop (Pillow bicubic, byte-exact) -> ArrayToTensor -> Normalize.
 * frames: uint8 [n_frames, H, W, 3] (decoded images, HWC); consecutive groups of frames_per_sample frames
 * share one record of params [n_samples][8] = {flip, ...} and one row of the coefficient tables
 * htab [n_samples][W][8], vtab [n_samples][H][8] = {first source index, tap count, 5 fixed-point taps, -}
 * (prepared on the host as Pillow's precompute_coeffs + normalize_coeffs_8bpc do, for the cropped window);
 * lut [256] = the float a byte maps to; out: fp32 [frames_per_sample, n_samples, 3, H, W] (store): frame-major,
 * so that out[t] is the contiguous batch of the t-th frame of every sample.
 * PRECONDITION on the tables: those of a zoom-in resize (output size >= the cropped source window, as RandomScaleCrop
 * produces: custom_transforms.py:62-84) with monotone first indices and at most 5 taps -- a 64 x 16 tile of outputs then
 * reads at most 69 x 21 source pixels, which is what the kernel stages in LDS.  Tables outside that class are not
 * rejected (they live on the device); the kernel clamps every index they imply into its staging arrays, so such a call
 * returns meaningless pixels but never touches memory outside [frames, frames + n_frames * H * W * 3) (the window's
 * first row / column are clamped into the frame and the dword loads are guarded at both ends of the buffer).
 * --------------------------------------------------------------------------------------------- */
int scsfm_augment_u8_f32(int n_frames, int frames_per_sample, int H, int W, const unsigned char* frames,
                         const int* params, const int* htab, const int* vtab, const float* lut, float* out,
                         void* stream);

/* ---------------------------------------------------------------------------------------------
 * Measurement hook (bench.py's roofline): between scsfm_profile_begin(n) and scsfm_profile_end, the first n
 * launches of the dominant kernel (the speculative forward of scsfm_pairs_fwd) are bracketed by HIP events
 * recorded on the stream they are launched on -- i.e. inside real steps, with whatever the other kernels of
 * the step leave in the caches.  scsfm_profile_end waits for them and reports the mean and the minimum
 * duration in microseconds and how many launches were bracketed.  Not thread-safe; off by default.
 * --------------------------------------------------------------------------------------------- */
int scsfm_profile_begin(int n);
int scsfm_profile_end(double* mean_us, double* min_us, int* count);

#ifdef __cplusplus
}
#endif
#endif /* SCSFM_HIP_H_ */
