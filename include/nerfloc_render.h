/* nerfloc_render.h — C-ABI of libnerfloc_render.so (MI355X / gfx950 HIP renderer for NeRF-Loc's
 * conditional-NeRF hot path).  Plain pointers and sizes only; no torch / C++ types.
 *
 * What each entry point replaces in the reference (paths relative to /root/reference/nerf_loc/models/):
 *   nl_knn               pytorch3d.ops.knn_points == ops/knn/knn_utils.py:97-174 -> ops/knn/src/knn_api.cpp:10-14
 *                        (`knn_points_idx`), CUDA path ops/knn/src/knn.cu:131-241, CPU path knn_cpu.cpp:13-64;
 *                        call sites conditional_nerf/model.py:289,318,376-383
 *   nl_mv_aggregate      MultiviewFeatureAggregator.forward, conditional_nerf/multiview_aggregator.py:156-222
 *                        (Projector.compute ibrnet/ibrnet.py:194-231, project_points_dict depth_fusion.py:128-147,
 *                         MixtureLogisticsDistDecoder visibility_decoder.py:99-148)
 *   nl_point_mlp         ConditionalNeRF.query neighbour branch, conditional_nerf/model.py:372-427
 *                        (base_mlp :63-71, MultiHeadAttention ibrnet/ibrnet.py:69-119, aggregation :415-427)
 *   nl_ray_unet          RayUnet.forward, conditional_nerf/ray_unet.py:55-69
 *   nl_heads_composite   sigma/rgb-blend/feat heads + alpha compositing + valid mask, conditional_nerf/model.py:525-598
 *   nl_coarse_weights    MultiviewFeatureAggregator.predict_weights_from_neuray, multiview_aggregator.py:95-154
 *   nl_sample_pdf        sample_pdf + sort/merge, conditional_nerf/utils.py:73-112 and model.py:492-495
 *   nl_render_rays       ConditionalNeRF.render_rays, conditional_nerf/model.py:472-600 (eval mode), rows a2-a18 fused
 *   nl_composite_backward  autograd of model.py:544-560,597 (compositing) — first slice of the backward pass
 *   nl_point_mlp_backward  autograd of model.py:372-427 w.r.t. the sample positions / directions / query features (frozen weights)
 *   nl_knn_backward      ops/knn/src/knn.cu:449-490 (KNearestNeighborBackwardKernel), knn_cpu.cpp:68-117
 *   nl_pack_weights      (no reference counterpart: state_dict fp32 tensors -> kernel layouts; names = SURVEY App. C)
 *   nl_frame_*           per-frame caches the reference keeps on the module: `support_neural_points['fine']`
 *                        (model.py:79,180-197) and `multiview_aggregator.vis_featmaps` (multiview_aggregator.py:29,178)
 *
 * Conventions
 *   - every function returns 0 (NL_OK) or a negative nl_status; nl_strerror() names it; nothing throws.
 *   - all tensor pointers are DEVICE pointers to contiguous fp32 unless marked HOST; the library never
 *     allocates device memory: the caller passes workspaces sized by the *_bytes() queries.
 *   - `stream` is a hipStream_t (as void*); kernels are enqueued asynchronously on it, no host sync.
 *   - indices are int32 on the device side (N, M < 2^31); sizes are int64_t in the signatures.
 */
#ifndef NERFLOC_RENDER_H
#define NERFLOC_RENDER_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define NL_ABI_VERSION 7   /* 2: nl_render_rays_ex / nl_render_opts (early termination, per-ray query centres); 3: nl_render_opts.flags,
                            * reserved fields validated, side stream owned by the nl_frame; 5: NL_PREC_F16MX; 6: nl_frame_diagnostics;
                            * 7: NL_RENDER_PRECISION_GUARD (the precision guard at the boundary), NL_DIAG_GUARD_* */
#define NL_MAX_VIEWS 16
#define NL_KNN_MAX_K 8

typedef enum nl_status {
  NL_OK = 0,
  NL_ERR_BAD_ARG = -1,       /* null pointer / negative size / inconsistent shapes */
  NL_ERR_UNSUPPORTED = -2,   /* shape outside what the kernels are built for (see nl_config) */
  NL_ERR_WORKSPACE = -3,     /* workspace too small; call the matching *_bytes() */
  NL_ERR_HIP = -4,           /* a HIP runtime call failed (hipGetLastError is left set) */
  NL_ERR_NO_DEVICE = -5
} nl_status;

typedef enum nl_precision {
  NL_PREC_F32 = 0,     /* f32-input MFMA (v_mfma_f32_32x32x2_f32): exact fp32 products/accumulation   */
  NL_PREC_BF16X3 = 1,  /* 3-term split-bf16 MFMA (hi*hi + hi*lo + lo*hi), fp32 accumulate: parity mode  */
  NL_PREC_BF16 = 2,    /* single bf16 MFMA, fp32 accumulate: throughput mode (does not meet 1e-4)       */
  NL_PREC_F16MX = 3    /* round 4 / 5 — parity mode, 1.5 (round 4: 2.0) instead of 3 matrix-instruction equivalents per product in the fused neural-point kernel
                        * (the MFMA-bound kernel): fp16 hi.hi (v_mfma_f32_32x32x16_f16) + the two cross terms hi.lo / lo.hi on gfx950's
                        * block-scaled instruction v_mfma_scale_f32_32x32x64_f8f6f4 — since round 5 with FP6 (e2m3) operands, 8 passes per K = 64 where FP8 takes 16
                        * (the cross terms are 2^-11 of a product: three mantissa bits leave 2^-15, whether the element is e4m3 or e2m3; what e2m3 lacks is range,
                        * which the block scales supply).  Round 6: the ray U-Net's conv_out (W = 256, S = 128: the second-largest kernel of nl_render_rays) multiplies
                        * in the same arithmetic, its B operand rebuilt per K = 64 slab in registers with a scale per 32-value block from the values themselves.
                        * Every other GEMM-shaped stage, the stage entry points and the backward passes run exactly as NL_PREC_BF16X3.
                        * RANGE: every MX block — a row's 32 values of one K half-slab — carries its own power-of-two scale 2^(floor(log2 max) - 2), taken from the
                        * values themselves: in the kernel for the activations (per row, slab and layer; the residual image uses the same scale x 2^-11), at
                        * packing time for the weights (per output row and half-slab).  No activation or weight magnitude saturates or flushes a block.  What is
                        * left is fp16's own range for the hi part: |activation| < 65504 (beyond it the value saturates, MODE.FP16_OVFL, instead of becoming inf)
                        * and >= 2^-14 for its full 11 bits.  Validated (tests/test_gpu_parity.py, tools/scale_sweep.py): feature maps x 1/64 ... 64 (activations
                        * to ~2e3), Student-t (nu = 3) weights, DepthFusionNet maps x 8.
                        * ACCURACY: a product carries ~2^-16 (NL_PREC_BF16X3: 2^-17, NL_PREC_F32: 2^-24).  On well-conditioned inputs that is 1-2.5e-5 of the
                        * outputs; where the network amplifies rounding (attention logits of |q.k / sqrt d| >> 10: see nl_frame_diagnostics) every mode's
                        * distance to the reference grows by the same factor and NL_PREC_F32 is the mode that stays at the reference's own level.          */
} nl_precision;

typedef struct nl_config {
  int32_t W;          /* model_3d_hidden_dim: multiple of 32, 32..256                      */
  int32_t C;          /* backbone2d_fpn_dim (feature channels of the support maps), 192    */
  int32_t S;          /* samples per ray seen by ray_unet (N_samples + N_importance), S%8==0, S<=256 */
  int32_t precision;  /* nl_precision for the GEMM-shaped stages                           */
} nl_config;

/* Per-frame inputs (reference `data` dict, nerf_pose_estimator.py:255-290, plus the two module caches). */
typedef struct nl_frame_desc {
  int32_t V, H, Wimg, h, w;      /* support views, full-res image size, feature-map size (H/4 fine, H/8 coarse) */
  int32_t vis_h, vis_w;          /* size of vis_featmaps (always H/4, Wimg/4: depth_fusion.py:239-282)     */
  float near_, far_;             /* data['depth_range'][0]                                             */
  const float* images;           /* (V,3,H,Wimg)   data['topk_images']                                 */
  const float* featmaps;         /* (V,h,w,C)      data['feat_fine_src'] (channels-last, as the reference stores it) */
  const float* vis_featmaps;     /* (V,32,vis_h,vis_w) multiview_aggregator.vis_featmaps                */
  const float* proj_ibr;         /* HOST (V,3,4)   rows 0..2 of  K4 @ inv(c2w)   (ibrnet.py:183)       */
  const float* proj_neuray;      /* HOST (V,3,4)   K @ inv(c2w)[:3]             (depth_fusion.py:90)   */
  const float* cam_centers;      /* HOST (V,3)     c2w[:3,3]                    (ibrnet.py:159)        */
  int64_t M;                     /* support neural points (fine level)                                 */
  const float* sp_xyz;           /* (M,3)   support_neural_points['fine']['xyz']                       */
  const float* sp_feature;       /* (M,C+3) ...['feature']  = [rgb, feat]                              */
  const float* sp_confidence;    /* (M,1)   ...['confidence']                                          */
  const float* sp_direction;     /* (M,4)   ...['direction'] = [unit view dir, depth]                  */
} nl_frame_desc;

typedef struct nl_frame nl_frame;   /* opaque: device tables + KNN grid living in caller-provided memory */

/* Outputs of render_rays (model.py:577-598).  Any pointer may be NULL to skip that output. */
typedef struct nl_render_out {
  float* rgb;                /* (R,3)  */
  float* depth;              /* (R)    */
  float* weights;            /* (R,S)  any alignment is accepted; 16-byte aligned (or null) lets the f16mx path at W = 256 fuse feat_mlp with the
                              *        compositing pass (the library falls back to the staged feature path otherwise: same results to ~1e-6) */
  uint8_t* mask;             /* (R)    0/1 */
  float* depth_uncertainty;  /* (R)    */
  float* feat;               /* (R,C)  */
  /* optional intermediates for testing / staged callers */
  float* sigma;              /* (R*S)   */
  float* feature_agg;        /* (R*S,W) */
  float* mv_feature_agg;     /* (R*S,W) */
  float* geo;                /* (R*S,W) ray_unet output */
  int32_t* knn_idx;          /* (R*S,8) */
  float* knn_d2;             /* (R*S,8) */
} nl_render_out;

/* ---- library ---------------------------------------------------------------------------------- */
int nl_abi_version(void);
const char* nl_strerror(int status);
/* Names of the state_dict tensors nl_pack_weights consumes, in the order it expects them. */
int nl_num_weights(void);
const char* nl_weight_name(int i);

/* Measurement hook (no reference counterpart): between nl_profile_begin and nl_profile_end every launch of the dominant
 * kernel (the fused neural-point kernel, rows a9-a11) made through this library is bracketed by HIP events on its own
 * stream; nl_profile_end synchronises, returns the summed device time in ms and the number of launches, and disarms.
 * bench.py uses it for roofline.dominant_kernel. */
int nl_profile_begin(void);
int nl_profile_end(float* fused_ms, int* launches);

/* Debug facility for tests (not thread-safe, process-global): with a gap set, every buffer the library carves from a caller workspace is followed by
 * `bytes` unused bytes (the *_workspace_bytes queries grow accordingly) and the carve records the regions between the buffers; a caller that filled its
 * workspace with `pattern` before a call learns from nl_debug_check_gaps how many of those regions a kernel wrote into.  0 switches it off. */
int nl_debug_bump_gap(size_t bytes);
int nl_debug_check_gaps(int pattern, int32_t* scratch /* device, 4 bytes */, int* bad_regions /* host */, int* checked_regions /* host, may be NULL */,
                        void* stream);

/* ---- weights ---------------------------------------------------------------------------------- */
size_t nl_packed_weights_bytes(const nl_config* cfg);
/* tensors[i] = DEVICE pointer of state_dict[nl_weight_name(i)] (fp32, contiguous, torch layout). */
int nl_pack_weights(const nl_config* cfg, const float* const* tensors, int n_tensors,
                    void* packed, size_t packed_bytes, void* stream);

/* ---- per-frame setup ---------------------------------------------------------- */
/* Workspace for the two setup calls below (stride = 1 for nl_cross_view_features). */
size_t nl_setup_workspace_bytes(int V, int H, int W, int stride);
/* DepthFusionNet's hand-made input (conditional_nerf/depth_fusion.py:150-227 depth2pts3d + get_diff_feats +
 * extract_depth_for_init, assembled at :269-278): cnn_in (V,12,H,W) = [rgb 3 | normalised inverse depth 1 | masked mean of the
 * cross-view colour difference 3 | its variance 3 | mean and variance of the inverse-depth difference 2].  imgs (V,3,H,W),
 * depths (V,H,W) metric, Ks (V,3,3), c2w (V,4,4), all fp32 device pointers; V <= 16.  Replaces ~60 framework ops over
 * (V, V*H*W, c) intermediates. */
int nl_cross_view_features(const float* imgs, const float* depths, const float* Ks, const float* c2w, int V, int H, int W,
                           float near_, float far_, float* cnn_in, void* workspace, size_t workspace_bytes, void* stream);
/* ConditionalNeRF.backproject_support_frame (conditional_nerf/model.py:203-265; get_rays utils.py:56-70): every pixel with
 * depth > 0 of the nearest-resized (H/stride, W/stride) depth map of every view becomes one row of the support tables, in the
 * reference's order (view, row, column).  feats (V,fh,fw,C) channels-last at the level's resolution.  Tables (capacity rows):
 * feature (.,3+C) = [rgb | feature], xyz (.,3) world, xyz_ref (.,3) in view 0's camera, direction (.,4) = unit viewing ray in
 * the world frame + depth.  *m_out (HOST) receives the number of rows; if it exceeds capacity nothing is written and
 * NL_ERR_WORKSPACE is returned (capacity = V*(H/stride)*(W/stride) always suffices).  Synchronises the stream once (the
 * reference synchronises once per view in nonzero()). */
int nl_backproject_support(const float* imgs, const float* feats, const float* depths, const float* Ks, const float* c2w,
                           int V, int H, int W, int fh, int fw, int C, int stride, int64_t capacity, float* feature, float* xyz,
                           float* xyz_ref, float* direction, int64_t* m_out, void* workspace, size_t workspace_bytes, void* stream);

/* Row a1 — get_rays (conditional_nerf/utils.py:56-70) and points_2d_to_rays (conditional_nerf/model.py:687-700): pixel ->
 * ray origin (camera centre) and unit direction in the world frame.  K (3,3) and c2w (4,4) are DEVICE pointers.  uv == NULL: the
 * whole H x W grid in row-major pixel order (R must be H*W); otherwise uv (R,2) = (x, y) pixel positions, truncated towards zero
 * like the reference's .long() look-up into the ray grid. */
int nl_get_rays(const float* K, const float* c2w, const float* uv, int H, int W, int64_t R, float* rays_o, float* rays_d, void* stream);

/* ---- per-frame state --------------------------------------------------------------------------- */
/* A frame keeps the DEVICE pointers of the descriptor (images, feature maps, support points: they must stay alive and unchanged
 * while the frame is used — call nl_frame_create again when the data changes) plus tables derived from them in frame_mem
 * (visibility maps repacked, KNN grid, and — built lazily on the first render with a given packed-weights blob and rebuilt when
 * that blob is re-packed — the per-point table T and the blend-projected feature maps; plus the scratch the training backward needs
 * per support point: d loss / d T (M, W) and (M, align32(C + 3)) staging rows, used by nl_*_backward_train / nl_render_rays_backward in
 * every mode but fp32).  A frame also owns the side stream (+ two
 * events) on which nl_render_rays runs the exact KNN beside the multi-view gather: created here, destroyed by nl_frame_destroy —
 * render calls never create streams or events (safe inside hipGraph capture from the first call on).  One frame must not be rendered
 * from two host threads / caller streams at the same time (like the reference module, a frame is not re-entrant); different frames are
 * independent. */
size_t nl_frame_bytes(const nl_config* cfg, const nl_frame_desc* desc);
int nl_frame_create(const nl_config* cfg, const nl_frame_desc* desc, void* frame_mem, size_t frame_bytes,
                    void* stream, nl_frame** out);
int nl_frame_destroy(nl_frame* frame);

/* Conditioning indicators of what has been rendered against `frame` so far (round 5), copied to HOST memory; synchronises `stream`.
 *   [NL_DIAG_TABLE_ABSMAX]  max |T| over the per-frame table T = support features x base_mlp.0's feature columns + bias (0 until the first render builds it)
 *   [NL_DIAG_LOGIT_ABSMAX]  largest |attention logit| (q.k / sqrt d_k, ibrnet.py:28-45) the neural-point branch has scored since nl_frame_create (round 6: every kernel
 *                           that scores logits reports it — the fused kernels of W = 128 / 256 and W = 64 and the staged attention kernel; a NaN logit is recorded as +inf).  The softmax over a sample's 8 neighbours turns a logit error e into a weight error ~e, and a split
 *                           product's logit error is its relative precision x |logit|: at |logit| ~ 100 NL_PREC_F16MX (2^-16) sits ~2e-4 from the fp64 result where
 *                           it sits 2e-5 at |logit| ~ 1 — and the reference's own fp32 arithmetic moves from 1e-6 to 1e-5.  The host mirror uses it to fall back to
 *                           a more exact mode (nerf_loc_amd.conditional_nerf: precision_guard).
 *   [NL_DIAG_POINT_KERNEL_GHZ]  the clock the chip ran the last fused neural-point launch at (the roofline's peak assumes 2.4 GHz; under matrix load it is ~1.7)
 * n: how many of the NL_DIAG_COUNT values to write. */
#define NL_DIAG_TABLE_ABSMAX 0
#define NL_DIAG_LOGIT_ABSMAX 1
#define NL_DIAG_POINT_KERNEL_GHZ 2   /* shader clock of the last fused neural-point launch (workgroup 0: s_memtime cycles / s_memrealtime): DVFS under matrix load */
#define NL_DIAG_GUARD_PRECISION 3    /* the nl_precision the last NL_RENDER_PRECISION_GUARD call against this frame produced its outputs in (-1: no guarded call yet) */
#define NL_DIAG_GUARD_ESCALATIONS 4  /* how many times a guarded call moved this frame to a more exact mode (0, 1 or 2 over a frame's life) */
#define NL_DIAG_COUNT 5
int nl_frame_diagnostics(const nl_frame* frame, float* host_out, int32_t n, void* stream);

/* ---- stages (each is also reachable through nl_render_rays) -------------------------------------- */
/* a8: exact K nearest support points (squared L2 in the reference's fp32 order, ascending, ties by index). */
int nl_knn(const nl_frame* frame, const float* xyz, int64_t N, int K, int32_t* idx, float* d2, void* stream);

/* a2-a3: z_vals (R,S) [given, or generated from near/far when z_vals_in==NULL] -> xyz (R*S,3), z_out (R,S). */
int nl_sample_points(const float* rays_o, const float* rays_d, int64_t R, int S, float near_, float far_,
                     const float* z_vals_in, float* z_out, float* xyz, void* stream);

size_t nl_mv_aggregate_workspace_bytes(const nl_config* cfg, int V, int64_t N);
/* a4-a7 (+ a15's per-view part): -> mv_feat (N,W), valid_s (N) = #views(in-bounds & in front) > 1, and optionally
 *   rgb_feat (N*V,196) [cols 0..194 = raw multi-view projection], vis_ang (N*V,8) = [vis, angle(4), pad]   (staged callers), and/or
 *   blend1 (N*V,32) = per-(sample,view) part of rgb_blending_mlp layer 1 (pre-activation, bias included), rgbv (N*V,4) = [r,g,b,vis]
 *   (what nl_heads_composite consumes).  query_center = data['pose'][:3,3] (HOST, 3 floats). */
int nl_mv_aggregate(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center,
                    const float* xyz, int64_t N, float* mv_feat, float* rgb_feat, float* vis_ang, int32_t* valid_s,
                    float* blend1, float* rgbv, void* ws, size_t ws_bytes, void* stream);

size_t nl_point_mlp_workspace_bytes(const nl_config* cfg, int64_t N);
/* a8-a12: xyz (N,3), dir (N,3) viewing direction per sample (NULL: nearest neighbour's, model.py:391-392),
 *         mv_feat (N,W) -> feature_agg (N,W); optional knn outputs. */
int nl_point_mlp(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* xyz, const float* dir,
                 int64_t dir_stride, const float* mv_feat, int64_t N, int K, float* feature_agg, int32_t* knn_idx, float* knn_d2,
                 void* ws, size_t ws_bytes, void* stream);

size_t nl_ray_unet_workspace_bytes(const nl_config* cfg, int64_t R);
/* a13: x (R*S,W) sample-major -> geo (R*S,W). */
int nl_ray_unet(const nl_config* cfg, const void* packed, const float* x, int64_t R, float* geo,
                void* ws, size_t ws_bytes, void* stream);

size_t nl_heads_composite_workspace_bytes(const nl_config* cfg, int V, int64_t R);
/* a14-a18 */
int nl_heads_composite(const nl_config* cfg, const void* packed, int V, const float* z_vals, const float* feature_agg,
                       const float* geo, const float* blend1, const float* rgbv, const int32_t* valid_s,
                       int64_t R, int white_bkgd, const nl_render_out* out, void* ws, size_t ws_bytes, void* stream);

/* a20 (hierarchical): coarse NeuRay weights (R,Sc) for z_coarse (R,Sc) along un-normalised K^-1[u,v,1] rays, Sc <= 64.
 * query_w2c_kinv HOST 12+9 floats: inv(pose)[:3] (3x4 row-major) then inv(K) (3x3). */
size_t nl_coarse_weights_workspace_bytes(int V, int64_t R, int Sc);
int nl_coarse_weights(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_w2c_kinv,
                      const float* pixel_coordinates, const float* z_coarse, int64_t R, int Sc, float* weights,
                      float* depth_coarse, void* ws, size_t ws_bytes, void* stream);
/* inverse-CDF sampling with caller-provided uniforms u (R,Ni), merged with z_base (R,Sb) and sorted -> z_out (R,Sb+Ni). */
int nl_sample_pdf(const float* z_coarse, const float* weights_coarse, int Sc, const float* u, int Ni,
                  const float* z_base, int Sb, int64_t R, float* z_out, void* stream);

/* ---- the fused path ----------------------------------------------------------------------------- */
size_t nl_render_rays_workspace_bytes(const nl_config* cfg, int V, int64_t R);   /* recommended size */
size_t nl_render_rays_min_workspace_bytes(const nl_config* cfg, int V);          /* one-ray-chunk minimum */
/* Options of nl_render_rays_ex.  early_term_eps > 0 enables early-termination compositing (BASELINE.json config 5; the reference
 * has no such switch): along every ray the colour blend (model.py:528-538), feat_mlp (model.py:595) and the compositing reads of
 * the samples behind the point where the transmittance T (model.py:549-552) falls below eps are skipped; their total compositing
 * weight is < eps, so rgb / feat change by < eps * max|value| (weights, depth, depth_uncertainty and mask are computed from all
 * samples and do not change at all).  0 = off = nl_render_rays.  The density itself cannot be skipped: the ray U-Net
 * (ray_unet.py:55-69) runs along the whole ray. */
typedef struct nl_render_opts {
  float early_term_eps;      /* 0 (off) or in (0, 1): e.g. 1e-5 keeps BASELINE's 1e-4 with a wide margin; anything else (NaN too) is NL_ERR_BAD_ARG */
  uint32_t flags;            /* NL_RENDER_* bits below; unknown bits are NL_ERR_BAD_ARG */
  /* Several query frames per launch: device pointer to (R, 3) per-ray query camera centres, or NULL.  When set
   * it replaces `query_center` (which may then be NULL): rays of different query poses against the SAME support frame go down in one
   * call — the query centre is the only per-query-frame quantity the ray path reads (ibrnet.py:144-167, the view-angle features) —
   * which amortises the launch chain for small per-frame batches (PoseOptimizer-sized: 512 rays). */
  const float* ray_centers;
  int32_t reserved[4];       /* must be 0 (checked: NL_ERR_BAD_ARG otherwise) */
} nl_render_opts;
/* nl_render_opts.flags */
#define NL_RENDER_NO_SIDE_STREAM 1u   /* keep every kernel on the caller's stream: no fork of the exact KNN onto the frame's side stream
                                       * (results are bit-identical either way; for profiling one kernel at a time and for callers that
                                       * must not see a second stream) */
#define NL_RENDER_PRECISION_GUARD 2u  /* round 6 (ABI 7) — the precision guard AT the boundary: after the batch the library reads the frame's conditioning indicator
                                       * (NL_DIAG_LOGIT_ABSMAX: one 4-byte device-to-host copy; the call SYNCHRONISES `stream`) and, while it lies beyond the range the
                                       * mode of the outputs was validated to — NL_GUARD_LOGIT_LIMIT_* below, DESIGN.md 2.3 — renders THIS batch again in the next more
                                       * exact mode (NL_PREC_F16MX -> NL_PREC_BF16X3 -> NL_PREC_F32; NL_PREC_BF16 is a throughput mode and is left alone).  The frame
                                       * then stays in that mode for every later guarded call (until nl_frame_create), so the second pass is paid once per frame;
                                       * NL_DIAG_GUARD_PRECISION / NL_DIAG_GUARD_ESCALATIONS say what happened.  A NaN logit counts as beyond every range.  Every batch
                                       * of a frame is checked, not only the first (the indicator is cumulative: a later batch with larger logits escalates when it
                                       * arrives).  Not capturable into a HIP graph (the synchronisation); NL_ERR_BAD_ARG in nl_render_rays_multi with more than one job. */
#define NL_RENDER_FLAGS_ALL 3u
#define NL_GUARD_LOGIT_LIMIT_F16MX 50.0f    /* tools/scale_sweep.py (profiles/r5_scale_sweep.txt): f16mx <= 5.1e-5 of the CPU oracle up to |logit| 64, 7.2e-5 at 95, 9.3e-5 at 142 */
#define NL_GUARD_LOGIT_LIMIT_BF16X3 500.0f  /* bf16x3 <= 4.2e-5 up to |logit| 475, 0.3-1.7e-4 at ~1000 */

/* rays_o, rays_d (R,3); z_vals (R,S) or NULL to generate linspace(near,far,S) (model.py:451-458,483-484). */
int nl_render_rays(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center,
                   const float* rays_o, const float* rays_d, const float* z_vals, int64_t R, int white_bkgd,
                   const nl_render_out* out, void* ws, size_t ws_bytes, void* stream);

/* nl_render_rays with options (opts == NULL: identical to nl_render_rays). */
int nl_render_rays_ex(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center,
                      const float* rays_o, const float* rays_d, const float* z_vals, int64_t R, int white_bkgd,
                      const nl_render_out* out, void* ws, size_t ws_bytes, void* stream, const nl_render_opts* opts);

/* Several FRAMES per call (replaces the reference's per-frame loop nerf_pose_estimator.py:289-290 + model.py:615-639 when a pose is scored
 * against several retrieved support sets, or several query images are rendered at once): job i renders its rays against its own nl_frame — its own support
 * views, tables and workspace — exactly as nl_render_rays_ex(cfg, packed, job.frame, ...) would (bit-identical outputs), but the jobs' launch chains are
 * issued on library-owned streams forked from `stream` and joined back into it, so that PoseOptimizer-sized batches (512 rays: a fifth of the chip) of
 * different frames fill the machine together.  Jobs that share a frame run one after the other on one stream.  The first call ON A DEVICE creates that device's
 * streams (one pool per device, keyed by hipGetDevice(): a process may render on several GPUs; that call is not capturable into a HIP graph, later ones are).  Jobs
 * of different frames run concurrently: their workspaces must not overlap (NL_ERR_BAD_ARG otherwise); jobs of one frame may share one.  A single launch over frames with different tables would need per-ray table pointers in every kernel; it is
 * not what this does. */
typedef struct nl_render_job {
  const nl_frame* frame;
  const float* query_center;   /* HOST (3), or NULL with opts->ray_centers */
  const float* rays_o; const float* rays_d; const float* z_vals;   /* as nl_render_rays */
  int64_t R;
  const nl_render_out* out;
  void* ws; size_t ws_bytes;   /* this job's workspace (jobs run concurrently: no sharing) */
  const nl_render_opts* opts;  /* or NULL */
} nl_render_job;
int nl_render_rays_multi(const nl_config* cfg, const void* packed, const nl_render_job* jobs, int32_t njobs, int32_t white_bkgd, void* stream);

/* ---- backward kernels, first slice ---------------------------------------------------- */
/* Gradient of the alpha compositing of nl_render_rays / nl_heads_composite (conditional_nerf/model.py:544-560, 597; what autograd
 * does for torch.cumprod & co. in the reference) w.r.t. its per-sample inputs.  The forward pass saves nothing: the transmittance is
 * recomputed from z_vals and sigma.  Inputs as in the forward: z_vals (R,S), sigma (R,S), rgb_s (R,S,3) per-sample colours, ft (R,S,C) the
 * per-sample rows that are composited into `feat` (or NULL).  Incoming gradients (each may be NULL = zero): g_rgb (R,3), g_depth (R),
 * g_unc (R) [depth_uncertainty], g_feat (R,C), g_weights (R,S).  Outputs: g_sigma (R,S); g_rgb_s (R,S,3) and g_ft (R,S,C) may be NULL. */
int nl_composite_backward(const float* z_vals, const float* sigma, const float* rgb_s, const float* ft, int64_t R, int S, int C, int white_bkgd,
                          const float* g_rgb, const float* g_depth, const float* g_unc, const float* g_feat, const float* g_weights, float* g_sigma,
                          float* g_rgb_s, float* g_ft, void* stream);
/* Backward of nl_knn's squared distances — replaces KNearestNeighborBackwardKernel, ops/knn/src/knn.cu:449-490 (CPU: knn_cpu.cpp:68-117):
 * g_xyz (N,3) = sum_k 2 g_d2[n,k] (xyz[n] - sp_xyz[idx[n,k]]); g_sp_xyz (M,3) or NULL receives the opposite sign by atomic adds and must be
 * zero-initialised by the caller (like the reference's at::zeros).  Padded neighbour slots (k >= M) are skipped. */
int nl_knn_backward(const float* xyz, const float* sp_xyz, const int32_t* idx, const float* g_d2, int64_t N, int K, int64_t M, float* g_xyz,
                    float* g_sp_xyz, void* stream);
/* Input gradient of nl_point_mlp (rows a8-a12; conditional_nerf/model.py:372-427, ibrnet.py:89-119) with FROZEN weights and a frozen support
 * table — the part of the backward pass PoseOptimizer needs (pose_optimizer.py:131-168 optimises the pose only) and the input-gradient half
 * of a training step: g_feature_agg (N,W) -> g_xyz (N,3) [through the positional encoding of the neighbour offsets], g_dir (N,3) [through
 * ray_diff_fc; NULL when dir is NULL] and g_mv_feat (N,W) [attention query + residual; may be NULL].  Nothing is saved by the forward call:
 * the staged forward is re-run into the workspace (KNN, neighbour encoding, base_mlp, k / v / q projections, attention), then transposed-
 * weight GEMMs and the reductions' derivatives walk back.  Arguments as nl_point_mlp (dir: one row per sample, row stride dir_stride).
 * The neighbour weights' dependence on the distances is exactly zero for this network (the normalised weights multiply K identical rows)
 * and is not propagated.  knn_idx / knn_d2 (N,K): the neighbours nl_point_mlp returned for the same xyz (both or neither; NULL: the search runs again).
 * Samples are processed in chunks that fit the workspace. */
size_t nl_point_mlp_backward_workspace_bytes(const nl_config* cfg, int64_t N);
int nl_point_mlp_backward(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* xyz, const float* dir, int64_t dir_stride,
                          const float* mv_feat, int64_t N, int K, const int32_t* knn_idx, const float* knn_d2, const float* g_feature_agg, float* g_xyz,
                          float* g_dir, float* g_mv_feat, void* ws, size_t ws_bytes, void* stream);

/* ---- training: weight gradients (compute_render_loss, conditional_nerf/model.py:641-685) ----------------------------
 * The *_train variants of the backward entry points do everything their frozen-weight counterparts do and, in the same pass over the
 * recomputed activations, ADD the gradients of the stage's parameters (and of the per-frame tables the stage reads) into caller-owned
 * buffers: the caller zero-fills them once per step, every call — and every workspace chunk inside a call — accumulates.  Weight
 * gradients are reduced in a fixed order (split-K partial tiles, no atomics); the table gradients are scatter-adds (atomics, like
 * index_add / grid_sample's backward in the reference's autograd).  `packed` must hold the CURRENT weights (nl_pack_weights after every
 * optimizer step: ~0.5 ms).  The weight-gradient products read their operands as 16-byte rows: nl_config.C must be a multiple of 4 here
 * (NL_ERR_UNSUPPORTED otherwise; the forward and the frozen-weight gradients take any C <= 192). */
typedef struct nl_train_grads {
  float* const* weights;     /* HOST array of nl_num_weights() DEVICE pointers, tensor i laid out like state_dict[nl_weight_name(i)];
                              * NULL entries (and a NULL array) = that gradient is not wanted */
  float* support_feature;    /* (M, C+3) gradient of the support table's features (knn_gather's backward), or NULL */
  float* feat_maps;          /* (V, h, w, C) gradient of the support feature maps (nl_frame_desc.featmaps, channels-last), or NULL */
  float* vis_featmaps;       /* (V, vh, vw, 32) CHANNELS-LAST gradient of the DepthFusionNet maps (nl_frame_desc.vis_featmaps is (V,32,vh,vw)), or NULL */
  float* blend_feat_maps;    /* (V, h, w, 32) gradient of the blend-projected feature maps P = featmaps . rgb_blending_mlp.0.weight[:, W+3 : W+3+C]^T
                              * (the blend taps P instead of projecting every tap, model.py:532-535 by linearity): the caller continues with
                              * d weight[:, W+3 : W+3+C] += sum P_grad^T featmaps and d featmaps += P_grad . weight[:, W+3 : W+3+C].  Or NULL */
  void* scratch;             /* split-K partial tiles: nl_train_scratch_bytes(cfg) */
  size_t scratch_bytes;
  int32_t reserved[4];       /* must be 0 */
} nl_train_grads;
size_t nl_train_scratch_bytes(const nl_config* cfg);
/* nl_point_mlp_backward + gradients of ray_diff_fc.*, base_mlp.*, base_mlp_attn.{w_qs, w_ks, w_vs, fc, layer_norm}.* and of the support
 * features.  (base_mlp_agg_weight's gradient is identically zero: its softmax runs over K identical rows, model.py:415-427.) */
size_t nl_point_mlp_backward_train_workspace_bytes(const nl_config* cfg, int64_t N);
int nl_point_mlp_backward_train(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* xyz, const float* dir, int64_t dir_stride,
                                const float* mv_feat, int64_t N, int K, const int32_t* knn_idx, const float* knn_d2, const float* g_feature_agg,
                                float* g_xyz, float* g_dir, float* g_mv_feat, const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream);

/* nl_mv_aggregate_backward + gradients of multiview_aggregator.out_fc.*, the four dist_decoder MLPs (24 tensors), the support feature maps
 * (grid_sample's backward: scatter-add of the taps) and the DepthFusionNet maps. */
size_t nl_mv_aggregate_backward_train_workspace_bytes(const nl_config* cfg, int V, int64_t N);
int nl_mv_aggregate_backward_train(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* xyz, int64_t N, const float* g_mv_feat,
                                   float* g_xyz, const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream);
/* nl_blend_backward + gradients of rgb_blending_mlp.* (layer 1: every column except the feature columns, see blend_feat_maps), the decoders,
 * the DepthFusionNet maps and the blend-projected feature maps. */
size_t nl_blend_backward_train_workspace_bytes(const nl_config* cfg, int V, int64_t N);
int nl_blend_backward_train(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center, const float* xyz,
                            const float* feature_agg, int64_t N, const float* g_rgb_s, float* g_xyz, float* g_feature_agg, float* g_query_center,
                            const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream);

/* The whole ray path backwards in ONE call (ConditionalNeRF.render_rays, model.py:472-600, as PoseOptimizer — pose_optimizer.py:131-168 — and a training
 * step — model.py:641-685 — differentiate it): cotangents of the per-ray outputs -> gradients of the rays (and, with `grads`, of all 84 parameter tensors,
 * the maps and the support features).  Equivalent to chaining the stage entry points above through compositing and the three heads, but every
 * per-frame / per-sample quantity is recomputed ONCE per call: one pass of the visibility decoders forward and one backward for the aggregation's and
 * the blend's uses of them, one geometry kernel for both sets of taps, one neighbour search.  z_vals (R,S): the sample depths the forward call used
 * (the hierarchical branch's resampled depths are constants: model.py:495 detaches them).  The query camera centre comes from the host (3 floats) or —
 * no device-to-host copy of a pose that is being optimised on the device — as per-ray rows in device memory (like nl_render_opts.ray_centers).
 * g_query_center_rows (R,3) or NULL: per-ray partial sums of
 * d/d(query camera centre) — the caller adds them up.  Rays are processed in chunks that fit the workspace. */
typedef struct nl_render_cotangents {
  const float* g_rgb;                 /* (R,3) or NULL (= zero) */
  const float* g_depth;               /* (R) */
  const float* g_depth_uncertainty;   /* (R) */
  const float* g_feat;                /* (R,C) */
  const float* g_weights;             /* (R,S) */
  const int32_t* knn_idx;             /* (R*S,8) and */
  const float* knn_d2;                /* (R*S,8): the neighbours the forward call returned for the same rays (nl_render_out.knn_idx / knn_d2), both or neither;
                                       * NULL: the search runs again */
  const void* reserved[1];            /* must be NULL */
} nl_render_cotangents;
size_t nl_render_rays_backward_workspace_bytes(const nl_config* cfg, int V, int64_t R, int train);
int nl_render_rays_backward(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center /* HOST, 3 floats, or NULL */,
                            const float* ray_centers /* DEVICE (R,3) per-ray query centres, or NULL: one of the two */,
                            const float* rays_o, const float* rays_d, const float* z_vals, int64_t R, int white_bkgd, const nl_render_cotangents* g,
                            float* g_rays_o, float* g_rays_d, float* g_query_center_rows, const nl_train_grads* grads, void* ws, size_t ws_bytes,
                            void* stream);

/* The gradient path's forward and backward as a PAIR sharing one workspace: nl_render_rays_forward_keep runs the STAGED forward (the arithmetic of the
 * backward pass's recompute: exact fp32 / split-FP16), writes the per-ray outputs and leaves every intermediate in `ws`; nl_render_rays_backward_kept walks
 * back from them — no recompute, no second neighbour search.  The caller keeps `ws` untouched in between and passes the same rays / query centre / `train`
 * flag to both.  The whole batch is one chunk: NL_ERR_WORKSPACE when it does not fit (then: nl_render_rays + nl_render_rays_backward, which chunk).
 * `out`: rgb, depth, weights, mask, depth_uncertainty required, feat optional (without it feat_mlp is not evaluated and g_feat must be NULL), the
 * optional intermediates are ignored.
 * beta (may be NULL): the training-mode uncertainty head (model.py:98,111,587-592; `render.use_render_uncertainty`, the reference configs' default):
 * beta[r] = sum_s w_s softplus(beta_mlp.0(geo_s)) + beta_min.  beta_mlp is not one of the nl_pack_weights tensors (eval-mode rendering never evaluates it):
 * its two tensors, the output, its cotangent and its gradients travel in this block; the same block goes to both calls of the pair. */
typedef struct nl_beta_head {
  const float* weight;   /* beta_mlp.0.weight (W) */
  const float* bias;     /* beta_mlp.0.bias (1) */
  float beta_min;        /* 0.1 (model.py:98) */
  float* beta;           /* forward call: (R) output */
  const float* g_beta;   /* backward call: (R) cotangent, or NULL */
  float* g_weight;       /* backward call: (W) +=, or NULL */
  float* g_bias;         /* backward call: (1) +=, or NULL */
} nl_beta_head;
size_t nl_render_rays_keep_workspace_bytes(const nl_config* cfg, int V, int64_t R, int train);
int nl_render_rays_forward_keep(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center /* HOST or NULL */,
                                const float* ray_centers /* DEVICE (R,3) or NULL */, const float* rays_o, const float* rays_d, const float* z_vals, int64_t R,
                                int white_bkgd, const nl_render_out* out, const nl_beta_head* beta, int train, void* ws, size_t ws_bytes, void* stream);
int nl_render_rays_backward_kept(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center /* HOST or NULL */,
                                 const float* ray_centers /* DEVICE (R,3) or NULL */, const float* rays_d, int64_t R, int white_bkgd,
                                 const nl_render_cotangents* g, const nl_beta_head* beta, float* g_rays_o, float* g_rays_d, float* g_query_center_rows,
                                 const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream);

/* nl_ray_unet_backward + gradients of the seven blocks' convolution weights / biases and LayerNorm([C, L]) tables (28 tensors). */
size_t nl_ray_unet_backward_train_workspace_bytes(const nl_config* cfg, int64_t R);
int nl_ray_unet_backward_train(const nl_config* cfg, const void* packed, const float* x, int64_t R, const float* g_geo, float* g_x,
                               const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream);

/* Input gradient of nl_mv_aggregate's feature rows (rows a4-a7; multiview_aggregator.py:156-222, ibrnet.py:169-231, visibility_decoder.py:64-148)
 * with frozen weights and frozen support maps: g_mv_feat (N,W) -> g_xyz (N,3).  The forward is recomputed in exact fp32; the way back goes through
 * out_fc (transposed-weight products, ELU), the visibility-weighted statistics, the bilinear taps' spatial derivative (zeros padding,
 * align_corners = True), the projection, and — for the visibility weights and the depth difference — the NeuRay decoders, the border-mode tap of
 * the visibility map and the NeuRay projection. */
size_t nl_mv_aggregate_backward_workspace_bytes(const nl_config* cfg, int V, int64_t N);
int nl_mv_aggregate_backward(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* xyz, int64_t N, const float* g_mv_feat,
                             float* g_xyz, void* ws, size_t ws_bytes, void* stream);
/* Input gradient of nl_ray_unet (row a13; ray_unet.py:55-69) with frozen weights: g_geo (R*S,W) -> g_x (R*S,W).  The unfused forward is re-run in exact
 * fp32 (every layer's pre-LayerNorm output stays in the workspace); then, layer by layer, the LayerNorm([C,L]) / ELU / MaxPool1d derivative and the
 * transposed-weight convolution, with the skip connections' gradients added where the concatenations were. */
size_t nl_ray_unet_backward_workspace_bytes(const nl_config* cfg, int64_t R);
int nl_ray_unet_backward(const nl_config* cfg, const void* packed, const float* x, int64_t R, const float* g_geo, float* g_x, void* ws, size_t ws_bytes,
                         void* stream);
/* a15 as a stage (model.py:528-538): per-sample colours rgb_s (N,3) = softmax-over-views blend of the tapped colours, from the sample positions and
 * feature_agg (N,W) (query_center HOST, 3 floats) — what nl_render_rays computes between the neural-point branch and the compositing — and its
 * input gradient: g_rgb_s (N,3) -> g_xyz (N,3), g_feature_agg (N,W) or NULL, g_query_center (N,3: per-sample contributions, the caller sums
 * them) or NULL.  One workspace size serves both. */
size_t nl_blend_workspace_bytes(const nl_config* cfg, int V, int64_t N);
int nl_blend(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center, const float* xyz, const float* feature_agg, int64_t N,
             float* rgb_s, void* ws, size_t ws_bytes, void* stream);
int nl_blend_backward(const nl_config* cfg, const void* packed, const nl_frame* frame, const float* query_center, const float* xyz, const float* feature_agg,
                      int64_t N, const float* g_rgb_s, float* g_xyz, float* g_feature_agg, float* g_query_center, void* ws, size_t ws_bytes, void* stream);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif
#ifdef __cplusplus
}
#endif
#endif /* NERFLOC_RENDER_H */
