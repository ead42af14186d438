// C-ABI of libnerfloc_render.so: weight packing, per-frame state, stage entry points and the chunked
// render_rays orchestrator (include/nerfloc_render.h documents which reference code each replaces).
#include <stddef.h>
#include <stdlib.h>
#include <string.h>
#include <new>
#include <vector>
#include <mutex>
#include <unordered_map>
#include "common.h"

// ---- launchers implemented in the other translation units -------------------------------------------
struct NlKnnGrid {
  NlGridParams* params; int* starts; int* counts; int* cursor; int* cell_of; float4* sorted; int M;
};
size_t nl_knn_grid_bytes(int64_t M);
int nl_knn_grid_build(NlKnnGrid* g, void* mem, const float* xyz, int64_t M, hipStream_t st);
int nl_knn_search(const NlKnnGrid* g, const float* xyz, int64_t N, int K, int* idx, float* d2, hipStream_t st);
int nl_launch_chw_to_hwc(const float* src, float* dst, int V, int Cc, int HW, hipStream_t st);
int nl_launch_mv_vis(const NlViews& vw, const float* visf_hwc, const float* dec_w, const float* xyz, int64_t N, float* vis_out, float* dd_out, hipStream_t st);
size_t nl_mv_decoder_pack_bytes();
int nl_pack_mv_decoder(const float* dec_valu_layout, void* out, hipStream_t st);
int nl_launch_mv_vis_mfma(const NlViews& vw, const float* visf_hwc, const void* dpack, const float* xyz, int64_t N, float* vis_out,
                          float* dd_out, bool x3, hipStream_t st);
int nl_launch_mv_stats(const NlViews& vw, const float* viewsdev, const float* images, const float* feat, int C, const float* xyz, int64_t N, const float* vis_in,
                       const float* dd_in, float* g393, int ldg, float* rgb_feat, float* vis_ang, int* valid_s, const float* pfeat, const float* blw,
                       float* bl1, float* rgbv, hipStream_t st);
int nl_launch_point_encode(const float* xyz, const float* dir, int dir_stride, int dir_div, int64_t N, int K, int64_t M, const int* idx, const float* d2,
                           const float* sp_xyz, const float* sp_feat, int F, const float* sp_conf, const float* sp_dir, const float* rd_w,
                           float inv_span, float* X, int ldx, float* wscale, hipStream_t st);
int nl_launch_attn(const float* Q, const float* KV, int64_t N, int K, float* O, hipStream_t st, unsigned* logit_amax = nullptr);
int nl_launch_ln_agg(const float* FC, const float* G, int64_t N, int W, const float* gamma, const float* beta, float eps, const float* wscale, float* out, hipStream_t st);
int nl_launch_ln_slab_elu(const float* in, int64_t R, int L, int Cc, const float* gamma, const float* beta, float eps, float* out, float* pooled, hipStream_t st);
bool nl_unet_inner_supported(int S, int precision);
int nl_launch_unet_inner(const NlUnetInnerArgs& a, int precision, hipStream_t st);
size_t nl_tgemm_mx_image_bytes(int Kpad);
int nl_launch_sample_points(const float* rays_o, const float* rays_d, int64_t R, int S, float near_, float far_, const float* z_in, float* z_out, float* xyz, hipStream_t st);
int nl_launch_sigma(const float* geo, int64_t N, int W, const float* w, const float* b, float* sigma, hipStream_t st);
int nl_launch_blend(const float* hA, const float* h1, const float* rgbv, int64_t N, int V, const float* w2, const float* b2, const float* w4, const float* b4, float* rgb_s, hipStream_t st,
                    const int* n_alive = nullptr, int S = 1);
int nl_launch_blend_taps(const NlViews& vw, const float* viewsdev, const float* pfeat, const float* blw, const float* xyz, const float* hA, const float* rgbv, int64_t N,
                         const float* w2, const float* b2, const float* w4, const float* b4, float* rgb_s, hipStream_t st, const int* n_alive, int S);
size_t nl_mv_front_pack_bytes();
int nl_pack_mv_front(const float* w_outfc0, const float* b_outfc0, void* out, hipStream_t st);
bool nl_mv_front_supported(int C, int V, int64_t N);
int nl_launch_mv_front(const NlViews& vw, const float* viewsdev, const float* images, const float* feat, const float* xyz, int64_t N, const float* vis_in,
                       const float* dd_in, const void* pack, float* t64, int* valid_s, float* rgbv, hipStream_t st);
int nl_launch_termination(const float* z_vals, const float* sigma, int64_t R, int S, float eps, int* n_alive, int* tile_list, int* tile_count, hipStream_t st);
int nl_launch_coarse_weights(const NlViews& vw, const float* w2c_kinv_host, const float* visf_hwc, const float* dec_w, const void* dpack,
                             int precision, const float* pix, const float* zc, int64_t R, int Sc, float* ws_alpha, float* ws_vis, float* ws_mask,
                             float* weights, float* depth_coarse, hipStream_t st);
int nl_launch_sample_pdf(const float* zc, const float* wc, int Sc, const float* u, int Ni, const float* zb, int Sb, int64_t R,
                         float* z_out, hipStream_t st);
int nl_launch_composite(const float* z_vals, const float* sigma, const float* rgb_s, const float* ft, const int* valid_s, int64_t R, int S, int C,
                        int white_bkgd, const nl_render_out* out, int64_t ray0, float* feat_dst, float* wsum_dst, hipStream_t st, const int* n_alive = nullptr,
                        float* w_scratch = nullptr);

size_t nl_point_stream_bytes(int W);
int nl_pack_point_stream(const float* w1, const float* w2, const float* w3, const float* wk, const float* wv, void* out, int W, int F, hipStream_t st);
int nl_pack_ptt(const float* w1, const float* b1, int W, int F, int Kpad, int Npad, float* B32, float* bias, hipStream_t st);
int nl_launch_wscale(const int* idx, const float* d2, const float* conf, int64_t N, int K, int64_t M, float* wscale, hipStream_t st);
bool nl_point_fused_supported(int W, int precision);
int nl_launch_point_fused(const NlPointFusedArgs& a, int W, int precision, hipStream_t st);
size_t nl_point_stream2_bytes(int W);
int nl_pack_point_stream2(const float* w1, const float* w2, const float* w3, const float* wk, const float* wv, const float* b2, const float* b3,
                          const float* rd_w, void* out, int W, int F, hipStream_t st, int mx = 0, int* mx_scratch = nullptr);
bool nl_point_fused2_supported(int W, int precision);
int nl_launch_sample_chain(const float* O, const float* T64, const float* wscale, const float* gamma, const float* beta, float eps, const void* wbase,
                           size_t off_g2, const float* bias_g2, size_t off_fc, size_t off_f0, size_t off_ba, const float* bias_f0, float* FA, float* fth,
                           float* blA, int64_t M, int precision, hipStream_t st, bool frag_out = false, bool frag_f16 = false);
int nl_launch_query_chain(const float* T64, const void* wbase, size_t off_g2, const float* bias_g2, size_t off_q, float* Q, int64_t M, int precision,
                          hipStream_t st);
int nl_launch_point_fused2(const NlPointFusedArgs& a, int W, int precision, hipStream_t st, bool mx = false, float* keep_kv = nullptr, unsigned* const* keep_mk = nullptr,
                           const float* tmax = nullptr, unsigned* logit_amax = nullptr, unsigned long long* clk = nullptr);
int nl_table_absmax(const float* x, size_t n, float* out, hipStream_t st);
bool nl_point_bwd_chain_supported(int W);
size_t nl_point_bwd_stream_bytes(int W);
int nl_pack_point_bwd_stream(const float* w1, const float* w2, const float* w3, const float* wk, const float* wv, void* out, int W, int F, hipStream_t st);
int nl_launch_point_bwd_chain(const float* gkv, const unsigned* const* mk, const void* wstream, float* gx, int64_t NK, int W, hipStream_t st, const float* q = nullptr,
                              const float* kv = nullptr, const float* go = nullptr, float* gq = nullptr);
// backward.hip: glue kernels of the neural-point branch's input gradient
int nl_launch_wgrad(const float* dY, int ldy, int M, const float* X, int ldx, int N, int64_t rows, int shift, int period, float* gW, int ldc, int cs, int co,
                    float* gb, float* scratch, size_t scratch_floats, hipStream_t st);
size_t nl_wgrad_scratch_floats(int64_t rows, int M, int N);
int nl_launch_wgrad_multi(int nsub, const float* const* dY, int ldy, int M, const float* const* X, int ldx, int N, int64_t rows, const int* shift, int period,
                          float* gW, int ldc, int cs, const int* co, float* gb, int bias_sub, float* scratch, size_t scratch_floats, hipStream_t st);
int nl_launch_colsum(const float* Y, int ldy, int64_t rows, int M, float* out, float* scratch, hipStream_t st);
int nl_launch_sp_feat_scatter(const float* gXF, int ld, int F, const int* idx, int64_t N, int K, int64_t M, float* g_sp_feat, hipStream_t st);
int nl_launch_ln_agg_backward(const float* FC, const float* G, const float* gy, int64_t N, int W, const float* gamma, float eps, const float* wscale, float* gx,
                              float* aff, hipStream_t st);
int nl_launch_attn_backward(const float* Q, const float* KV, const float* gO, int64_t N, int K, float* gQ, float* gKV, hipStream_t st);
int nl_launch_lrelu_mask(float* g, const float* h, size_t n, hipStream_t st);
int nl_launch_add(const float* a, const float* b, float* o, size_t n, hipStream_t st);
int nl_launch_mv_geom_backward(const NlViews& vw, const float* viewsdev, const float* images, const float* feat, int C, const float* pfeat, const float* xyz,
                               int64_t N, const float* vis_in, const float* dd_in, const float* g393, int ldg, const float* g_pf, const float* g_rgbv,
                               const float* g_ang, float* g_xyz, float* g_qc, float* g_vis, float* g_dd, float* sc_feat, float* sc_pfeat, const float* stats,
                               hipStream_t st);
int nl_dec_train_row(void);
size_t nl_dec_wpart_floats(void);
int nl_launch_copy_rows(const float* src, int lds, float* dst, int ldd, int64_t rows, int cols, bool add, hipStream_t st);
int nl_launch_dec_backward(const NlViews& vw, const float* visf_hwc, const float* dec_w, const void* dpack, const float* xyz, int64_t N, const float* g_vis,
                           const float* g_dd, float* part, float* g_xyz, float* tr, float* const* decw, float* scratch, size_t scratch_floats, float* sc_vis,
                           hipStream_t st);
int nl_launch_blend_backward(const float* hA, const float* h1, const float* rgbv, int64_t N, int V, const float* w2, const float* b2, const float* w4,
                             const float* b4, const float* blw, const float* g_rgb_s, float* g_hA, float* g_pf, float* g_rgbv, float* g_ang, float* tr,
                             hipStream_t st);
int nl_launch_blend_inputs8(const NlViews& vw, const float* viewsdev, const float* xyz, int64_t N, const float* rgbv, float* x8, hipStream_t st);
int nl_launch_blw_unpack(const float* t, float* g, int W, int F, hipStream_t st);
int nl_launch_elu_mask(float* g, const float* e, size_t n, hipStream_t st);
int nl_launch_ln_slab_elu_backward(const float* x, int64_t R, int L, int Cc, const float* gamma, const float* beta, float eps, const float* g_out, int ldgo, int pool,
                                   float* g_x, float* aff, hipStream_t st);
int nl_launch_table_add_t(const float* t, float* g, int L, int Cc, hipStream_t st);
int nl_launch_colsum_tables(const float* Y, int64_t rows, int L, int Cc, float* gw, float* gb, float* scratch, hipStream_t st);
int nl_launch_ray_feat_sum(const float* z, const float* sigma, const float* ft, int64_t R, int S, int C, float* hc, float* wsum4, hipStream_t st);
int nl_launch_sigma_backward(const float* geo, int64_t N, int W, const float* w, const float* b, const float* g_sigma, float* g_geo, float* gpre4, hipStream_t st);
int nl_launch_gw_total(const float* g_wts, const float* g_feat, const float* b2, int64_t R, int S, int C, float* gw, const float* g_beta, const float* bv,
                       hipStream_t st);
int nl_launch_beta_forward(const float* wts, const float* bv, int64_t R, int S, float beta_min, float* beta, hipStream_t st);
int nl_launch_beta_backward(const float* geo, int64_t N, int S, int W, const float* wb, const float* bb, const float* wts, const float* g_beta, float* g_geo, float* gpre4,
                            hipStream_t st);
int nl_launch_ray_reduce(const float* ga, const float* gb, const float* gc, const float* g_dir, const float* g_qcN, const float* z, int64_t R, int S, float* g_o,
                         float* g_d, float* g_qc, hipStream_t st);
int nl_launch_add2d(const float* a, int lda, const float* b, int ldb, float* o, int ldo, int64_t rows, int cols, hipStream_t st);
int nl_launch_point_encode_backward(const float* xyz, const float* dir, int dir_stride, int dir_div, int64_t N, int K, int64_t M, const int* idx,
                                    const float* sp_xyz, const float* sp_dir, const float* rd_w, float inv_span, const float* gX, int ldg, float* g_xyz,
                                    float* g_dir, float* tr, hipStream_t st);

// CU count for persistent kernels, per device id (common.h)
int nl_persistent_cus() {
  static std::mutex mu;
  static int cus[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return NL_ERR_HIP;
  std::lock_guard<std::mutex> lk(mu);
  if (dev < 64 && cus[dev] > 0) return cus[dev];
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, dev) != hipSuccess) return NL_ERR_HIP;
  const int n = prop.multiProcessorCount > 8 ? prop.multiProcessorCount / 8 * 8 : 8;
  if (dev < 64) cus[dev] = n;
  return n;
}

namespace {

// A/B switches used while developing the kernels (NERFLOC_POINT_V1, NERFLOC_NO_TMERGE, NERFLOC_NO_CHAIN) exist only in a build
// with -DNERFLOC_DEBUG_SWITCHES; the shipped library reads no environment variable on the render path.
#ifdef NERFLOC_DEBUG_SWITCHES
inline bool dbg_switch(const char* name) { return getenv(name) != nullptr; }
#else
inline bool dbg_switch(const char*) { return false; }
#endif

// ------------------------------------------------------------------------------------------ weight table
const char* kWeightNames[] = {
    "ray_diff_fc.0.weight", "ray_diff_fc.0.bias", "ray_diff_fc.2.weight", "ray_diff_fc.2.bias",
#define NL_DEC(d)                                                                                                         \
  "multiview_aggregator.dist_decoder." d "_decoder.0.weight", "multiview_aggregator.dist_decoder." d "_decoder.0.bias",   \
  "multiview_aggregator.dist_decoder." d "_decoder.2.weight", "multiview_aggregator.dist_decoder." d "_decoder.2.bias",   \
  "multiview_aggregator.dist_decoder." d "_decoder.4.weight", "multiview_aggregator.dist_decoder." d "_decoder.4.bias"
    NL_DEC("mean"), NL_DEC("var"), NL_DEC("aw"), NL_DEC("vis"),
#undef NL_DEC
    "multiview_aggregator.out_fc.0.weight", "multiview_aggregator.out_fc.0.bias",
    "multiview_aggregator.out_fc.2.weight", "multiview_aggregator.out_fc.2.bias",
    "base_mlp.0.weight", "base_mlp.0.bias", "base_mlp.2.weight", "base_mlp.2.bias", "base_mlp.4.weight", "base_mlp.4.bias",
    "base_mlp_attn.w_qs.weight", "base_mlp_attn.w_ks.weight", "base_mlp_attn.w_vs.weight", "base_mlp_attn.fc.weight",
    "base_mlp_attn.layer_norm.weight", "base_mlp_attn.layer_norm.bias",
#define NL_UN(n) "ray_unet." n ".0.weight", "ray_unet." n ".0.bias", "ray_unet." n ".1.weight", "ray_unet." n ".1.bias"
    NL_UN("conv1"), NL_UN("conv2"), NL_UN("conv3"), NL_UN("trans_conv3"), NL_UN("trans_conv2"), NL_UN("trans_conv1"), NL_UN("conv_out"),
#undef NL_UN
    "sigma_mlp.0.weight", "sigma_mlp.0.bias",
    "feat_mlp.0.weight", "feat_mlp.0.bias", "feat_mlp.2.weight", "feat_mlp.2.bias",
    "rgb_blending_mlp.0.weight", "rgb_blending_mlp.0.bias", "rgb_blending_mlp.2.weight", "rgb_blending_mlp.2.bias",
    "rgb_blending_mlp.4.weight", "rgb_blending_mlp.4.bias",
};
constexpr int kNumWeights = sizeof(kWeightNames) / sizeof(kWeightNames[0]);
enum {
  T_RD0W = 0, T_RD0B, T_RD2W, T_RD2B, T_DEC = 4,  // 24 decoder tensors
  T_OUT0W = 28, T_OUT0B, T_OUT2W, T_OUT2B, T_B0W, T_B0B, T_B2W, T_B2B, T_B4W, T_B4B,
  T_WQ, T_WK, T_WV, T_FC, T_LNW, T_LNB, T_UNET = 44,  // 7 x {conv w, conv b, ln w, ln b}
  T_SIGW = 72, T_SIGB, T_F0W, T_F0B, T_F2W, T_F2B, T_BL0W, T_BL0B, T_BL2W, T_BL2B, T_BL4W, T_BL4B
};
static_assert(kNumWeights == 84, "weight table");

// ------------------------------------------------------------------------------------------ GEMM layer table
enum {
  G_OUTFC0 = 0, G_OUTFC2, G_BASE0, G_BASE2, G_BASE4, G_KV, G_Q, G_FC, G_CONV1, G_CONV2, G_CONV3,
  G_T3E, G_T3O, G_T2E, G_T2O, G_T1E, G_T1O, G_T3M, G_T2M, G_T1M, G_FEAT0P, G_BLENDAP, G_QP, G_CONVOUT, G_FEAT0, G_FEAT2, G_BLENDA, G_BLENDP, G_PTT,
  G_FC_T, G_Q_T, G_KV_T, G_BASE4_T, G_BASE2_T, G_BASE0_T,   // transposed weights: input gradients of the neural-point branch (do_point_backward)
  G_OUTFC2_T, G_OUTFC0_T, G_BLENDA_T,                         // ... of the multi-view aggregation's out_fc and of the blend's per-sample projection
  G_UB_OUTA, G_UB_OUTB, G_UB_T1, G_UB_T2, G_UB_T3, G_UB_C3, G_UB_C2, G_UB_C1,   // ... of the ray U-Net's seven convolutions (do_unet_backward)
  G_BASE0_TF,                                                                    // training: base_mlp.0 towards its support-feature columns
  G_FEAT0_T, G_FEAT2_T,                                                          // whole-path backward: feat_mlp's two layers towards their inputs
  G_BASE0_S,                                                                     // base_mlp.0's posenc + ray_diff_fc columns (the staged forward on the table T)
  G_CONV1F, G_CONVOUTF,   // conv1 / conv_out with the feature_agg channels of every 32-block in ACCUMULATOR order: their input is the chain kernel's fragment image
  G_COUNT
};
enum { U_CONV1 = 0, U_CONV2, U_CONV3, U_T3, U_T2, U_T1, U_OUT, U_COUNT };

struct GemmDim { int K, N, Kpad, Npad; bool bias; };

struct Layout {
  GemmDim g[G_COUNT];
  size_t b32[G_COUNT], bhi[G_COUNT], blo[G_COUNT], bst[G_COUNT], bsh[G_COUNT], bias[G_COUNT];   // bsh: the weight stream in fp16 hi / lo (split-FP16 arithmetic)
  size_t rd_w, dec_w, sig_w, sig_b, bl2_w, bl2_b, bl4_w, bl4_b, ln_g, ln_b;
  size_t pt_stream, pt_stream2, pt_stream2_mx, pt_stream2_f16, pt_bwd_stream, pt_mx_sc, mvf_pack, pt_bias, blw, dec_mfma, zeros;   // blw: [32][8] rgb/vis/angle columns of rgb_blending_mlp.0 + bias[32]  // fused point-branch weight stream (W in {64,128,256}) and its 3 bias rows
  size_t mx_convout;   // NL_PREC_F16MX (round 6): fp6 images + block scales of G_CONVOUTF for tgemm_mx_kernel (W = 256)
  size_t mx_feat0;     // ... and of G_FEAT0P for feat_comp_mx_kernel (feat_mlp.0 + compositing in one kernel)
  size_t un_g[U_COUNT], un_b[U_COUNT];     // LayerNorm([C, L]) affine tables, position-major (L, C)
  size_t un_gl[U_COUNT], un_bl[U_COUNT];   // the same tables in the accumulator-lane order of the GEMM that fuses the LayerNorm (un_n x un_so)
  int un_c[U_COUNT], un_l[U_COUNT], un_n[U_COUNT], un_so[U_COUNT];
  size_t total;
};

bool cfg_ok(const nl_config* c) {
  return c && c->W >= 32 && c->W <= 256 && c->W % 32 == 0 && c->C > 0 && c->C <= 192 && c->S >= 8 && c->S <= 256 && c->S % 8 == 0 &&
         c->precision >= 0 && c->precision <= 2;   // (NL_PREC_F16MX is normalised to BF16X3 + a flag at every entry point: NL_EFF_CFG)
}

Layout make_layout(const nl_config* c) {
  Layout L;
  memset(&L, 0, sizeof(L));
  const int W = c->W, C = c->C, F = C + 3, S = c->S;
  auto set = [&](int i, int K, int N, bool bias) { L.g[i] = {K, N, (int)nl_align_up(K, 32), (int)nl_align_up(N, 32), bias}; };
  set(G_OUTFC0, 2 * F + 3, 64, true);
  set(G_OUTFC2, 64, W, true);
  set(G_BASE0, F + 90, W, true);
  set(G_BASE2, W, W, true);
  set(G_BASE4, W, W, true);
  set(G_KV, W, 256, false);
  set(G_Q, W, 128, false);
  set(G_FC, 128, W, false);
  set(G_CONV1, 3 * W, 64, true);
  set(G_CONV2, 3 * 64, 128, true);
  set(G_CONV3, 3 * 128, 128, true);
  set(G_T3E, 128, 128, true);
  set(G_T3O, 256, 128, true);
  set(G_T2E, 256, 64, true);
  set(G_T2O, 512, 64, true);
  set(G_T1E, 128, 32, true);
  set(G_T1O, 256, 32, true);
  // both output phases of a stride-2 transposed convolution as ONE GEMM: K = [x[m] | x[m+1]], N = [even outputs | odd outputs]
  // (the even phase's second K half is zero: 33 % more MACs for half the launches and one pass over the activations)
  set(G_T3M, 256, 256, true);
  set(G_T2M, 512, 128, true);
  set(G_T1M, 256, 64, true);
  // feat_mlp.0 and the blend projection with K in ACCUMULATOR order, for the per-sample chain kernel (tgemm.hip: sample_chain_kernel)
  set(G_FEAT0P, W, W, true);
  set(G_BLENDAP, W, 32, false);
  set(G_QP, W, 128, false);
  set(G_CONVOUT, 3 * (W + 32), W, true);
  set(G_CONV1F, 3 * W, 64, true);
  set(G_CONVOUTF, 3 * (W + 32), W, true);
  set(G_FEAT0, W, W, true);
  // feat_mlp's last Linear is applied AFTER compositing (it is linear): K = [composited hidden (W) | sum of weights (1)],
  // the bias row multiplies the weight sum (model.py:594-597)
  set(G_FEAT2, W + 32, C, false);
  // colour-blend layer 1 split by linearity (model.py:532-535): per-sample part (feature_agg columns), and a per-frame
  // projection of the support feature maps through the feature columns (G_BLENDP, applied once per frame; the
  // per-(sample, view) value is then a bilinear tap of the projected map inside mv_stats)
  set(G_BLENDA, W, 32, false);
  set(G_BLENDP, C, 32, false);
  // per-frame neural-point table T = sp_feature . base_mlp.0.weight[:, :F]^T + bias, columns in accumulator order (point_fused.hip)
  set(G_PTT, F, W, true);
  // dX = dY . W for y = x W^T: K = the layer's outputs, N = its inputs; base_mlp.0 only towards its posenc + ray_diff_fc columns
  // (the feature columns multiply rows of the frozen support table)
  set(G_FC_T, W, 128, false);
  set(G_Q_T, 128, W, false);
  set(G_KV_T, 256, W, false);
  set(G_BASE4_T, W, W, false);
  set(G_BASE2_T, W, W, false);
  set(G_BASE0_T, W, 96, false);
  set(G_BASE0_TF, W, F, false);
  set(G_BASE0_S, 90, W, false);
  set(G_FEAT0_T, W, W, false);
  set(G_FEAT2_T, C, W, false);
  set(G_OUTFC2_T, W, 64, false);
  set(G_OUTFC0_T, 64, (int)nl_align_up(2 * F + 3, 32), false);   // = ldg_of(C): the statistics row incl. its zero padding (416 columns: generic kernels)
  set(G_BLENDA_T, 32, W, false);
  // convolution input gradients: K = the layer's output channels x 3 taps (transposed convolutions: [even | odd | odd of the previous position])
  set(G_UB_OUTA, 3 * W, W, false);   // conv_out -> its feature_agg input channels
  set(G_UB_OUTB, 3 * W, 32, false);  // conv_out -> its x2 input channels
  set(G_UB_T1, 96, 128, false);
  set(G_UB_T2, 192, 256, false);
  set(G_UB_T3, 384, 128, false);
  set(G_UB_C3, 384, 128, false);
  set(G_UB_C2, 384, 64, false);
  set(G_UB_C1, 192, W, false);
  size_t off = 0;
  auto take = [&](size_t bytes) { size_t o = off; off += nl_align_up(bytes, 256); return o; };
  for (int i = 0; i < G_COUNT; ++i) {
    const size_t n = (size_t)L.g[i].Kpad * L.g[i].Npad;
    L.b32[i] = take(n * 4);
    L.bhi[i] = take(n * 2);
    L.blo[i] = take(n * 2);
    L.bst[i] = take(L.g[i].N <= 256 ? nl_tgemm_stream_bytes(L.g[i].Kpad, L.g[i].N) : 0);
    L.bsh[i] = take(L.g[i].N <= 256 ? nl_tgemm_stream_bytes(L.g[i].Kpad, L.g[i].N) : 0);
    L.bias[i] = take((size_t)L.g[i].Npad * 4);
  }
  L.rd_w = take(4 * (64 + 16 + 27 * 16 + 27));
  L.dec_w = take(4 * 4 * 2178);
  L.sig_w = take(4 * W); L.sig_b = take(4);
  L.bl2_w = take(4 * 512); L.bl2_b = take(4 * 16); L.bl4_w = take(4 * 16); L.bl4_b = take(4);
  L.ln_g = take(4 * W); L.ln_b = take(4 * W);
  const int uc[U_COUNT] = {64, 128, 128, 128, 64, 32, W};
  const int ul[U_COUNT] = {S, S / 2, S / 4, S / 4, S / 2, S, S};
  for (int u = 0; u < U_COUNT; ++u) {
    L.un_c[u] = uc[u]; L.un_l[u] = ul[u];
    L.un_g[u] = take(4 * (size_t)uc[u] * ul[u]);
    L.un_b[u] = take(4 * (size_t)uc[u] * ul[u]);
    // the fused GEMM's view of the slab: a transposed convolution's merged launch has rows = input positions, columns = both phases
    const bool tr = u == U_T3 || u == U_T2 || u == U_T1;
    L.un_n[u] = tr ? 2 * uc[u] : uc[u];
    L.un_so[u] = tr ? ul[u] / 2 : ul[u];
    const size_t lm = 4 * (size_t)(L.un_so[u] > 32 ? L.un_so[u] : 32) * nl_tgemm_nrt(L.un_n[u]) * 32;
    L.un_gl[u] = take(lm);
    L.un_bl[u] = take(lm);
  }
  L.blw = take(4 * (256 + 32));
  L.dec_mfma = take(nl_mv_decoder_pack_bytes());
  L.pt_bias = take(4 * 3 * (size_t)W);
  L.pt_stream = take((W == 64 || W == 128 || W == 256) ? nl_point_stream_bytes(W) : 256);
  L.pt_stream2 = take((W == 128 || W == 256) ? nl_point_stream2_bytes(W) : 256);
  L.pt_stream2_mx = take((W == 128 || W == 256) ? nl_point_stream2_bytes(W) : 256);   // NL_PREC_F16MX: f16 fragments + fp8 images of layers 2, 3, k / v
  L.pt_stream2_f16 = take((W == 128 || W == 128 * 2) ? nl_point_stream2_bytes(W) : 256);  // split-FP16 stream: the gradient path's fused forward (pt_forward_keep_fused)
  L.pt_bwd_stream = take(nl_point_bwd_chain_supported(W) ? nl_point_bwd_stream_bytes(W) : 256);   // transposed weights of the branch's rows: the frozen-weight way back (point_bwd.hip)
  L.pt_mx_sc = take(4 * 64);
  L.mvf_pack = take(nl_mv_front_pack_bytes());                                          // out_fc.0 as register-resident A fragments of mv_front_kernel (C = 192)                                                            // their per-chunk scale bytes while packing
  L.zeros = take(4096);
  L.mx_convout = take(W == 256 ? nl_tgemm_mx_image_bytes(L.g[G_CONVOUTF].Kpad) : 0);
  L.mx_feat0 = take(W == 256 ? nl_tgemm_mx_image_bytes(L.g[G_FEAT0P].Kpad) : 0);
  L.total = off;
  return L;
}

// ------------------------------------------------------------------------------------------ pack kernels
__device__ __forceinline__ unsigned short pk_f2bf(float x) {
  unsigned int u = __float_as_uint(x);
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}

// dst[k0+k][n] (f32 [Kpad][Npad]) and bf16 hi/lo [n][Kpad] <- src[off + n*ld_n + k*ld_k], k < kc, n < N
__global__ void pack_block_kernel(const float* __restrict__ src, int off, int ld_n, int ld_k, int kc, int N, int k0,
                                  float* __restrict__ b32, unsigned short* __restrict__ bhi, unsigned short* __restrict__ blo,
                                  int Kpad, int Npad, unsigned short* __restrict__ bst, int nrts, int n0, int perm = 0,
                                  unsigned short* __restrict__ bsh = nullptr) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= kc * N) return;
  int k = i / N, n = i - k * N;
  // perm (k0 == 0 only): K position k of the packed matrix holds source column 32 c + m(8 ks + t, hh) for k = 32 c + 16 ks + 8 hh + t,
  // m(r, hh) = (r & 3) + 8 (r >> 2) + 4 hh — the order in which a 32x32 accumulator tile hands its rows to the next MFMA as B operand
  int ksrc = k;
  if (perm) { const int r = 8 * ((k >> 4) & 1) + (k & 7), hh = (k >> 3) & 1; ksrc = (k & ~31) + (r & 3) + 8 * (r >> 2) + 4 * hh; }
  float v = src[off + (size_t)n * ld_n + (size_t)ksrc * ld_k];
  b32[(size_t)(k0 + k) * Npad + n] = v;
  unsigned short h = pk_f2bf(v);
  float hf = __uint_as_float(((unsigned int)h) << 16);
  bhi[(size_t)n * Kpad + k0 + k] = h;
  const unsigned short l = pk_f2bf(v - hf);
  blo[(size_t)n * Kpad + k0 + k] = l;
  // weight stream of tgemm.hip: chunk (32 k) = [part hi/lo][k-step][row tile][lane = (n&31) + 32*((k>>3)&1)][k&7]
  if (!bst) return;   // (matrices wider than 256 columns have no streaming layout: generic kernels only)
  const int kk = k0 + k, ng = n0 + n;
  const size_t e = (size_t)(kk >> 5) * (4 * nrts * 512) + ((size_t)(((kk >> 4) & 1) * nrts + (ng >> 5)) * 64 + (ng & 31) + 32 * ((kk >> 3) & 1)) * 8 + (kk & 7);
  bst[e] = h;
  bst[e + (size_t)2 * nrts * 512] = l;
  if (bsh) {   // the same stream in fp16: hi = round(v), lo = round(v - hi)
    const _Float16 g = (_Float16)v;
    bsh[e] = __builtin_bit_cast(unsigned short, g);
    bsh[e + (size_t)2 * nrts * 512] = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)g));
  }
}

// MX-FP6 images of a 256-column layer for tgemm_mx_kernel (tgemm.hip): one thread = one MX block = (slab of 64 k, row tile, lane, image).  The 32 weights of output
// column 32 rt + (lane & 31) whose k-slots belong to half lane >> 5 of the slab, in the natural position order P = 8 s + t <-> k = 64 slab + 16 s + 8 hh + t (what the
// kernel's activation images have): image 0 = e2m3(f16(w)) (meets the activations' residual image), image 1 = e2m3(w - f16(w)) (meets their hi image).  Block scale
// 2^(floor(log2 max) - 2): the largest magnitude lands in [4, 8) (e2m3 saturates at 7.5).  Per slab: [rt][image][lane] dwords 0-3 (16 KB) | [rt][image][lane]
// {dword 4, dword 5, E8M0 scale, 0} (16 KB).  K rows past Kpad are zero.
__device__ __forceinline__ unsigned pk_e2m3(float a) {   // a >= 0, already divided by the block scale; round to nearest even, saturating
  if (!(a < 7.5f)) return 31u;
  if (a < 1.f) return (unsigned)rintf(a * 8.f);
  const int e = a < 2.f ? 0 : a < 4.f ? 1 : 2;
  unsigned m = (unsigned)rintf(ldexpf(a, 3 - e));
  unsigned c = ((unsigned)(e + 1) << 3) + (m - 8u);
  return c > 31u ? 31u : c;
}
__global__ void pack_tgemm_mx6_kernel(const float* __restrict__ b32, int Kpad, int Npad, int N, int nslab, unsigned char* __restrict__ out) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= nslab * 8 * 64 * 2) return;
  const int im = e & 1, lane = (e >> 1) & 63, rt = (e >> 7) & 7, sl = e >> 10;
  const int hh = lane >> 5, n = 32 * rt + (lane & 31);
  float v[32], mx = 0.f;
  for (int P = 0; P < 32; ++P) {
    const int k = 64 * sl + 16 * (P >> 3) + 8 * hh + (P & 7);
    const float w = (k < Kpad && n < N) ? b32[(size_t)k * Npad + n] : 0.f;
    const float h = (float)(_Float16)w;
    v[P] = im == 0 ? h : w - h;
    mx = fmaxf(mx, fabsf(v[P]));
  }
  int E = -60;
  if (mx > 0.f) { int ex; (void)frexpf(mx, &ex); E = ex - 1; }   // mx = 1.xxx 2^E
  int sb = E - 2 + 127;
  sb = sb < 1 ? 1 : (sb > 254 ? 254 : sb);
  const float inv = ldexpf(1.f, 127 - sb);
  unsigned d[6] = {0u, 0u, 0u, 0u, 0u, 0u};
  for (int P = 0; P < 32; ++P) {
    const unsigned c = pk_e2m3(fabsf(v[P]) * inv) | (v[P] < 0.f ? 32u : 0u);
    const int b = 6 * P;
    d[b >> 5] |= c << (b & 31);
    if ((b & 31) > 26) d[(b >> 5) + 1] |= c >> (32 - (b & 31));
  }
  unsigned char* base = out + (size_t)sl * (16384 + 16384);
  unsigned* a = reinterpret_cast<unsigned*>(base + ((size_t)(rt * 2 + im) * 64 + lane) * 16);
  a[0] = d[0]; a[1] = d[1]; a[2] = d[2]; a[3] = d[3];
  unsigned* b2 = reinterpret_cast<unsigned*>(base + 16384 + ((size_t)(rt * 2 + im) * 64 + lane) * 16);
  b2[0] = d[4]; b2[1] = d[5]; b2[2] = (unsigned)sb; b2[3] = 0u;   // (the matrix instruction reads byte 0 of the scale register)
}

// [32][8] = rgb(3) | vis(1) | angle(4) columns of rgb_blending_mlp.0.weight (32, W+F+5), then its bias[32]
__global__ void pack_blw_kernel(const float* __restrict__ w, const float* __restrict__ b, float* __restrict__ dst, int W, int F) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < 256) {
    const int j = i >> 3, c = i & 7;
    const int col = c < 3 ? W + c : W + F + (c - 3);
    dst[i] = w[(size_t)j * (W + F + 5) + col];
  } else if (i < 288) dst[i] = b[i - 256];
}

// dst[l][c] = src[c][l]: LayerNorm([C, L]) affine tables, stored position-major like the activations
__global__ void transpose_kernel(const float* __restrict__ src, float* __restrict__ dst, int Cc, int L) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= Cc * L) return;
  const int l = i / Cc, c = i - l * Cc;
  dst[i] = src[(size_t)c * L + l];
}

// LayerNorm affine table (So positions x N channels, position-major) -> the order in which tgemm_kernel's LNSLAB epilogue reads it:
// [wave of the ray][row tile][gq][lane][4]: lane (j, hh) of wave w holds position (32 w + j) % So, channels 32 rt + 8 gq + 4 hh + 0..3 — one
// contiguous KB per load instruction instead of 64 rows
__global__ void ln_lane_major_kernel(const float* __restrict__ src, float* __restrict__ dst, int N, int So, int NRT, int total) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= total) return;
  const int i = e & 3, lane = (e >> 2) & 63, gq = (e >> 8) & 3, rt = (e >> 10) % NRT, wq = e / (1024 * NRT);
  const int j = lane & 31, hh = lane >> 5, n = 32 * rt + 8 * gq + 4 * hh + i, t = (32 * wq + j) % So;
  dst[e] = n < N ? src[(size_t)t * N + n] : 0.f;
}

__global__ void copy_kernel(const float* __restrict__ src, float* __restrict__ dst, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

struct Packer {
  const float* const* t;
  char* base;
  const Layout* L;
  hipStream_t st;
  int rc = NL_OK;
  uint64_t has_bst = 0, has_bsh = 0;   // layers whose streaming images this pass wrote
  void mark(int g, bool bsh) { if (L->g[g].N <= 256) { has_bst |= 1ull << g; if (bsh) has_bsh |= 1ull << g; } }
  void block(int g, int k0, const float* src, int off, int ld_n, int ld_k, int kc, int perm = 0) {
    const GemmDim& d = L->g[g];
    mark(g, true);
    int n = kc * d.N;
    hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(n, 256)), dim3(256), 0, st, src, off, ld_n, ld_k, kc, d.N, k0,
                       (float*)(base + L->b32[g]), (unsigned short*)(base + L->bhi[g]), (unsigned short*)(base + L->blo[g]), d.Kpad, d.Npad,
                       (unsigned short*)(base + L->bst[g]), nl_tgemm_nrt(d.N), 0, perm, d.N <= 256 ? (unsigned short*)(base + L->bsh[g]) : nullptr);
  }
  void copy(const float* src, size_t dst_off, int n) {
    hipLaunchKernelGGL(copy_kernel, dim3((unsigned)nl_cdiv(n, 256)), dim3(256), 0, st, src, (float*)(base + dst_off), n);
  }
  void transpose(const float* src, size_t dst_off, int Cc, int Lp) {
    hipLaunchKernelGGL(transpose_kernel, dim3((unsigned)nl_cdiv(Cc * Lp, 256)), dim3(256), 0, st, src, (float*)(base + dst_off), Cc, Lp);
  }
  void lane_major(size_t src_off, size_t dst_off, int N, int So) {
    const int nrt = nl_tgemm_nrt(N), total = (So > 32 ? So : 32) * nrt * 32;
    hipLaunchKernelGGL(ln_lane_major_kernel, dim3((unsigned)nl_cdiv(total, 256)), dim3(256), 0, st, (const float*)(base + src_off), (float*)(base + dst_off),
                       N, So, nrt, total);
  }
  void linear(int g, const float* w, const float* b) {  // torch (out, in)
    block(g, 0, w, 0, L->g[g].K, 1, L->g[g].K);
    if (b) copy(b, L->bias[g], L->g[g].N);
  }
  // conv taps over concatenated sources: weight (co, ci, 3); K index = tap-major then source channels
  // K order = per source (channel range [c0, c0 + wd) of the ci input channels), per 32-channel block, per tap: see NlGemmSeg::ntap
  // perm_mask bit s: source s arrives as a fragment image (NlGemmSeg::frag): its 32-channel blocks in accumulator order (pack_block_kernel: perm)
  void conv3(int g, const float* w, const float* b, int ci, const int* widths, int nsrc, unsigned perm_mask = 0) {
    int k0 = 0, c0 = 0;
    for (int sidx = 0; sidx < nsrc; ++sidx) {
      for (int cb = 0; cb < widths[sidx] / 32; ++cb)
        for (int j = 0; j < 3; ++j) { block(g, k0, w, (c0 + 32 * cb) * 3 + j, ci * 3, 3, 32, (perm_mask >> sidx) & 1); k0 += 32; }
      c0 += widths[sidx];
    }
    copy(b, L->bias[g], L->g[g].N);
  }
  // transposed conv weight (ci, co, 3): even phase uses tap 1; odd phase taps 2 (ioff 0) then 0 (ioff +1)
  // merged phases (see G_T3M): columns [0, co) = even phase (tap 1 on x[m]), [co, 2 co) = odd phase (tap 2 on x[m], tap 0 on x[m+1])
  void convT_merged(int g, const float* w, const float* b, int ci, int co) {
    const GemmDim& d = L->g[g];
    mark(g, true);
    auto win = [&](int k0, int tap, int n0) {
      const int n = ci * co;
      hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(n, 256)), dim3(256), 0, st, w, tap, 3, co * 3, ci, co, k0,
                         (float*)(base + L->b32[g]) + n0, (unsigned short*)(base + L->bhi[g]) + (size_t)n0 * d.Kpad,
                         (unsigned short*)(base + L->blo[g]) + (size_t)n0 * d.Kpad, d.Kpad, d.Npad, (unsigned short*)(base + L->bst[g]),
                         nl_tgemm_nrt(d.N), n0, 0, (unsigned short*)(base + L->bsh[g]));
    };
    win(0, 1, 0);
    win(0, 2, co);
    win(ci, 0, co);
    copy(b, L->bias[g], co);
    copy(b, L->bias[g] + 4 * (size_t)co, co);
  }
  // input-gradient weights of Conv1d(k = 3, padding 1), weight (co, ci, 3), for the input channels [n0, n0 + nn): K order [32-co block][tap slot
  // tau][32] like conv3 (NlGemmSeg::ntap), slot tau reads the output-gradient row t + tau - 1 and therefore carries tap 2 - tau
  void conv3_dgrad(int g, const float* w, int co, int ci, int n0, int nn) {
    const GemmDim& d = L->g[g];
    mark(g, false);
    int k0 = 0;
    for (int cb = 0; cb < co / 32; ++cb)
      for (int tau = 0; tau < 3; ++tau) {
        hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(32 * nn, 256)), dim3(256), 0, st, w, 32 * cb * ci * 3 + n0 * 3 + (2 - tau), 3, ci * 3, 32, nn, k0,
                           (float*)(base + L->b32[g]), (unsigned short*)(base + L->bhi[g]), (unsigned short*)(base + L->blo[g]), d.Kpad, d.Npad,
                           (unsigned short*)(base + L->bst[g]), nl_tgemm_nrt(d.N), 0);
        k0 += 32;
      }
  }
  // input-gradient weights of ConvTranspose1d(k = 3, stride 2), weight (ci, co, 3), against the merged-phase gradient rows [even | odd]:
  // K = [even: tap 1 | odd: tap 2 | odd of the previous position: tap 0]
  void convT_dgrad(int g, const float* w, int ci, int co) {
    const GemmDim& d = L->g[g];
    mark(g, false);
    const int taps[3] = {1, 2, 0};
    for (int part = 0; part < 3; ++part)
      hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(co * ci, 256)), dim3(256), 0, st, w, taps[part], co * 3, 3, co, ci, part * co,
                         (float*)(base + L->b32[g]), (unsigned short*)(base + L->bhi[g]), (unsigned short*)(base + L->blo[g]), d.Kpad, d.Npad,
                         (unsigned short*)(base + L->bst[g]), nl_tgemm_nrt(d.N), 0);
  }
  void convT(int ge, int go, const float* w, const float* b, int ci, int co) {
    block(ge, 0, w, 1, 3, co * 3, ci);
    block(go, 0, w, 2, 3, co * 3, ci);
    block(go, ci, w, 0, 3, co * 3, ci);
    copy(b, L->bias[ge], co);
    copy(b, L->bias[go], co);
  }
};

// ------------------------------------------------------------------------------------------ frame
}  // namespace

struct nl_frame {
  NlViews views;
  int C;
  const float* images; const float* feat; float* visf_hwc;
  const float* sp_xyz; const float* sp_feat; const float* sp_conf; const float* sp_dir;
  int64_t M;
  NlKnnGrid grid;
  float* ptt;               // [(M+1)][W] table T (see G_PTT), built lazily per (frame, weights); row M = bias
  const void* ptt_for; uint64_t ptt_gen;
  float* pfeat;             // (V,h,w,32) feature maps projected through the blend layer's feature columns
  const void* pfeat_for; uint64_t pfeat_gen;   // packed weights (address + pack generation) pfeat was computed with (lazily, first render of the frame)
  float* views_dev;         // device copy of the per-view matrices: [16][12] proj_ibr rows, then [16][3] camera centres
  float *tr_gT, *tr_tmp;    // training scratch sized by the table: d loss / d T (M, W) and (M, ldf_of(C)) staging rows (pt_backward_only)
  float views_host[16 * 15];
  // side stream of the fused render path (exact KNN beside the multi-view gather): owned by the frame, created in nl_frame_create —
  // never lazily inside a render call (stream / event creation is illegal during graph capture) and never shared between frames, so two
  // renderers on two caller streams do not record into each other's events.  side_ok == false: everything runs on the caller's stream.
  hipStream_t side; hipEvent_t ev_fork, ev_join; bool side_ok;
  // round 6: the search in parts (do_point: part p + 1 of the exact KNN is released onto the side stream when the neural-point kernel of part p starts) —
  // ev_go[p]: recorded on the caller's stream in front of part p - 1's neural-point launch; ev_done[p]: part p's neighbours and aggregation scales are written
  static constexpr int kMaxParts = 4;
  hipEvent_t ev_go[kMaxParts], ev_done[kMaxParts]; bool parts_ok;
  // precision guard (NL_RENDER_PRECISION_GUARD): the mode guarded calls render this frame in once one of them found the conditioning indicator beyond the
  // configured mode's validated range (-1: none yet), how often that happened, and the mode the last guarded call's outputs were produced in.  Host-side
  // state of a frame that is documented as not re-entrant; mutable because render calls take the frame as const.
  mutable int guard_prec = -1; mutable int guard_escalations = 0; mutable int guard_last_prec = -1;
};

namespace {

// Debug facility (nl_debug_bump_gap / nl_debug_check_gaps, used by the test-suite's guarded workspaces): with a gap size set, every buffer carved from a
// workspace is followed by that many untouched bytes, and the carve records [exact end of the buffer, start of the next one) — the caller fills the workspace
// with a pattern before the call and the check finds any byte a kernel wrote outside its buffer, also BETWEEN two buffers of one workspace.
size_t g_bump_gap = 0;
std::vector<std::pair<char*, size_t>> g_bump_gaps;
struct Bump {
  char* base; size_t off;
  template <class T> T* take(size_t count) {
    size_t o = off;
    off += nl_align_up(count * sizeof(T), 256) + g_bump_gap;
    if (g_bump_gap && base && g_bump_gaps.size() < (1u << 16)) g_bump_gaps.push_back({base + o + count * sizeof(T), off - o - count * sizeof(T)});
    return base ? (T*)(base + o) : nullptr;
  }
};

// ---- per-stage buffers ---------------------------------------------------------------------------
struct MvBufs { float *vis, *dd, *g393, *t64; };
struct PtBufs { int* idx; float *d2, *X, *H1, *H2, *KV, *Q, *O, *FCo, *wscale; };
struct UnBufs { float *r1, *c1, *r2, *c2, *r3, *c3, *x0r, *x0, *x1r, *x1, *x2r, *x2, *outr; };
struct HdBufs { float *sigma, *fth, *hc, *wsum, *blA, *rgb_s; int *n_alive, *tile_list, *tile_count; };

// leading dimensions of the two feature-width-dependent staging rows (416 and 288 at C = 192).  LDG: mv_stats zero-fills columns
// 2F+3 .. LDG-1, so out_fc's K is a whole number of 32-wide chunks; LDX: [posenc 63 | ray_diff 27 | F] padded likewise
inline int ldg_of(int C) { return (int)nl_align_up(2 * (C + 3) + 3, 32); }
inline int ldx_of(int C) { return (int)nl_align_up(C + 3 + 90, 32); }

void carve_mv(Bump& b, const nl_config* c, int V, int64_t N, MvBufs& m) {
  m.vis = b.take<float>((size_t)V * N); m.dd = b.take<float>((size_t)V * N);
  m.g393 = b.take<float>((size_t)N * ldg_of(c->C)); m.t64 = b.take<float>((size_t)N * 64);
}
void carve_pt(Bump& b, const nl_config* c, int64_t N, int K, PtBufs& p, bool force_generic = false) {
  const int W = c->W;
  p.idx = b.take<int>((size_t)N * K); p.d2 = b.take<float>((size_t)N * K);
  if (!force_generic && nl_point_fused_supported(W, c->precision)) { p.X = p.H1 = p.H2 = p.KV = nullptr; }
  else {
    p.X = b.take<float>((size_t)N * K * ldx_of(c->C));
    p.H1 = b.take<float>((size_t)N * K * W); p.H2 = b.take<float>((size_t)N * K * W);
    p.KV = b.take<float>((size_t)N * K * 256);
  }
  p.Q = b.take<float>((size_t)N * 128); p.O = b.take<float>((size_t)N * 128);
  p.FCo = b.take<float>((size_t)N * W); p.wscale = b.take<float>((size_t)N);
}
void carve_un(Bump& b, const nl_config* c, int64_t R, UnBufs& u) {
  const size_t N = (size_t)R * c->S;
  u.r1 = b.take<float>(N * 64); u.c1 = b.take<float>(N / 2 * 64);
  u.r2 = b.take<float>(N / 2 * 128); u.c2 = b.take<float>(N / 4 * 128);
  u.r3 = b.take<float>(N / 4 * 128); u.c3 = b.take<float>(N / 8 * 128);
  u.x0r = b.take<float>(N / 4 * 128); u.x0 = b.take<float>(N / 4 * 128);
  u.x1r = b.take<float>(N / 2 * 64); u.x1 = b.take<float>(N / 2 * 64);
  u.x2r = b.take<float>(N * 32); u.x2 = b.take<float>(N * 32);
  u.outr = b.take<float>(N * c->W);
}
void carve_hd(Bump& b, const nl_config* c, int V, int64_t R, HdBufs& h) {
  const size_t N = (size_t)R * c->S;
  h.sigma = b.take<float>(N); h.fth = b.take<float>(N * c->W); h.hc = b.take<float>((size_t)R * c->W); h.wsum = b.take<float>((size_t)R);
  h.blA = b.take<float>(N * 32); h.rgb_s = b.take<float>(N * 3);
  h.n_alive = b.take<int>((size_t)R); h.tile_list = b.take<int>(N / 32 + 2); h.tile_count = b.take<int>(1);
}

// whether the fused render path runs statistics + out_fc.0 in mv_front_kernel and recomputes the blend taps (round 4): a function of the configuration alone,
// so that workspace sizing and the render call agree
inline bool front_path(const nl_config* c, int V) {
  return !dbg_switch("NERFLOC_NO_FRONT") && c->precision != NL_PREC_F32 && nl_mv_front_supported(c->C, V, 1);
}

struct RenderBufs {
  float *xyz, *z, *G, *bl1, *rgbv, *FA, *geo; int* valid_s;
  MvBufs mv; PtBufs pt; UnBufs un; HdBufs hd;
};
void carve_render(Bump& b, const nl_config* c, int V, int64_t R, RenderBufs& rb) {
  const size_t N = (size_t)R * c->S;
  rb.xyz = b.take<float>(N * 3); rb.z = b.take<float>(N);
  rb.G = b.take<float>(N * c->W);
  // the blend layer's per-(sample, view) rows exist only where mv_front_kernel + blend_taps_kernel do not apply (other feature widths, fp32 mode, debug switch):
  // at config 2 that is 0.67 GB of the chunk's workspace
  rb.bl1 = front_path(c, V) ? nullptr : b.take<float>(N * V * 32);
  rb.rgbv = b.take<float>(N * V * 4);
  // (the statistics row of the old path is not needed then either, but carve_mv serves the stage API too)
  rb.valid_s = b.take<int>(N);
  rb.FA = b.take<float>(N * c->W); rb.geo = b.take<float>(N * c->W);
  carve_mv(b, c, V, N, rb.mv); carve_pt(b, c, N, 8, rb.pt); carve_un(b, c, R, rb.un); carve_hd(b, c, V, R, rb.hd);
}

// ---- GEMM helper ----------------------------------------------------------------------------------
struct Ctx {
  const nl_config* c; Layout L; const char* pk; hipStream_t st;
  uint64_t has_bst = ~0ull, has_bsh = ~0ull;   // layers whose streaming-kernel images exist in pk (pack_info)
  bool mx = false;                             // NL_PREC_F16MX: the fused neural-point kernel multiplies as fp16 hi.hi + two MX-FP8 cross terms (everything else: BF16X3)
  template <class T> const T* p(size_t off) const { return (const T*)(pk + off); }
};

struct SegSpec { const float* ptr; int ld; int k; int ioff; int rdiv; int ntap = 1; int frag = 0; };   // frag: NlGemmSeg::frag

struct TileMap { const int* map; const int* count; };
struct RowEpi { const float* res; int ldres; const float* gamma; const float* beta; const float* scale; float eps; float* out; int kind = NL_EPI_LNROW; int pool = 0;
                const float* sig_w = nullptr; const float* sig_b = nullptr; float* sig_out = nullptr;
                unsigned* maskout = nullptr; const unsigned* maskin = nullptr;
                const float* tab = nullptr; const int* tabidx = nullptr; int ldtab = 0, tabK = 0, tabM = 0; };   // out: destination when fused; mask*: sign bits (common.h: ep_maskout / ep_maskin)

// fills the launch descriptor; *fused says whether the optional row epilogue will run inside the GEMM (else the caller runs it)
int run_gemm(const Ctx& x, int g, const SegSpec* segs, int nseg, int64_t M, float* C, int ldc, int act,
             int So = 0, int Li = 0, int Lo = 0, int ostride = 1, int ooff = 0, const RowEpi* epi = nullptr, bool* fused = nullptr,
             const TileMap* tiles = nullptr) {
  NlGemmArgs a;
  memset(&a, 0, sizeof(a));
  if (tiles) { a.tile_map = tiles->map; a.tile_count = tiles->count; }
  int ksum = 0;
  for (int i = 0; i < NL_GEMM_MAX_SEG; ++i) a.kstart[i] = 0x7fffffff;
  for (int i = 0; i < nseg; ++i) {
    a.kstart[i] = ksum;
    a.seg[i].ptr = segs[i].ptr; a.seg[i].ld = segs[i].ld; a.seg[i].k = segs[i].k; a.seg[i].ioff = segs[i].ioff;
    a.seg[i].rdiv = segs[i].rdiv > 0 ? segs[i].rdiv : 1;
    a.seg[i].vec = ((((size_t)segs[i].ptr) & 15) == 0 && (segs[i].ld & 3) == 0) ? 1 : 0;
    a.seg[i].ntap = segs[i].ntap > 1 ? segs[i].ntap : 1;
    a.seg[i].frag = segs[i].frag;
    if (a.seg[i].ntap > 1 && (segs[i].k & 31)) return NL_ERR_BAD_ARG;
    ksum += ((segs[i].k + 31) & ~31) * a.seg[i].ntap;   // every segment occupies round_up(k, 32) slots of K-space (one k-tile = one segment)
  }
  const GemmDim& d = x.L.g[g];
  if (ksum != ((d.K + 31) & ~31)) return NL_ERR_BAD_ARG;
  a.nseg = nseg; a.M = (int)M; a.K = d.K; a.N = d.N; a.Kpad = d.Kpad; a.Npad = d.Npad;
  int prec = x.c->precision;
  if (prec == NL_PREC_F16X3_INTERNAL) {   // split-FP16 where the streaming kernel applies (its only implementation), exact fp32 elsewhere
    a.Bst = x.pk + x.L.bsh[g];
    a.zeros = x.p<float>(x.L.zeros); a.C = C; a.ldc = ldc; a.N = d.N; a.M = (int)M;
    a.epi = NL_EPI_NONE;
    if (tiles || d.N > 256 || !((x.has_bsh >> g) & 1) || !nl_tgemm_supported(a, prec)) { prec = NL_PREC_F32; a.Bst = nullptr; }   // (a requested fused epilogue is simply not fused)
  }
  if (prec == NL_PREC_F32) a.B = x.pk + x.L.b32[g];
  else if (prec != NL_PREC_F16X3_INTERNAL) { a.B = x.pk + x.L.bhi[g]; a.Blo = x.pk + x.L.blo[g]; a.Bst = ((x.has_bst >> g) & 1) ? x.pk + x.L.bst[g] : nullptr; }
  // NL_PREC_F16MX: conv_out multiplies as fp16 hi.hi + two MX-FP6 cross terms too (tgemm_mx_kernel) when its images exist
  if (x.mx && g == G_CONVOUTF && prec == NL_PREC_BF16X3 && x.c->W == 256 && ((x.has_bsh >> g) & 1) && !dbg_switch("NERFLOC_NO_TGEMM_MX")) {
    a.Bsh_mx = x.pk + x.L.bsh[g]; a.Bmx = x.pk + x.L.mx_convout;
  }
  for (int i = 0; i < nseg; ++i) if (segs[i].frag == 3 && ((x.has_bsh >> g) & 1)) a.Bsh16 = x.pk + x.L.bsh[g];   // split-FP16 fragments: the layer's fp16 stream (conv1)
  a.zeros = x.p<float>(x.L.zeros);
  a.bias = d.bias ? x.p<float>(x.L.bias[g]) : nullptr;
  a.C = C; a.ldc = ldc; a.act = act;
  a.So = So; a.Li = Li; a.Lo = Lo; a.ostride = ostride; a.ooff = ooff;
  if (fused) *fused = false;
  if (epi) {
    a.epi = epi->kind; a.ep_pool = epi->pool; a.ep_res = epi->res; a.ep_ldres = epi->ldres; a.ep_gamma = epi->gamma; a.ep_beta = epi->beta;
    a.ep_scale = epi->scale; a.ep_eps = epi->eps;
    a.ep_sig_w = epi->sig_w; a.ep_sig_b = epi->sig_b; a.ep_sig_out = epi->sig_out;
    a.ep_maskout = epi->maskout; a.ep_maskin = epi->maskin;
    a.ep_tab = epi->tab; a.ep_tabidx = epi->tabidx; a.ep_ldtab = epi->ldtab; a.ep_tabK = epi->tabK; a.ep_tabM = epi->tabM;
    if (nl_tgemm_supported(a, prec)) { a.C = epi->out; if (fused) *fused = true; }
    else if (epi->tab) return NL_ERR_UNSUPPORTED;   // (no other kernel knows the table: the caller checks *fused first or keeps the full-width layer)
    else { a.epi = NL_EPI_NONE; a.ep_maskout = nullptr; a.ep_maskin = nullptr; }
  }
  if (tiles && !nl_tgemm_supported(a, prec)) { a.tile_map = nullptr; a.tile_count = nullptr; }   // generic kernels compute every row
  return nl_gemm_launch(a, prec, x.st);
}

#define NL_TRY(e) do { int _rc = (e); if (_rc != NL_OK) return _rc; } while (0)
// First statement of every entry point that takes an nl_config: NL_PREC_F16MX is BF16X3 everywhere but in the fused neural-point kernel of the render path,
// so the library works on a BF16X3 copy of the configuration and remembers the request in nl_mx_ (nl_render_rays_ex passes it on as Ctx::mx)
#define NL_EFF_CFG(cfg)                                                                 \
  nl_config nl_eff_cfg_;                                                                 \
  bool nl_mx_ = false;                                                                   \
  if ((cfg) && (cfg)->precision == NL_PREC_F16MX) { nl_eff_cfg_ = *(cfg); nl_eff_cfg_.precision = NL_PREC_BF16X3; (cfg) = &nl_eff_cfg_; nl_mx_ = true; } \
  (void)nl_mx_

// qrows != null: per-ray query centres (device, row = sample / S) instead of the one host-side centre qc
NlViews with_query(const nl_frame* f, const float* qc, const float* qrows = nullptr, int S = 1) {
  NlViews v = f->views;
  v.qcam[0] = qc ? qc[0] : 0.f; v.qcam[1] = qc ? qc[1] : 0.f; v.qcam[2] = qc ? qc[2] : 0.f;
  v.qrows = qrows; v.qS = S > 0 ? S : 1;
  return v;
}

// Every nl_pack_weights call stamps its destination with a fresh generation number (host-side registry keyed by the blob's
// address): the per-frame tables derived from the weights are rebuilt when a blob is RE-packed in place, not only when another
// blob is used.
// ... and the registry remembers WHICH layers of the blob have a streaming-kernel image (bit g: bf16 hi / lo stream, fp16 hi / lo stream): run_gemm keeps a
// product off the streaming kernel when its stream was never written (it would multiply by zeros: the transposed out_fc.0 did, for feature widths whose
// statistics row fits 256 columns, until tools/grad_fuzz.py) — the generic kernels read the plain images every layer has.
struct PackInfo { uint64_t gen, bst, bsh; };
std::mutex g_gen_mu;
std::unordered_map<const void*, PackInfo> g_pack_gen;
uint64_t g_gen_next = 1;
uint64_t pack_generation(const void* pk) {
  std::lock_guard<std::mutex> lk(g_gen_mu);
  auto it = g_pack_gen.find(pk);
  return it == g_pack_gen.end() ? 0 : it->second.gen;
}
PackInfo pack_info(const void* pk) {
  std::lock_guard<std::mutex> lk(g_gen_mu);
  auto it = g_pack_gen.find(pk);
  return it == g_pack_gen.end() ? PackInfo{0, ~0ull, ~0ull} : it->second;   // (a blob this process did not pack, e.g. copied: trusted as complete)
}
void bump_generation(const void* pk) {
  std::lock_guard<std::mutex> lk(g_gen_mu);
  g_pack_gen[pk] = PackInfo{g_gen_next++, 0, 0};
}
void set_pack_streams(const void* pk, uint64_t bst, uint64_t bsh) {
  std::lock_guard<std::mutex> lk(g_gen_mu);
  auto it = g_pack_gen.find(pk);
  if (it != g_pack_gen.end()) { it->second.bst = bst; it->second.bsh = bsh; }
}

// per-frame projection of the support feature maps through the blend layer (exact fp32 MFMA), done once per (frame, weights)
int ensure_pfeat(const Ctx& x, const nl_frame* fc) {
  nl_frame* f = const_cast<nl_frame*>(fc);
  const uint64_t gen = pack_generation(x.pk);
  if (f->pfeat_for == (const void*)x.pk && f->pfeat_gen == gen) return NL_OK;
  nl_config c32 = *x.c;
  c32.precision = NL_PREC_F32;
  Ctx x32 = x;
  x32.c = &c32;
  SegSpec s{f->feat, f->C, f->C, 0, 1};
  NL_TRY(run_gemm(x32, G_BLENDP, &s, 1, (int64_t)f->views.V * f->views.h * f->views.w, f->pfeat, 32, NL_ACT_NONE));
  f->pfeat_for = (const void*)x.pk; f->pfeat_gen = gen;
  return NL_OK;
}

// per-frame table T for the fused point kernel (exact fp32 MFMA), once per (frame, weights)
int ensure_ptt(const Ctx& x, const nl_frame* fc) {
  nl_frame* f = const_cast<nl_frame*>(fc);
  const uint64_t gen = pack_generation(x.pk);
  if (f->ptt_for == (const void*)x.pk && f->ptt_gen == gen) return NL_OK;
  const int W = x.c->W, F = f->C + 3;
  nl_config c32 = *x.c;
  c32.precision = NL_PREC_F32;
  Ctx x32 = x;
  x32.c = &c32;
  if (f->M > 0) {
    SegSpec s{f->sp_feat, F, F, 0, 1};
    NL_TRY(run_gemm(x32, G_PTT, &s, 1, f->M, f->ptt, W, NL_ACT_NONE));
  }
  NL_CHECK_HIP(hipMemcpyAsync(f->ptt + (size_t)f->M * W, x.pk + x.L.bias[G_PTT], sizeof(float) * W, hipMemcpyDeviceToDevice, x.st));
  // max |T| (bias row included): what the f16mx kernel bounds base_mlp.0's outputs with; kept in the slack behind the per-view matrices (float 248 of that 1-KB block)
  NL_TRY(nl_table_absmax(f->ptt, (size_t)(f->M + 1) * W, f->views_dev + 248, x.st));
  f->ptt_for = (const void*)x.pk; f->ptt_gen = gen;
  return NL_OK;
}

// ---- fork / join of the fused render path's side stream ------------------------------------------------------------------
// The exact KNN (+ the aggregation scale) only needs the sample positions, like the multi-view gather kernels: nl_render_rays forks it
// onto the frame's side stream and joins before the neural-point kernel (events: graph-capturable).  SideJoin makes the join
// unconditional: whatever path leaves the scope after the fork — including an error return — the caller's stream waits for the
// side stream first, so no kernel is left writing the caller's workspace behind its back and an active capture stays well-formed.
struct SideJoin {
  hipStream_t main = nullptr, side = nullptr; hipEvent_t ev = nullptr; bool armed = false;
  int parts = 1;            // > 1: only part 0 of the search has been issued (rows [0, part_rows)); do_point issues the others, each when the part before it starts
  int64_t part_rays = 0;    // rays per part (parts are cut at ray boundaries: the per-sample direction is the ray's)
  void arm(hipStream_t m, hipStream_t s_, hipEvent_t e) { main = m; side = s_; ev = e; armed = true; }
  int join() {
    if (!armed) return NL_OK;
    armed = false;
    if (hipEventRecord(ev, side) != hipSuccess || hipStreamWaitEvent(main, ev, 0) != hipSuccess) return NL_ERR_HIP;
    return NL_OK;
  }
  ~SideJoin() { (void)join(); }
};

// ---- measurement hook: HIP events around the dominant kernel (nl_profile_begin / nl_profile_end) -----------------
struct ProfState { bool on = false; std::vector<hipEvent_t> ev; int used = 0; };
ProfState g_prof;
bool prof_arm(hipEvent_t* e0, hipEvent_t* e1) {
  if (!g_prof.on) return false;
  if (g_prof.used + 2 > (int)g_prof.ev.size()) {
    hipEvent_t a, b;
    if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) return false;
    g_prof.ev.push_back(a); g_prof.ev.push_back(b);
  }
  *e0 = g_prof.ev[g_prof.used]; *e1 = g_prof.ev[g_prof.used + 1];
  g_prof.used += 2;
  return true;
}

// front (fused render path, C = 192, non-fp32 modes; round 4): statistics + out_fc.0 in mv_front_kernel — no statistics row, no per-(sample, view) blend rows
// (bl1 is then not written: the blend tail recomputes its taps, do_heads_pre); rgbv still carries the tapped colours + visibility
int do_mv(const Ctx& x, const nl_frame* f, const float* qc, const float* xyz, int64_t N, float* G, float* rgb_feat,
          float* vis_ang, int* valid_s, float* bl1, float* rgbv, const MvBufs& m, bool skip_g = false, const float* qrows = nullptr, int qS = 1,
          bool front = false) {
  const NlViews vw = with_query(f, qc, qrows, qS);
  if (bl1 || front) NL_TRY(ensure_pfeat(x, f));
  if (x.c->precision == NL_PREC_F32) NL_TRY(nl_launch_mv_vis(vw, f->visf_hwc, x.p<float>(x.L.dec_w), xyz, N, m.vis, m.dd, x.st));
  else NL_TRY(nl_launch_mv_vis_mfma(vw, f->visf_hwc, x.p<char>(x.L.dec_mfma), xyz, N, m.vis, m.dd, x.c->precision == NL_PREC_BF16X3, x.st));
  if (front) {
    NL_TRY(nl_launch_mv_front(vw, f->views_dev, f->images, f->feat, xyz, N, m.vis, m.dd, x.p<char>(x.L.mvf_pack), m.t64, valid_s, rgbv, x.st));
    if (skip_g) return NL_OK;
    SegSpec s1{m.t64, 64, 64, 0, 1};
    NL_TRY(run_gemm(x, G_OUTFC2, &s1, 1, N, G, x.c->W, NL_ACT_ELU));
    return NL_OK;
  }
  NL_TRY(nl_launch_mv_stats(vw, f->views_dev, f->images, f->feat, f->C, xyz, N, m.vis, m.dd, m.g393, ldg_of(f->C), rgb_feat, vis_ang, valid_s, f->pfeat,
                            x.p<float>(x.L.blw), bl1, rgbv, x.st));
  SegSpec s0{m.g393, ldg_of(f->C), ldg_of(f->C), 0, 1};
  NL_TRY(run_gemm(x, G_OUTFC0, &s0, 1, N, m.t64, 64, NL_ACT_ELU));
  if (skip_g) return NL_OK;   // the consumers recompute G from the hidden rows (sample_chain_kernel)
  SegSpec s1{m.t64, 64, 64, 0, 1};
  NL_TRY(run_gemm(x, G_OUTFC2, &s1, 1, N, G, x.c->W, NL_ACT_ELU));
  return NL_OK;
}

// knn_done != null: the caller already ran the KNN (+ the aggregation scale) on a side stream and hands over the event to wait for
// chain != null && chain->t64 != null (fused render path, W = 256, bf16 modes): the query rows before the branch and fc + LayerNorm +
// scale, feat_mlp.0 (chain->fth, may be null) and the blend projection (chain->blA) after it run as two chain kernels that recompute
// the multiview feature rows G from out_fc's hidden rows t64 (G is then not read here and need not exist); *chain->done reports it
struct ChainOut { float* fth; float* blA; bool* done; const float* t64 = nullptr; bool fa_frag = false; bool fa_f16 = false; };   // fa_frag: FA leaves the chain kernel as a fragment image (fa_f16: in split-FP16)
int do_point(const Ctx& x, const nl_frame* f, const float* xyz, const float* dir, int dir_stride, int dir_div, const float* G, int64_t N, int K,
             float* FA, const PtBufs& p, SideJoin* knn_done = nullptr, const ChainOut* chain = nullptr) {
  const int W = x.c->W, F = f->C + 3;
  const bool fused_path = K == 8 && nl_point_fused_supported(W, x.c->precision);
  if (!knn_done) NL_TRY(nl_knn_search(&f->grid, xyz, N, K, p.idx, p.d2, x.st));
  const float* t64 = chain ? chain->t64 : nullptr;
  if (t64) {
    NL_TRY(nl_launch_query_chain(t64, x.pk, x.L.bst[G_OUTFC2], x.p<float>(x.L.bias[G_OUTFC2]), x.L.bst[G_QP], p.Q, N, x.c->precision, x.st));
  } else {
    SegSpec sg{G, W, W, 0, 1};
    NL_TRY(run_gemm(x, G_Q, &sg, 1, N, p.Q, 128, NL_ACT_NONE));
  }
  const int parts = knn_done ? knn_done->parts : 1;
  if (knn_done && parts == 1) NL_TRY(knn_done->join());
  if (parts > 1 && (!fused_path || dir_div <= 0)) return NL_ERR_UNSUPPORTED;   // (render_rays_impl plans parts for the fused kernels only)
  if (fused_path) {
    NL_TRY(ensure_ptt(x, f));
    if (!knn_done) NL_TRY(nl_launch_wscale(p.idx, p.d2, f->sp_conf, N, K, f->M, p.wscale, x.st));
    NlPointFusedArgs a;
    a.xyz = xyz; a.dir = dir; a.dir_stride = dir_stride; a.dir_div = dir_div > 0 ? dir_div : 1;
    a.idx = p.idx; a.Q = p.Q; a.O = p.O; a.ptt = f->ptt; a.sp_xyz = f->sp_xyz; a.sp_dir = f->sp_dir;
    a.wstream = x.p<uint4>(x.L.pt_stream); a.bias = x.p<float>(x.L.pt_bias); a.rd_w = x.p<float>(x.L.rd_w);
    const bool mx = x.mx && x.c->precision == NL_PREC_BF16X3 && nl_point_fused2_supported(W, x.c->precision);
    a.wstream2 = x.p<uint4>(mx ? x.L.pt_stream2_mx : x.L.pt_stream2);
    a.N = (int)N; a.M = (int)(f->M > 0x7fffffff ? 0x7fffffff : f->M); a.inv_span = 1.f / (f->views.far_ - f->views.near_);
    hipEvent_t pe0 = nullptr, pe1 = nullptr;
    if (prof_arm(&pe0, &pe1)) NL_CHECK_HIP(hipEventRecord(pe0, x.st));
    const bool use_v1 = dbg_switch("NERFLOC_POINT_V1");
    a.logit_amax = reinterpret_cast<unsigned*>(f->views_dev + 249);   // (read by the v1 kernel; the v2 launcher takes it as a parameter)
    // The search in parts (round 6; VERDICT r5 item 1c: "KNN of chunk k + 1 under point_fused2 of chunk k — build it, keep the timeline whichever way it falls").
    // Only part 0 of the search is issued up front; part p + 1 is released onto the side stream when the neural-point launch of part p starts (rows of different
    // samples are independent: the kernel runs once per part).  BUILT, MEASURED, OFF (NL_KNN_PARTS = 1): beside the persistent matrix kernel the search finds room
    // for ONE of its workgroups per CU (88 of the 512 registers per lane are left) and runs at 130 queries / us instead of 700; the front end does end 0.30 / 0.40 ms
    // earlier with 2 / 4 parts, but the neural-point launches take 0.34 ms longer per 262 144 queries searched beside them (+17-20 %: the search's vector instructions
    // issue from the same SIMDs) and wait for their part — 7.70 (1 part) / 7.89 (2) / 8.05 ms (4) per config-2 step, same box (profiles/r6_knn_parts_*_timeline.txt).
    const int64_t rows_pp = parts > 1 ? knn_done->part_rays * a.dir_div : N;
    for (int pt = 0; pt < parts; ++pt) {
      const int64_t n0 = pt * rows_pp, n1 = n0 + rows_pp < N ? n0 + rows_pp : N;
      if (n0 >= N) break;
      if (parts > 1) {
        const int64_t m0 = n1, m1 = m0 + rows_pp < N ? m0 + rows_pp : N;
        if (pt + 1 < parts && m0 < N) {
          NL_CHECK_HIP(hipEventRecord(f->ev_go[pt + 1], x.st));
          NL_CHECK_HIP(hipStreamWaitEvent(f->side, f->ev_go[pt + 1], 0));
          NL_TRY(nl_knn_search(&f->grid, xyz + 3 * m0, m1 - m0, K, p.idx + (size_t)K * m0, p.d2 + (size_t)K * m0, f->side));
          NL_TRY(nl_launch_wscale(p.idx + (size_t)K * m0, p.d2 + (size_t)K * m0, f->sp_conf, m1 - m0, K, f->M, p.wscale + m0, f->side));
          NL_CHECK_HIP(hipEventRecord(f->ev_done[pt + 1], f->side));
        }
        NL_CHECK_HIP(hipStreamWaitEvent(x.st, f->ev_done[pt], 0));
      }
      NlPointFusedArgs b = a;
      b.xyz = xyz + 3 * n0; b.dir = dir + (size_t)dir_stride * (n0 / a.dir_div); b.idx = p.idx + (size_t)K * n0; b.Q = p.Q + (size_t)128 * n0; b.O = p.O + (size_t)128 * n0;
      b.N = (int)(n1 - n0);
      int rc2 = NL_ERR_UNSUPPORTED;
      if (!use_v1 && nl_point_fused2_supported(W, x.c->precision)) rc2 = nl_launch_point_fused2(b, W, x.c->precision, x.st, mx, nullptr, nullptr, f->views_dev + 248,
                                                                                                        reinterpret_cast<unsigned*>(f->views_dev + 249),
                                                                                                        reinterpret_cast<unsigned long long*>(f->views_dev + 250));
      if (rc2 == NL_ERR_UNSUPPORTED) rc2 = nl_launch_point_fused(b, W, x.c->precision, x.st);   // (e.g. more rows than 32-bit buffer offsets reach)
      NL_TRY(rc2);
    }
    if (knn_done && parts > 1) NL_TRY(knn_done->join());   // (the last part's event was waited for above: this only disarms the guard)
    if (pe1) NL_CHECK_HIP(hipEventRecord(pe1, x.st));
  } else {
    if (!p.X) return NL_ERR_UNSUPPORTED;
    NL_TRY(nl_launch_point_encode(xyz, dir, dir_stride, dir_div, N, K, f->M, p.idx, p.d2, f->sp_xyz, f->sp_feat, F, f->sp_conf, f->sp_dir,
                                  x.p<float>(x.L.rd_w), 1.f / (f->views.far_ - f->views.near_), p.X, ldx_of(f->C), p.wscale, x.st));
    const int64_t MK = N * K;
    SegSpec sx{p.X, ldx_of(f->C), F + 90, 0, 1};
    NL_TRY(run_gemm(x, G_BASE0, &sx, 1, MK, p.H1, W, NL_ACT_LRELU));
    SegSpec s1{p.H1, W, W, 0, 1}, s2{p.H2, W, W, 0, 1};
    NL_TRY(run_gemm(x, G_BASE2, &s1, 1, MK, p.H2, W, NL_ACT_LRELU));
    NL_TRY(run_gemm(x, G_BASE4, &s2, 1, MK, p.H1, W, NL_ACT_LRELU));
    NL_TRY(run_gemm(x, G_KV, &s1, 1, MK, p.KV, 256, NL_ACT_NONE));
    NL_TRY(nl_launch_attn(p.Q, p.KV, N, K, p.O, x.st, reinterpret_cast<unsigned*>(f->views_dev + 249)));   // (the staged path reports the conditioning indicator too)
  }
  if (chain && chain->done) *chain->done = false;
  if (t64) {   // (the caller checked W, precision and the 32-bit offset range before leaving G unmaterialised)
    NL_TRY(nl_launch_sample_chain(p.O, t64, p.wscale, x.p<float>(x.L.ln_g), x.p<float>(x.L.ln_b), 1e-6f, x.pk, x.L.bst[G_OUTFC2],
                                  x.p<float>(x.L.bias[G_OUTFC2]), x.L.bst[G_FC], x.L.bst[G_FEAT0P], chain->fa_f16 ? x.L.bsh[G_BLENDAP] : x.L.bst[G_BLENDAP],
                                  x.p<float>(x.L.bias[G_FEAT0P]), FA, chain->fth, chain->blA, N, x.c->precision, x.st, chain->fa_frag, chain->fa_f16));
    if (chain->done) *chain->done = true;
    return NL_OK;
  }
  SegSpec so{p.O, 128, 128, 0, 1};
  // fc + residual + LayerNorm + aggregation scale: inside the GEMM's epilogue when the streaming kernel takes it
  const RowEpi ep{G, W, x.p<float>(x.L.ln_g), x.p<float>(x.L.ln_b), p.wscale, 1e-6f, FA};
  bool fused = false;
  NL_TRY(run_gemm(x, G_FC, &so, 1, N, p.FCo, W, NL_ACT_NONE, 0, 0, 0, 1, 0, &ep, &fused));
  if (!fused) NL_TRY(nl_launch_ln_agg(p.FCo, G, N, W, x.p<float>(x.L.ln_g), x.p<float>(x.L.ln_b), 1e-6f, p.wscale, FA, x.st));
  return NL_OK;
}

// ---- input gradient of the neural-point branch (frozen weights) ---------------------------------------------------------------------
// Where a training step's backward calls ADD the gradients of the weights and of the per-frame tables (nl_train_grads, resolved)
struct TrainOut {
  float* w[kNumWeights];
  float* sp_feat;
  float *feat_maps, *pfeat_maps, *vis_maps;   // (V,h,w,C), (V,h,w,32), (V,vh,vw,32)
  float* scratch; size_t scratch_floats;
  bool any(int a, int b) const { for (int i = a; i < b; ++i) if (w[i]) return true; return false; }
};
// gW[tw] += dY^T X (and gb[tb] += column sums of dY) for whichever of the two the caller asked for
int wgrad_to(const TrainOut* tg, hipStream_t st, int tw, int tb, const float* dY, int ldy, int Mo, const float* X, int ldxx, int Ni, int64_t rows) {
  if (!tg || (!tg->w[tw] && (tb < 0 || !tg->w[tb]))) return NL_OK;
  if (!tg->w[tw]) return nl_launch_colsum(dY, ldy, rows, Mo, tg->w[tb], tg->scratch, st);
  return nl_launch_wgrad(dY, ldy, Mo, X, ldxx, Ni, rows, 0, 0, tg->w[tw], Ni, 1, 0, tb >= 0 ? tg->w[tb] : nullptr, tg->scratch, tg->scratch_floats, st);
}
inline int ldf_of(int C) { return (int)nl_align_up(C + 3, 32); }
struct PtBwdBufs { int* idx; float *d2, *X, *H1, *H2, *H3, *KV, *Q, *O, *FCo, *wscale, *gpre, *gO, *gQ, *gKV, *gA, *gB, *gX, *aff, *tr, *gXF; unsigned* mk[3]; };
void carve_ptb(Bump& b, const nl_config* c, int64_t N, int K, PtBwdBufs& p, bool train = false) {
  const int W = c->W;
  const size_t NK = (size_t)N * K;
  p.idx = b.take<int>(NK); p.d2 = b.take<float>(NK);
  p.X = b.take<float>(NK * ldx_of(c->C));
  p.H1 = b.take<float>(NK * W); p.H2 = b.take<float>(NK * W); p.H3 = b.take<float>(NK * W);
  p.KV = b.take<float>(NK * 256);
  p.Q = b.take<float>((size_t)N * 128); p.O = b.take<float>((size_t)N * 128); p.FCo = b.take<float>((size_t)N * W); p.wscale = b.take<float>((size_t)N);
  p.gpre = b.take<float>((size_t)N * W); p.gO = b.take<float>((size_t)N * 128); p.gQ = b.take<float>((size_t)N * 128);
  p.gKV = b.take<float>(NK * 256); p.gA = b.take<float>(NK * W); p.gB = b.take<float>(NK * W); p.gX = b.take<float>(NK * 96);
  for (int i = 0; i < 3; ++i) p.mk[i] = b.take<unsigned>((NK / 32 + 8) * 256);   // LeakyReLU sign bits of the three base_mlp layers: 32 bytes per row
  p.aff = p.tr = p.gXF = nullptr;
  if (train) {
    p.aff = b.take<float>((size_t)N * 2 * W); p.tr = b.take<float>(NK * 68);
    if (c->precision == NL_PREC_F32) p.gXF = b.take<float>(NK * ldf_of(c->C));   // (otherwise the support features' gradient goes through the table: pt_backward_only)
  }
}

// dX = (dY . W) * LeakyReLU'(h): the mask inside the streaming GEMM's epilogue where that kernel runs, a separate pass otherwise (fp32 mode)
// the layers' sign bits exist when the forward layers ran on the streaming kernel (every mode but fp32: pt_forward_staged checks it)
inline bool pt_mask_bits(const Ctx& x) { return x.c->precision != NL_PREC_F32; }
inline bool pt_table(const Ctx& x) { return x.c->precision != NL_PREC_F32; }   // base_mlp.0 through the per-frame table (pt_forward_staged)
int gemm_lrelu_masked(const Ctx& x, int g, const SegSpec& s, int64_t M, float* out, int ld, const float* h, const unsigned* bits = nullptr) {
  if (x.c->precision != NL_PREC_F32 && (s.k & 31) == 0 && (((size_t)s.ptr) & 15) == 0 && (s.ld & 3) == 0 && (ld & 3) == 0 && (((size_t)h) & 15) == 0 && x.L.g[g].N <= 256) {
    RowEpi ep{h, ld, nullptr, nullptr, nullptr, 0.f, out, NL_EPI_NONE};
    ep.maskin = bits;
    bool streamed = false;
    NL_TRY(run_gemm(x, g, &s, 1, M, out, ld, NL_ACT_LRELU_MASK, 0, 0, 0, 1, 0, &ep, &streamed));
    return streamed ? NL_OK : NL_ERR_UNSUPPORTED;   // (the generic kernels do not know this activation)
  }
  NL_TRY(run_gemm(x, g, &s, 1, M, out, ld, NL_ACT_NONE));
  return nl_launch_lrelu_mask(out, h, (size_t)M * ld, x.st);
}

// Re-runs the staged forward (point.hip kernels + segment GEMMs in the configured precision) into the workspace, then walks back:
// g_FA -> LayerNorm/scale -> {residual -> g_G ; fc^T -> attention -> {w_qs^T -> g_G ; [w_ks; w_vs]^T -> base_mlp^T x 3 with LeakyReLU masks ->
// posenc / ray_diff_fc -> g_xyz, g_dir}}.  The aggregation scale sum_k w_k is a constant of the backward pass: it is identically 1 (or 0)
// whatever the distances are (model.py:419-427 normalises the weights; the K rows they multiply are identical, see point.hip).
// the staged forward of the branch into the workspace (everything the way back reads); dir: one row per dir_div samples
int pt_forward_staged(const Ctx& x, const nl_frame* f, const float* xyz, const float* dir, int dir_stride, int dir_div, const float* G, int64_t N, int K,
                      const PtBwdBufs& p, const int* idx_in, const float* d2_in) {
  const int W = x.c->W, F = f->C + 3, ldx = ldx_of(f->C);
  const int64_t NK = N * K;
  const float inv_span = 1.f / (f->views.far_ - f->views.near_);
  const int64_t M = f->M;
  const int* idx = idx_in && d2_in ? idx_in : p.idx;
  const float* d2 = idx_in && d2_in ? d2_in : p.d2;
  if (idx == p.idx) NL_TRY(nl_knn_search(&f->grid, xyz, N, K, p.idx, p.d2, x.st));   // (the caller may hand over the forward call's neighbours)
  // every mode but fp32: base_mlp.0 on the per-frame table T like the fused kernel — the encoded rows are the 96 posenc + ray_diff_fc columns only (the 195
  // gathered feature columns per row are never written: 400 MB per 65 k samples), the layer's K is 96 instead of 288, and its epilogue adds T[neighbour]
  const bool tab = pt_table(x);
  if (tab) NL_TRY(ensure_ptt(x, f));
  NL_TRY(nl_launch_point_encode(xyz, dir, dir_stride, dir_div, N, K, M, idx, d2, f->sp_xyz, f->sp_feat, tab ? 0 : F, f->sp_conf, f->sp_dir, x.p<float>(x.L.rd_w),
                                inv_span, p.X, tab ? 96 : ldx, p.wscale, x.st));
  // (the encoded rows' pad columns are zero and so are the weights' pad rows: taking all ldx columns keeps the streaming kernel applicable)
  SegSpec sx{p.X, ldx, ldx, 0, 1}, s1{p.H1, W, W, 0, 1}, s2{p.H2, W, W, 0, 1}, s3{p.H3, W, W, 0, 1}, sg{G, W, W, 0, 1}, so{p.O, 128, 128, 0, 1};
  if (tab) sx = SegSpec{p.X, 96, 96, 0, 1};
  if (pt_mask_bits(x)) {   // the layers also leave their outputs' signs as bits: the way back reads 32 bytes per row instead of the 1 KB activation row
    const int gs[3] = {tab ? G_BASE0_S : G_BASE0, G_BASE2, G_BASE4};
    const SegSpec* ss[3] = {&sx, &s1, &s2};
    float* hs[3] = {p.H1, p.H2, p.H3};
    for (int i = 0; i < 3; ++i) {
      RowEpi ep{nullptr, 0, nullptr, nullptr, nullptr, 0.f, hs[i], NL_EPI_NONE};
      ep.maskout = p.mk[i];
      if (i == 0 && tab) { ep.tab = f->ptt; ep.tabidx = idx; ep.ldtab = W; ep.tabK = K; ep.tabM = (int)(M > 0x7fffffff ? 0x7fffffff : M); }
      bool streamed = false;
      NL_TRY(run_gemm(x, gs[i], ss[i], 1, NK, hs[i], W, NL_ACT_LRELU, 0, 0, 0, 1, 0, &ep, &streamed));
      if (!streamed) return NL_ERR_UNSUPPORTED;
    }
  } else {
    NL_TRY(run_gemm(x, G_BASE0, &sx, 1, NK, p.H1, W, NL_ACT_LRELU));
    NL_TRY(run_gemm(x, G_BASE2, &s1, 1, NK, p.H2, W, NL_ACT_LRELU));
    NL_TRY(run_gemm(x, G_BASE4, &s2, 1, NK, p.H3, W, NL_ACT_LRELU));
  }
  NL_TRY(run_gemm(x, G_KV, &s3, 1, NK, p.KV, 256, NL_ACT_NONE));
  NL_TRY(run_gemm(x, G_Q, &sg, 1, N, p.Q, 128, NL_ACT_NONE));
  NL_TRY(nl_launch_attn(p.Q, p.KV, N, K, p.O, x.st));
  return run_gemm(x, G_FC, &so, 1, N, p.FCo, W, NL_ACT_NONE);
}
// Frozen weights (pose refinement: no weight gradient wants the layers' activations), W = 128 / 256, K = 8, non-fp32 modes: the branch's forward as ONE launch of the
// fused neural-point kernel in split-FP16 (point_fused2_kernel<NRT, true, false, F16, KEEP>) that also leaves the k / v rows and the three layers' sign bits — what
// pt_backward_only reads — instead of an encode kernel, four (N x 8)-row GEMMs through HBM and an attention kernel (round 4: 1.7 -> 0.6 ms of a 512-ray step).
bool pt_keep_fused_ok(const Ctx& x, const nl_frame* f, int64_t N, int K) {
  return !dbg_switch("NERFLOC_NO_KEEP_FUSED") && K == 8 && x.c->precision == NL_PREC_F16X3_INTERNAL && nl_point_fused2_supported(x.c->W, NL_PREC_BF16X3) && f->M >= 1 &&
         N * 8 * 1024 <= 0x7fffffffll && ((int64_t)f->M + 1) * x.c->W * 4 <= 0x7fffffffll;
}
int pt_forward_keep_fused(const Ctx& x, const nl_frame* f, const float* xyz, const float* dir, int dir_stride, int dir_div, const float* G, int64_t N,
                          const PtBwdBufs& p, const int* idx_in, const float* d2_in) {
  const int W = x.c->W, K = 8;
  const int* idx = idx_in && d2_in ? idx_in : p.idx;
  const float* d2 = idx_in && d2_in ? d2_in : p.d2;
  if (idx == p.idx) NL_TRY(nl_knn_search(&f->grid, xyz, N, K, p.idx, p.d2, x.st));
  NL_TRY(ensure_ptt(x, f));
  NL_TRY(nl_launch_wscale(idx, d2, f->sp_conf, N, K, f->M, p.wscale, x.st));
  SegSpec sg{G, W, W, 0, 1}, so{p.O, 128, 128, 0, 1};
  NL_TRY(run_gemm(x, G_Q, &sg, 1, N, p.Q, 128, NL_ACT_NONE));
  NlPointFusedArgs a;
  memset(&a, 0, sizeof(a));
  a.xyz = xyz; a.dir = dir; a.dir_stride = dir_stride; a.dir_div = dir_div > 0 ? dir_div : 1;
  a.idx = idx; a.Q = p.Q; a.O = p.O; a.ptt = f->ptt; a.sp_xyz = f->sp_xyz; a.sp_dir = f->sp_dir;
  a.wstream = nullptr; a.bias = x.p<float>(x.L.pt_bias); a.rd_w = x.p<float>(x.L.rd_w);
  a.wstream2 = x.p<uint4>(x.L.pt_stream2_f16);
  a.N = (int)N; a.M = (int)(f->M > 0x7fffffff ? 0x7fffffff : f->M); a.inv_span = 1.f / (f->views.far_ - f->views.near_);
  unsigned* mk[3] = {p.mk[0], p.mk[1], p.mk[2]};
  NL_TRY(nl_launch_point_fused2(a, W, NL_PREC_BF16X3, x.st, false, p.KV, mk));
  return run_gemm(x, G_FC, &so, 1, N, p.FCo, W, NL_ACT_NONE);
}

int pt_backward_only(const Ctx& xb, const Ctx& x, const nl_frame* f, const float* xyz, const float* dir, int dir_stride, int dir_div, const float* G, int64_t N,
                     int K, const float* gFA, float* g_xyz, float* g_dir, float* g_G, const PtBwdBufs& p, const int* idx_in, const float* d2_in,
                     const TrainOut* tg) {
  const int W = x.c->W, F = f->C + 3, ldx = ldx_of(f->C);
  // training: gW += dY^T X right after each dY exists (its buffer is reused by the next layer's)
  auto wg = [&](int tw, int tb, const float* dY, int ldy, int Mo, const float* X, int ldxx, int Ni, int64_t rows) -> int {
    return wgrad_to(tg, x.st, tw, tb, dY, ldy, Mo, X, ldxx, Ni, rows);
  };
  const int64_t NK = N * K;
  const float inv_span = 1.f / (f->views.far_ - f->views.near_);
  const int64_t M = f->M;
  const int* idx = idx_in && d2_in ? idx_in : p.idx;
  const bool aff = tg && (tg->w[T_LNW] || tg->w[T_LNB]);
  NL_TRY(nl_launch_ln_agg_backward(p.FCo, G, gFA, N, W, x.p<float>(x.L.ln_g), 1e-6f, p.wscale, p.gpre, aff ? p.aff : nullptr, x.st));
  if (aff) {
    if (tg->w[T_LNW]) NL_TRY(nl_launch_colsum(p.aff, 2 * W, N, W, tg->w[T_LNW], tg->scratch, x.st));
    if (tg->w[T_LNB]) NL_TRY(nl_launch_colsum(p.aff + W, 2 * W, N, W, tg->w[T_LNB], tg->scratch, x.st));
  }
  NL_TRY(wg(T_FC, -1, p.gpre, W, W, p.O, 128, 128, N));
  SegSpec sp{p.gpre, W, W, 0, 1}, sgq{p.gQ, 128, 128, 0, 1}, skv{p.gKV, 256, 256, 0, 1}, sa{p.gA, W, W, 0, 1}, sb{p.gB, W, W, 0, 1};
  NL_TRY(run_gemm(xb, G_FC_T, &sp, 1, N, p.gO, 128, NL_ACT_NONE));
  // frozen weights, W = 128 / 256, K = 8: the attention's way back, the four (N x 8)-row products and the LeakyReLU masks in between as ONE launch that keeps the rows
  // in registers (point_bwd.hip); d query comes back from it
  const bool chain = !tg && K == 8 && pt_mask_bits(x) && pt_table(x) && !dbg_switch("NERFLOC_NO_BWD_CHAIN") && nl_point_bwd_chain_supported(W) && NK * 1024 <= 0x7fffffffll;
  const bool chain_att = chain && !dbg_switch("NERFLOC_NO_BWD_ATT");
  if (chain) {
    const unsigned* mk[3] = {p.mk[0], p.mk[1], p.mk[2]};
    if (chain_att) NL_TRY(nl_launch_point_bwd_chain(nullptr, mk, x.p<char>(x.L.pt_bwd_stream), p.gX, NK, W, x.st, p.Q, p.KV, p.gO, p.gQ));
    else {
      NL_TRY(nl_launch_attn_backward(p.Q, p.KV, p.gO, N, K, p.gQ, p.gKV, x.st));
      NL_TRY(nl_launch_point_bwd_chain(p.gKV, mk, x.p<char>(x.L.pt_bwd_stream), p.gX, NK, W, x.st));
    }
    if (g_G) {   // residual path + query projection
      NL_TRY(run_gemm(xb, G_Q_T, &sgq, 1, N, p.FCo, W, NL_ACT_NONE));   // (FCo is free from here on)
      NL_TRY(nl_launch_add(p.gpre, p.FCo, g_G, (size_t)N * W, x.st));
    }
    return nl_launch_point_encode_backward(xyz, dir, dir_stride, dir_div, N, K, M, idx, f->sp_xyz, f->sp_dir, x.p<float>(x.L.rd_w), inv_span, p.gX, 96, g_xyz, g_dir,
                                           nullptr, x.st);
  }
  NL_TRY(nl_launch_attn_backward(p.Q, p.KV, p.gO, N, K, p.gQ, p.gKV, x.st));
  NL_TRY(wg(T_WQ, -1, p.gQ, 128, 128, G, W, W, N));
  NL_TRY(wg(T_WK, -1, p.gKV, 256, 128, p.H3, W, W, NK));
  NL_TRY(wg(T_WV, -1, p.gKV + 128, 256, 128, p.H3, W, W, NK));
  if (g_G) {   // residual path + query projection
    NL_TRY(run_gemm(xb, G_Q_T, &sgq, 1, N, p.FCo, W, NL_ACT_NONE));   // (FCo is free from here on)
    NL_TRY(nl_launch_add(p.gpre, p.FCo, g_G, (size_t)N * W, x.st));
  }
  const bool bits = pt_mask_bits(x);
  NL_TRY(gemm_lrelu_masked(xb, G_KV_T, skv, NK, p.gA, W, p.H3, bits ? p.mk[2] : nullptr));
  NL_TRY(wg(T_B4W, T_B4B, p.gA, W, W, p.H2, W, W, NK));
  NL_TRY(gemm_lrelu_masked(xb, G_BASE4_T, sa, NK, p.gB, W, p.H2, bits ? p.mk[1] : nullptr));
  NL_TRY(wg(T_B2W, T_B2B, p.gB, W, W, p.H1, W, W, NK));
  NL_TRY(gemm_lrelu_masked(xb, G_BASE2_T, sb, NK, p.gA, W, p.H1, bits ? p.mk[0] : nullptr));
  const bool tab = pt_table(x);
  if (!tab) NL_TRY(wg(T_B0W, T_B0B, p.gA, W, W, p.X, ldx, F + 90, NK));
  else if (tg) {
    // base_mlp.0 on the table: its posenc / ray_diff_fc columns and the bias from the 96-wide rows; the feature columns and the support features through
    // d T = the rows' gradients summed per support point (M, W): d W[:, :F] = d T^T . sp_feature, d sp_feature = d T . W[:, :F]
    if (tg->w[T_B0W]) NL_TRY(nl_launch_wgrad(p.gA, W, W, p.X, 96, 90, NK, 0, 0, tg->w[T_B0W] + F, F + 90, 1, 0, tg->w[T_B0B], tg->scratch, tg->scratch_floats, x.st));
    else if (tg->w[T_B0B]) NL_TRY(nl_launch_colsum(p.gA, W, NK, W, tg->w[T_B0B], tg->scratch, x.st));
    if ((tg->w[T_B0W] || tg->sp_feat) && M > 0) {
      const int ldf = ldf_of(f->C);
      NL_CHECK_HIP(hipMemsetAsync(f->tr_gT, 0, sizeof(float) * (size_t)M * W, x.st));
      NL_TRY(nl_launch_sp_feat_scatter(p.gA, W, W, idx, N, K, M, f->tr_gT, x.st));
      if (tg->w[T_B0W]) {
        NL_TRY(nl_launch_copy_rows(f->sp_feat, F, f->tr_tmp, ldf, M, F, false, x.st));   // (rows of 195 floats are not 16-byte aligned)
        NL_TRY(nl_launch_wgrad(f->tr_gT, W, W, f->tr_tmp, ldf, F, M, 0, 0, tg->w[T_B0W], F + 90, 1, 0, nullptr, tg->scratch, tg->scratch_floats, x.st));
      }
      if (tg->sp_feat) {
        SegSpec st_{f->tr_gT, W, W, 0, 1};
        NL_TRY(run_gemm(xb, G_BASE0_TF, &st_, 1, M, f->tr_tmp, ldf, NL_ACT_NONE));
        NL_TRY(nl_launch_copy_rows(f->tr_tmp, ldf, tg->sp_feat, F, M, F, true, x.st));
      }
    }
  }
  NL_TRY(run_gemm(xb, G_BASE0_T, &sa, 1, NK, p.gX, 96, NL_ACT_NONE));
  const bool rdw = tg && (tg->w[T_RD0W] || tg->w[T_RD0B] || tg->w[T_RD2W] || tg->w[T_RD2B]);
  NL_TRY(nl_launch_point_encode_backward(xyz, dir, dir_stride, dir_div, N, K, M, idx, f->sp_xyz, f->sp_dir, x.p<float>(x.L.rd_w), inv_span, p.gX, 96, g_xyz,
                                         g_dir, rdw ? p.tr : nullptr, x.st));
  if (rdw) {   // ray_diff_fc (model.py:36-39): rows [input 4 | hidden 16 | d hidden 16 | d output 32]
    NL_TRY(wg(T_RD2W, T_RD2B, p.tr + 36, 68, 27, p.tr + 4, 68, 16, NK));
    NL_TRY(wg(T_RD0W, T_RD0B, p.tr + 20, 68, 16, p.tr, 68, 4, NK));
  }
  if (tg && tg->sp_feat && !tab) {   // the gathered support features (columns 0 .. F-1 of the encoded rows)
    const int ldf = ldf_of(f->C);
    NL_TRY(run_gemm(xb, G_BASE0_TF, &sa, 1, NK, p.gXF, ldf, NL_ACT_NONE));
    NL_TRY(nl_launch_sp_feat_scatter(p.gXF, ldf, F, idx, N, K, M, tg->sp_feat, x.st));
  }
  return NL_OK;
}
int do_point_backward(const Ctx& xb, const Ctx& x, const nl_frame* f, const float* xyz, const float* dir, int dir_stride, const float* G, int64_t N, int K,
                      const float* gFA, float* g_xyz, float* g_dir, float* g_G, const PtBwdBufs& p, const int* idx_in = nullptr, const float* d2_in = nullptr,
                      const TrainOut* tg = nullptr) {
  if (!tg && dir && pt_keep_fused_ok(x, f, N, K)) NL_TRY(pt_forward_keep_fused(x, f, xyz, dir, dir_stride, 1, G, N, p, idx_in, d2_in));
  else NL_TRY(pt_forward_staged(x, f, xyz, dir, dir_stride, 1, G, N, K, p, idx_in, d2_in));
  return pt_backward_only(xb, x, f, xyz, dir, dir_stride, 1, G, N, K, gFA, g_xyz, g_dir, g_G, p, idx_in, d2_in, tg);
}

// ---- input gradients of the multi-view aggregation and of the colour blend (frozen weights) ------------------------------------------
struct MvBwdBufs { float *vis, *dd, *g393, *t64, *G, *gA, *gt64, *gg393, *gvis, *gdd, *gpart, *bl1, *rgbv, *blA, *ghA, *gpf, *grgbv, *gang, *dtr, *btr, *ang; int* valid_s; };
void carve_mvb(Bump& b, const nl_config* c, int V, int64_t N, bool blend, MvBwdBufs& m, bool train = false) {
  const int W = c->W, ldg = ldg_of(c->C);
  m.vis = b.take<float>((size_t)V * N); m.dd = b.take<float>((size_t)V * N); m.gvis = b.take<float>((size_t)V * N); m.gdd = b.take<float>((size_t)V * N);
  m.gpart = b.take<float>((size_t)V * N * 3);
  m.g393 = b.take<float>((size_t)N * ldg); m.valid_s = b.take<int>((size_t)N);
  if (!blend) {
    m.t64 = b.take<float>((size_t)N * 64); m.G = b.take<float>((size_t)N * W); m.gA = b.take<float>((size_t)N * W);
    m.gt64 = b.take<float>((size_t)N * 64); m.gg393 = b.take<float>((size_t)N * ldg);
    m.bl1 = m.rgbv = m.blA = m.ghA = m.gpf = m.grgbv = m.gang = nullptr;
  } else {
    m.bl1 = b.take<float>((size_t)N * V * 32); m.rgbv = b.take<float>((size_t)N * V * 4); m.blA = b.take<float>((size_t)N * 32);
    m.ghA = b.take<float>((size_t)N * 32); m.gpf = b.take<float>((size_t)N * V * 32); m.grgbv = b.take<float>((size_t)N * V * 4);
    m.gang = b.take<float>((size_t)N * V * 4);
    m.t64 = m.G = m.gA = m.gt64 = m.gg393 = nullptr;
  }
  m.dtr = m.btr = m.ang = nullptr;
  if (train) {
    if (c->precision == NL_PREC_F32) m.dtr = b.take<float>((size_t)V * N * nl_dec_train_row());   // (the MFMA decoder backward needs no rows)
    if (blend) { m.btr = b.take<float>((size_t)V * N * 68); m.ang = b.take<float>((size_t)V * N * 8 + 256); }
  }
}

// the recomputed forward both need: visibility / depth difference (exact fp32 decoders: the backward kernel differentiates those) and the
// statistics rows (+ the blend's per-(sample, view) layer-1 part when bl1 != null)
int mv_recompute(const Ctx& x32, const nl_frame* f, const NlViews& vw, const float* xyz, int64_t N, const MvBwdBufs& m) {
  if (m.bl1) NL_TRY(ensure_pfeat(x32, f));
  if (x32.c->precision == NL_PREC_F32) NL_TRY(nl_launch_mv_vis(vw, f->visf_hwc, x32.p<float>(x32.L.dec_w), xyz, N, m.vis, m.dd, x32.st));
  else NL_TRY(nl_launch_mv_vis_mfma(vw, f->visf_hwc, x32.p<char>(x32.L.dec_mfma), xyz, N, m.vis, m.dd, true, x32.st));   // split-FP16 decoders (§2)
  return nl_launch_mv_stats(vw, f->views_dev, f->images, f->feat, f->C, xyz, N, m.vis, m.dd, m.g393, ldg_of(f->C), nullptr, nullptr, m.valid_s, f->pfeat,
                            x32.p<float>(x32.L.blw), m.bl1, m.rgbv, x32.st);
}

// the 24 decoder tensors from the rows the decoder backward kernels emit (backward.hip: [x 32 | per decoder: h1 32, h2 32, d a1 32, d a2 32, d out 2, pad 2])
int dec_wgrads(const TrainOut* tg, hipStream_t st, const float* tr, int64_t rows) {
  const int ld = nl_dec_train_row();
  for (int d = 0; d < 4; ++d) {
    const float* q = tr + 32 + 132 * d;
    const int t0 = T_DEC + 6 * d;
    NL_TRY(wgrad_to(tg, st, t0, t0 + 1, q + 64, ld, 32, tr, ld, 32, rows));
    NL_TRY(wgrad_to(tg, st, t0 + 2, t0 + 3, q + 96, ld, 32, q, ld, 32, rows));
    NL_TRY(wgrad_to(tg, st, t0 + 4, t0 + 5, q + 128, ld, d < 2 ? 2 : 1, q + 32, ld, 32, rows));
  }
  return NL_OK;
}

// g_G (N, W) -> g_xyz (N, 3): out_fc backwards (two transposed-weight products, ELU masks), the visibility-weighted statistics, the bilinear taps'
// spatial derivative, the IBRNet projection; visibility / depth difference through the NeuRay decoders and the NeuRay projection.
// out_fc on the recomputed statistics rows -> m.t64, m.G
int mv_outfc_forward(const Ctx& x32, const nl_frame* f, int64_t N, const MvBwdBufs& m) {
  const int W = x32.c->W, ldg = ldg_of(f->C);
  SegSpec s0{m.g393, ldg, ldg, 0, 1}, s1{m.t64, 64, 64, 0, 1};
  NL_TRY(run_gemm(x32, G_OUTFC0, &s0, 1, N, m.t64, 64, NL_ACT_ELU));
  return run_gemm(x32, G_OUTFC2, &s1, 1, N, m.G, W, NL_ACT_ELU);
}
// gG (N, W) -> m.gg393 (the statistics rows' gradient) + out_fc's weight gradients
int mv_outfc_backward(const Ctx& xb, const Ctx& x32, const nl_frame* f, int64_t N, const float* gG, const MvBwdBufs& m, const TrainOut* tg) {
  const int W = x32.c->W, ldg = ldg_of(f->C);
  NL_CHECK_HIP(hipMemcpyAsync(m.gA, gG, sizeof(float) * (size_t)N * W, hipMemcpyDeviceToDevice, x32.st));
  NL_TRY(nl_launch_elu_mask(m.gA, m.G, (size_t)N * W, x32.st));
  NL_TRY(wgrad_to(tg, x32.st, T_OUT2W, T_OUT2B, m.gA, W, W, m.t64, 64, 64, N));
  SegSpec sa{m.gA, W, W, 0, 1}, st{m.gt64, 64, 64, 0, 1};
  NL_TRY(run_gemm(xb, G_OUTFC2_T, &sa, 1, N, m.gt64, 64, NL_ACT_NONE));
  NL_TRY(nl_launch_elu_mask(m.gt64, m.t64, (size_t)N * 64, x32.st));
  NL_TRY(wgrad_to(tg, x32.st, T_OUT0W, T_OUT0B, m.gt64, 64, 64, m.g393, ldg, 2 * (f->C + 3) + 3, N));
  return run_gemm(xb, G_OUTFC0_T, &st, 1, N, m.gg393, ldg, NL_ACT_NONE);
}
// gradients of the tapped values (statistics rows: gg393; blend: g_pf / g_rgbv / g_ang; either may be null) -> g_xyz (written), g_qc, the maps' scatter-adds;
// then visibility / depth difference back through the decoders (ONE pass for whatever consumers contributed) -> += g_xyz, the decoders' gradients
int mv_geom_dec_backward(const Ctx& x32, const nl_frame* f, const NlViews& vw, const float* xyz, int64_t N, const float* gg393, bool blend, float* g_xyz,
                         float* g_qc, const MvBwdBufs& m, const TrainOut* tg) {
  NL_TRY(nl_launch_mv_geom_backward(vw, f->views_dev, f->images, f->feat, f->C, blend ? f->pfeat : nullptr, xyz, N, m.vis, m.dd, gg393, ldg_of(f->C),
                                    blend ? m.gpf : nullptr, blend ? m.grgbv : nullptr, blend ? m.gang : nullptr, g_xyz, g_qc, m.gvis, m.gdd,
                                    tg ? tg->feat_maps : nullptr, tg && blend ? tg->pfeat_maps : nullptr, gg393 ? m.g393 : nullptr, x32.st));
  const bool decw = tg && tg->any(T_DEC, T_DEC + 24);
  const bool f32 = x32.c->precision == NL_PREC_F32;   // fp32: rows for dec_wgrads; otherwise the MFMA kernel accumulates the 24 tensors' gradients itself
  NL_TRY(nl_launch_dec_backward(vw, f->visf_hwc, x32.p<float>(x32.L.dec_w), f32 ? nullptr : x32.p<char>(x32.L.dec_mfma), xyz, N, m.gvis, m.gdd, m.gpart, g_xyz,
                                decw && f32 ? m.dtr : nullptr, decw && !f32 ? tg->w + T_DEC : nullptr, tg ? tg->scratch : nullptr, tg ? tg->scratch_floats : 0,
                                tg ? tg->vis_maps : nullptr, x32.st));
  return decw && f32 ? dec_wgrads(tg, x32.st, m.dtr, (int64_t)vw.V * N) : NL_OK;
}
int do_mv_backward(const Ctx& xb, const Ctx& x32, const nl_frame* f, const float* xyz, int64_t N, const float* gG, float* g_xyz, const MvBwdBufs& m,
                   const TrainOut* tg = nullptr) {
  const NlViews vw = with_query(f, nullptr);
  NL_TRY(mv_recompute(x32, f, vw, xyz, N, m));
  NL_TRY(mv_outfc_forward(x32, f, N, m));
  NL_TRY(mv_outfc_backward(xb, x32, f, N, gG, m, tg));
  return mv_geom_dec_backward(x32, f, vw, xyz, N, m.gg393, false, g_xyz, nullptr, m, tg);
}

// rgb_s = blend(feature_agg, per-view taps) forward (staged) and its input gradient
int do_blend_forward(const Ctx& x, const nl_frame* f, const float* qc, const float* xyz, const float* FA, int64_t N, float* rgb_s, const MvBwdBufs& m) {
  const int W = x.c->W;
  const NlViews vw = with_query(f, qc);
  nl_config c32 = *x.c;   // the same arithmetic as the backward call's recomputed forward
  c32.precision = x.c->precision == NL_PREC_F32 ? NL_PREC_F32 : NL_PREC_F16X3_INTERNAL;
  Ctx x32 = x; x32.c = &c32;
  NL_TRY(mv_recompute(x32, f, vw, xyz, N, m));
  SegSpec sa{FA, W, W, 0, 1};
  NL_TRY(run_gemm(x32, G_BLENDA, &sa, 1, N, m.blA, 32, NL_ACT_NONE));
  return nl_launch_blend(m.blA, m.bl1, m.rgbv, N, vw.V, x.p<float>(x.L.bl2_w), x.p<float>(x.L.bl2_b), x.p<float>(x.L.bl4_w), x.p<float>(x.L.bl4_b), rgb_s, x.st);
}

// g_rgb_s -> m.ghA / m.gpf / m.grgbv / m.gang (+ rgb_blending_mlp's weight gradients) and g_FA (may be null); needs m.blA, m.bl1, m.rgbv of the forward
int blend_tail_backward(const Ctx& xb, const Ctx& x32, const nl_frame* f, const NlViews& vw, const float* xyz, const float* FA, int64_t N, const float* g_rgb_s,
                        float* g_FA, const MvBwdBufs& m, const TrainOut* tg) {
  const int W = x32.c->W;
  SegSpec sg{m.ghA, 32, 32, 0, 1};
  const bool blw = tg && (tg->any(T_BL0W, T_BL4B + 1));
  NL_TRY(nl_launch_blend_backward(m.blA, m.bl1, m.rgbv, N, vw.V, x32.p<float>(x32.L.bl2_w), x32.p<float>(x32.L.bl2_b), x32.p<float>(x32.L.bl4_w),
                                  x32.p<float>(x32.L.bl4_b), x32.p<float>(x32.L.blw), g_rgb_s, m.ghA, m.gpf, m.grgbv, m.gang, blw ? m.btr : nullptr, x32.st));
  if (blw) {   // rgb_blending_mlp (model.py:84-93, 532-535)
    const int64_t NV = N * vw.V;
    const int F = f->C + 3;
    NL_TRY(wgrad_to(tg, x32.st, T_BL2W, T_BL2B, m.btr + 32, 68, 16, m.btr, 68, 32, NV));
    NL_TRY(wgrad_to(tg, x32.st, T_BL4W, T_BL4B, m.btr + 64, 68, 1, m.btr + 48, 68, 16, NV));
    if (tg->w[T_BL0W]) {
      // layer 1 by linearity: the feature_agg columns (per sample), the [rgb | visibility | view angles] columns (per sample and view); the feature
      // columns multiply the per-frame projected maps, whose gradient goes back as a map (nl_train_grads.blend_feat_maps)
      NL_TRY(nl_launch_wgrad(m.ghA, 32, 32, FA, W, W, N, 0, 0, tg->w[T_BL0W], W + F + 5, 1, 0, nullptr, tg->scratch, tg->scratch_floats, x32.st));
      float* t8 = m.ang + (size_t)NV * 8;
      NL_CHECK_HIP(hipMemsetAsync(t8, 0, sizeof(float) * 256, x32.st));
      NL_TRY(nl_launch_blend_inputs8(vw, f->views_dev, xyz, N, m.rgbv, m.ang, x32.st));
      NL_TRY(nl_launch_wgrad(m.gpf, 32, 32, m.ang, 8, 8, NV, 0, 0, t8, 8, 1, 0, nullptr, tg->scratch, tg->scratch_floats, x32.st));
      NL_TRY(nl_launch_blw_unpack(t8, tg->w[T_BL0W], W, F, x32.st));
    }
    if (tg->w[T_BL0B]) NL_TRY(nl_launch_colsum(m.gpf, 32, NV, 32, tg->w[T_BL0B], tg->scratch, x32.st));
  }
  if (g_FA) NL_TRY(run_gemm(xb, G_BLENDA_T, &sg, 1, N, g_FA, W, NL_ACT_NONE));
  return NL_OK;
}
int do_blend_backward(const Ctx& xb, const Ctx& x32, const nl_frame* f, const float* qc, const float* xyz, const float* FA, int64_t N, const float* g_rgb_s,
                      float* g_xyz, float* g_FA, float* g_qc, const MvBwdBufs& m, const TrainOut* tg = nullptr) {
  const int W = x32.c->W;
  const NlViews vw = with_query(f, qc);
  NL_TRY(mv_recompute(x32, f, vw, xyz, N, m));
  SegSpec sa{FA, W, W, 0, 1};
  NL_TRY(run_gemm(x32, G_BLENDA, &sa, 1, N, m.blA, 32, NL_ACT_NONE));
  NL_TRY(blend_tail_backward(xb, x32, f, vw, xyz, FA, N, g_rgb_s, g_FA, m, tg));
  return mv_geom_dec_backward(x32, f, vw, xyz, N, nullptr, true, g_xyz, g_qc, m, tg);
}

// sigma_out (optional): when conv_out's LayerNorm runs inside the GEMM, the density head is evaluated there too and
// *sigma_done is set; otherwise the caller runs sigma_kernel on geo
// need_geo == false: the caller only wants the density (model.py:525 is the sole consumer of the U-Net's output); when the density
// head runs inside conv_out's epilogue the (N, W) output rows are then never written (0.5 GB per config-2 batch)
// in_frag: `in` is feature_agg as the chain kernel's fragment image (NlGemmSeg::frag) instead of fp32 rows
// fuse_inner: the caller needs nothing of the five inner layers but their result (u.x2) — the inference render path; the backward passes and the stage entry
// point keep the separate launches (their pre-LayerNorm outputs and block outputs are read back)
int do_unet(const Ctx& x, const float* in, int64_t R, float* geo, const UnBufs& u, float* sigma_out = nullptr, bool* sigma_done = nullptr,
            bool need_geo = true, int in_frag = 0, bool fuse_inner = false) {   // in_frag: NlGemmSeg::frag of `in` (0: fp32 rows, 1: fragment image, 3: its split-FP16 form)
  const int W = x.c->W, S = x.c->S;
  // the two phases of every transposed convolution as one launch (bf16 modes; the fp32 kernels keep the separate phases)
  static const bool no_merge = dbg_switch("NERFLOC_NO_TMERGE");
  const bool merged = x.c->precision != NL_PREC_F32 && !no_merge;
  auto g = [&](int i) { return x.p<float>(x.L.un_g[i]); };
  auto b = [&](int i) { return x.p<float>(x.L.un_b[i]); };
  auto gl = [&](int i) { return x.p<float>(x.L.un_gl[i]); };   // accumulator-lane order: what the GEMM's fused LayerNorm reads
  auto bl = [&](int i) { return x.p<float>(x.L.un_bl[i]); };
  const float eps = 1e-5f;
  // feature_agg's two consumers take their weights with the channels of every 32-block in accumulator order in the bf16 modes (G_CONV1F / G_CONVOUTF, streaming
  // kernel): from the chain kernel's fragment image (in_frag) or from fp32 rows read in that order — the same products in the same order either way
  const bool korder = in_frag || ((x.c->precision == NL_PREC_BF16X3 || x.c->precision == NL_PREC_BF16) && !dbg_switch("NERFLOC_NO_FRAG") && (((size_t)in) & 15) == 0 &&
                                  ((x.has_bst >> G_CONV1F) & 1) && ((x.has_bst >> G_CONVOUTF) & 1));
  const int fa_mode = in_frag ? in_frag : (korder ? 2 : 0);
  {  // conv1: W -> 64 over S
    SegSpec s[1] = {{in, W, W, 0, 1, 3, fa_mode}};   // 3 taps, interleaved per 32-channel block
    const RowEpi ep{nullptr, 0, gl(U_CONV1), bl(U_CONV1), nullptr, eps, u.c1, NL_EPI_LNSLAB, 1};   // LN + ELU + MaxPool inside the GEMM when one workgroup = one ray
    bool fused = false;
    NL_TRY(run_gemm(x, korder ? G_CONV1F : G_CONV1, s, 1, R * S, u.r1, 64, NL_ACT_NONE, S, S, S, 1, 0, &ep, &fused));
    if (!fused) NL_TRY(nl_launch_ln_slab_elu(u.r1, R, S, 64, g(U_CONV1), b(U_CONV1), eps, nullptr, u.c1, x.st));
  }
  // conv2 ... trans_conv1 as one kernel (unet_inner.hip: a pair of rays per workgroup, every slab in LDS): S = 128, the streaming kernels' weight images
  static const bool no_inner = dbg_switch("NERFLOC_NO_UNET_INNER");
  const uint64_t inner_bits = (1ull << G_CONV2) | (1ull << G_CONV3) | (1ull << G_T3M) | (1ull << G_T2M) | (1ull << G_T1M);
  if (fuse_inner && merged && !no_inner && nl_unet_inner_supported(S, x.c->precision) && (x.has_bst & inner_bits) == inner_bits) {
    NlUnetInnerArgs ia;
    ia.c1 = u.c1; ia.c2 = u.c2; ia.x2 = u.x2; ia.R = (int)R; ia.eps = eps;
    const int gs[5] = {G_CONV2, G_CONV3, G_T3M, G_T2M, G_T1M}, us[5] = {U_CONV2, U_CONV3, U_T3, U_T2, U_T1};
    for (int i = 0; i < 5; ++i) { ia.w[i] = x.pk + x.L.bst[gs[i]]; ia.bias[i] = x.p<float>(x.L.bias[gs[i]]); ia.gl[i] = gl(us[i]); ia.bl[i] = bl(us[i]); }
    NL_TRY(nl_launch_unet_inner(ia, x.c->precision, x.st));
  } else {
  {  // conv2: 64 -> 128 over S/2
    SegSpec s[1] = {{u.c1, 64, 64, 0, 1, 3}};
    const RowEpi ep{nullptr, 0, gl(U_CONV2), bl(U_CONV2), nullptr, eps, u.c2, NL_EPI_LNSLAB, 1};   // two rays per workgroup at S = 128
    bool fused = false;
    NL_TRY(run_gemm(x, G_CONV2, s, 1, R * (S / 2), u.r2, 128, NL_ACT_NONE, S / 2, S / 2, S / 2, 1, 0, &ep, &fused));
    if (!fused) NL_TRY(nl_launch_ln_slab_elu(u.r2, R, S / 2, 128, g(U_CONV2), b(U_CONV2), eps, nullptr, u.c2, x.st));
  }
  {  // conv3: 128 -> 128 over S/4
    SegSpec s[1] = {{u.c2, 128, 128, 0, 1, 3}};
    const RowEpi ep{nullptr, 0, gl(U_CONV3), bl(U_CONV3), nullptr, eps, u.c3, NL_EPI_LNSLAB, 1};   // one ray per wave at S = 128
    bool fused = false;
    NL_TRY(run_gemm(x, G_CONV3, s, 1, R * (S / 4), u.r3, 128, NL_ACT_NONE, S / 4, S / 4, S / 4, 1, 0, &ep, &fused));
    if (!fused) NL_TRY(nl_launch_ln_slab_elu(u.r3, R, S / 4, 128, g(U_CONV3), b(U_CONV3), eps, nullptr, u.c3, x.st));
  }
  {  // trans_conv3: S/8 -> S/4
    const int Li = S / 8, Lo = S / 4;
    SegSpec o[2] = {{u.c3, 128, 128, 0, 1}, {u.c3, 128, 128, 1, 1}};
    bool tfused = false;
    // row m = output positions 2m, 2m+1: the (Li x 2 co) view of the merged output IS the ray's (Lo x co) slab, so LayerNorm([C, L]) + ELU
    // ride in the GEMM's epilogue (the affine tables are stored position-major: same memory either way)
    const RowEpi ept3{nullptr, 0, gl(U_T3), bl(U_T3), nullptr, eps, u.x0, NL_EPI_LNSLAB, 0};
    if (merged) NL_TRY(run_gemm(x, G_T3M, o, 2, R * Li, u.x0r, 256, NL_ACT_NONE, Li, Li, Li, 1, 0, &ept3, &tfused));
    else {
      SegSpec e[1] = {{u.c3, 128, 128, 0, 1}};
      NL_TRY(run_gemm(x, G_T3E, e, 1, R * Li, u.x0r, 128, NL_ACT_NONE, Li, Li, Lo, 2, 0));
      NL_TRY(run_gemm(x, G_T3O, o, 2, R * Li, u.x0r, 128, NL_ACT_NONE, Li, Li, Lo, 2, 1));
    }
    if (!tfused) NL_TRY(nl_launch_ln_slab_elu(u.x0r, R, Lo, 128, g(U_T3), b(U_T3), eps, u.x0, nullptr, x.st));
  }
  {  // trans_conv2 on cat[c2, x0]: S/4 -> S/2
    const int Li = S / 4, Lo = S / 2;
    SegSpec o[4] = {{u.c2, 128, 128, 0, 1}, {u.x0, 128, 128, 0, 1}, {u.c2, 128, 128, 1, 1}, {u.x0, 128, 128, 1, 1}};
    bool tfused = false;
    const RowEpi ept2{nullptr, 0, gl(U_T2), bl(U_T2), nullptr, eps, u.x1, NL_EPI_LNSLAB, 0};
    if (merged) NL_TRY(run_gemm(x, G_T2M, o, 4, R * Li, u.x1r, 128, NL_ACT_NONE, Li, Li, Li, 1, 0, &ept2, &tfused));
    else {
      SegSpec e[2] = {{u.c2, 128, 128, 0, 1}, {u.x0, 128, 128, 0, 1}};
      NL_TRY(run_gemm(x, G_T2E, e, 2, R * Li, u.x1r, 64, NL_ACT_NONE, Li, Li, Lo, 2, 0));
      NL_TRY(run_gemm(x, G_T2O, o, 4, R * Li, u.x1r, 64, NL_ACT_NONE, Li, Li, Lo, 2, 1));
    }
    if (!tfused) NL_TRY(nl_launch_ln_slab_elu(u.x1r, R, Lo, 64, g(U_T2), b(U_T2), eps, u.x1, nullptr, x.st));
  }
  {  // trans_conv1 on cat[c1, x1]: S/2 -> S
    const int Li = S / 2, Lo = S;
    SegSpec o[4] = {{u.c1, 64, 64, 0, 1}, {u.x1, 64, 64, 0, 1}, {u.c1, 64, 64, 1, 1}, {u.x1, 64, 64, 1, 1}};
    bool tfused = false;
    const RowEpi ept1{nullptr, 0, gl(U_T1), bl(U_T1), nullptr, eps, u.x2, NL_EPI_LNSLAB, 0};
    if (merged) NL_TRY(run_gemm(x, G_T1M, o, 4, R * Li, u.x2r, 64, NL_ACT_NONE, Li, Li, Li, 1, 0, &ept1, &tfused));
    else {
      SegSpec e[2] = {{u.c1, 64, 64, 0, 1}, {u.x1, 64, 64, 0, 1}};
      NL_TRY(run_gemm(x, G_T1E, e, 2, R * Li, u.x2r, 32, NL_ACT_NONE, Li, Li, Lo, 2, 0));
      NL_TRY(run_gemm(x, G_T1O, o, 4, R * Li, u.x2r, 32, NL_ACT_NONE, Li, Li, Lo, 2, 1));
    }
    if (!tfused) NL_TRY(nl_launch_ln_slab_elu(u.x2r, R, Lo, 32, g(U_T1), b(U_T1), eps, u.x2, nullptr, x.st));
  }
  }
  {  // conv_out on cat[in, x2]
    SegSpec s[2] = {{in, W, W, 0, 1, 3, fa_mode}, {u.x2, 32, 32, 0, 1, 3}};
    RowEpi ep{nullptr, 0, gl(U_OUT), bl(U_OUT), nullptr, eps, geo, NL_EPI_LNSLAB, 0};
    if (sigma_out) { ep.sig_w = x.p<float>(x.L.sig_w); ep.sig_b = x.p<float>(x.L.sig_b); ep.sig_out = sigma_out; }
    bool fused = false;
    if (!need_geo && sigma_out) ep.out = nullptr;   // (the unfused fallback below still writes `geo`)
    NL_TRY(run_gemm(x, korder ? G_CONVOUTF : G_CONVOUT, s, 2, R * S, u.outr, W, NL_ACT_NONE, S, S, S, 1, 0, &ep, &fused));
    if (!fused) NL_TRY(nl_launch_ln_slab_elu(u.outr, R, S, W, g(U_OUT), b(U_OUT), eps, geo, nullptr, x.st));
    if (sigma_done) *sigma_done = fused && sigma_out;
  }
  return NL_OK;
}

// ---- input gradient of the ray U-Net (frozen weights) -------------------------------------------------------------------------------
struct UnBwdBufs { UnBufs u; float *geo, *gout, *gx2, *gx2r, *gcat1, *gx1r, *gcat2, *gx0r, *gc3, *gr3, *gc2, *gr2, *gc1, *gr1, *tmp, *aff; };
void carve_unb(Bump& b, const nl_config* c, int64_t R, UnBwdBufs& q, bool train = false) {
  const size_t N = (size_t)R * c->S;
  const int W = c->W;
  carve_un(b, c, R, q.u);
  q.geo = b.take<float>(N * W); q.gout = b.take<float>(N * W);
  q.gx2 = b.take<float>(N * 32); q.gx2r = b.take<float>(N * 32);
  q.gcat1 = b.take<float>(N / 2 * 128); q.gx1r = b.take<float>(N / 2 * 64);
  q.gcat2 = b.take<float>(N / 4 * 256); q.gx0r = b.take<float>(N / 4 * 128);
  q.gc3 = b.take<float>(N / 8 * 128); q.gr3 = b.take<float>(N / 4 * 128);
  q.gc2 = b.take<float>(N / 4 * 128); q.gr2 = b.take<float>(N / 2 * 128);
  q.gc1 = b.take<float>(N / 2 * 64); q.gr1 = b.take<float>(N * 64);
  q.tmp = b.take<float>(N * W);
  q.aff = train ? b.take<float>(2 * N * (W > 64 ? W : 64)) : nullptr;   // the largest slab per ray (conv_out: S x W; conv1: S x 64; conv2: S/2 x 128), [d y * xhat | d y]
}

// Unfused forward in exact fp32 (every layer's pre-LayerNorm output stays in the workspace), then layer by layer backwards: LayerNorm / ELU /
// MaxPool derivative (one block per ray) -> transposed-weight convolution (segment GEMM over the gradient rows' taps), the skip connections'
// gradients added where the concatenations were.
int unet_backward_only(const Ctx& xb, const Ctx& x32, const float* in, int64_t R, const float* g_geo, float* g_in, const UnBwdBufs& q, const TrainOut* tg);
int do_unet_backward(const Ctx& xb, const Ctx& x32, const float* in, int64_t R, const float* g_geo, float* g_in, const UnBwdBufs& q, const TrainOut* tg = nullptr) {
  NL_TRY(do_unet(x32, in, R, q.geo, q.u));   // fp32: separate GEMM + ln_slab_elu launches
  return unet_backward_only(xb, x32, in, R, g_geo, g_in, q, tg);
}
// (the forward's pre-LayerNorm outputs and block outputs are in q.u)
int unet_backward_only(const Ctx& xb, const Ctx& x32, const float* in, int64_t R, const float* g_geo, float* g_in, const UnBwdBufs& q, const TrainOut* tg) {
  const int W = x32.c->W, S = x32.c->S;
  const UnBufs& u = q.u;
  auto g = [&](int i) { return x32.p<float>(x32.L.un_g[i]); };
  auto b = [&](int i) { return x32.p<float>(x32.L.un_b[i]); };
  const float eps = 1e-5f;
  hipStream_t st = x32.st;
  // LayerNorm / ELU / MaxPool backward of block `li`; training: + its affine tables' gradients (column sums over the rays, transposed into the
  // state_dict's (C, L) layout)
  auto ln_bwd = [&](int li, const float* x, int L, int Cc, const float* go, int ldgo, int pool, float* gx) -> int {
    float* gw = tg ? tg->w[T_UNET + 4 * li + 2] : nullptr;
    float* gb = tg ? tg->w[T_UNET + 4 * li + 3] : nullptr;
    const bool aff = gw || gb;
    NL_TRY(nl_launch_ln_slab_elu_backward(x, R, L, Cc, g(li), b(li), eps, go, ldgo, pool, gx, aff ? q.aff : nullptr, st));
    if (!aff) return NL_OK;
    return nl_launch_colsum_tables(q.aff, R, L, Cc, gw, gb, tg->scratch, st);   // sums over the rays, straight into the channel-major tables
  };
  // Conv1d(k = 3, padding 1) weight (co, ci_total, 3): one product per tap, the input rows shifted by tap - 1 inside each ray
  auto conv_wg = [&](int li, const float* dY, int co, const float* X, int ci, int ci_total, int ci0, int L, bool bias) -> int {
    float* gw = tg ? tg->w[T_UNET + 4 * li] : nullptr;
    float* gb = tg && bias ? tg->w[T_UNET + 4 * li + 1] : nullptr;
    if (!gw && gb) return nl_launch_colsum(dY, co, R * L, co, gb, tg->scratch, st);
    if (!gw) return NL_OK;
    const float* dys[3] = {dY, dY, dY};
    const float* xs[3] = {X, X, X};
    const int sh[3] = {-1, 0, 1}, cos_[3] = {ci0 * 3, ci0 * 3 + 1, ci0 * 3 + 2};
    return nl_launch_wgrad_multi(3, dys, co, co, xs, ci, ci, R * L, sh, L, gw, ci_total * 3, 3, cos_, gb, 1, tg->scratch, tg->scratch_floats, st);   // the three taps
  };
  // ConvTranspose1d(k = 3, stride 2, padding 1, output_padding 1) weight (ci_total, co, 3): y[2m] = x[m] w1, y[2m+1] = x[m] w2 + x[m+1] w0;
  // gy = the merged rows (R Li, 2 co) [even | odd]
  auto convT_wg = [&](int li, const float* X, int ci, int ci0, const float* gy, int co, int Li) -> int {
    float* gw = tg ? tg->w[T_UNET + 4 * li] : nullptr;
    if (!gw) return NL_OK;
    float* base = gw + (size_t)ci0 * co * 3;
    const float* dys[3] = {X, X, X};                       // (operand roles swapped: the weight's rows are the INPUT channels)
    const float* xs[3] = {gy, gy + co, gy + co};
    const int sh[3] = {0, 0, -1}, cos_[3] = {1, 2, 0};
    return nl_launch_wgrad_multi(3, dys, ci, ci, xs, 2 * co, co, R * Li, sh, Li, base, co * 3, 3, cos_, nullptr, -1, tg->scratch, tg->scratch_floats, st);
  };
  auto convT_bias = [&](int li, const float* gy, int co, int Lo) -> int {
    float* gb = tg ? tg->w[T_UNET + 4 * li + 1] : nullptr;
    return gb ? nl_launch_colsum(gy, co, R * Lo, co, gb, tg->scratch, st) : NL_OK;
  };
  // conv_out
  NL_TRY(ln_bwd(U_OUT, u.outr, S, W, g_geo, W, 0, q.gout));
  NL_TRY(conv_wg(U_OUT, q.gout, W, in, W, W + 32, 0, S, true));
  NL_TRY(conv_wg(U_OUT, q.gout, W, u.x2, 32, W + 32, W, S, false));
  { SegSpec s[1] = {{q.gout, W, W, 0, 1, 3}};
    NL_TRY(run_gemm(xb, G_UB_OUTA, s, 1, R * S, g_in, W, NL_ACT_NONE, S, S, S, 1, 0));
    NL_TRY(run_gemm(xb, G_UB_OUTB, s, 1, R * S, q.gx2, 32, NL_ACT_NONE, S, S, S, 1, 0)); }
  // trans_conv1: slab (S x 32) = merged rows (S/2 x 64)
  NL_TRY(ln_bwd(U_T1, u.x2r, S, 32, q.gx2, 32, 0, q.gx2r));
  NL_TRY(convT_wg(U_T1, u.c1, 64, 0, q.gx2r, 32, S / 2));
  NL_TRY(convT_wg(U_T1, u.x1, 64, 64, q.gx2r, 32, S / 2));
  NL_TRY(convT_bias(U_T1, q.gx2r, 32, S));
  { SegSpec s[2] = {{q.gx2r, 64, 64, 0, 1}, {q.gx2r + 32, 64, 32, -1, 1}};
    NL_TRY(run_gemm(xb, G_UB_T1, s, 2, R * (S / 2), q.gcat1, 128, NL_ACT_NONE, S / 2, S / 2, S / 2, 1, 0)); }
  // trans_conv2: output x1 = columns 64..127 of cat[c1, x1]'s gradient
  NL_TRY(ln_bwd(U_T2, u.x1r, S / 2, 64, q.gcat1 + 64, 128, 0, q.gx1r));
  NL_TRY(convT_wg(U_T2, u.c2, 128, 0, q.gx1r, 64, S / 4));
  NL_TRY(convT_wg(U_T2, u.x0, 128, 128, q.gx1r, 64, S / 4));
  NL_TRY(convT_bias(U_T2, q.gx1r, 64, S / 2));
  { SegSpec s[2] = {{q.gx1r, 128, 128, 0, 1}, {q.gx1r + 64, 128, 64, -1, 1}};
    NL_TRY(run_gemm(xb, G_UB_T2, s, 2, R * (S / 4), q.gcat2, 256, NL_ACT_NONE, S / 4, S / 4, S / 4, 1, 0)); }
  // trans_conv3: output x0 = columns 128..255 of cat[c2, x0]'s gradient
  NL_TRY(ln_bwd(U_T3, u.x0r, S / 4, 128, q.gcat2 + 128, 256, 0, q.gx0r));
  NL_TRY(convT_wg(U_T3, u.c3, 128, 0, q.gx0r, 128, S / 8));
  NL_TRY(convT_bias(U_T3, q.gx0r, 128, S / 4));
  { SegSpec s[2] = {{q.gx0r, 256, 256, 0, 1}, {q.gx0r + 128, 256, 128, -1, 1}};
    NL_TRY(run_gemm(xb, G_UB_T3, s, 2, R * (S / 8), q.gc3, 128, NL_ACT_NONE, S / 8, S / 8, S / 8, 1, 0)); }
  // conv3 (+ MaxPool): gradient of its pooled output c3
  NL_TRY(ln_bwd(U_CONV3, u.r3, S / 4, 128, q.gc3, 128, 1, q.gr3));
  NL_TRY(conv_wg(U_CONV3, q.gr3, 128, u.c2, 128, 128, 0, S / 4, true));
  { SegSpec s[1] = {{q.gr3, 128, 128, 0, 1, 3}};
    NL_TRY(run_gemm(xb, G_UB_C3, s, 1, R * (S / 4), q.tmp, 128, NL_ACT_NONE, S / 4, S / 4, S / 4, 1, 0)); }
  NL_TRY(nl_launch_add2d(q.gcat2, 256, q.tmp, 128, q.gc2, 128, R * (S / 4), 128, st));
  // conv2 (+ MaxPool)
  NL_TRY(ln_bwd(U_CONV2, u.r2, S / 2, 128, q.gc2, 128, 1, q.gr2));
  NL_TRY(conv_wg(U_CONV2, q.gr2, 128, u.c1, 64, 64, 0, S / 2, true));
  { SegSpec s[1] = {{q.gr2, 128, 128, 0, 1, 3}};
    NL_TRY(run_gemm(xb, G_UB_C2, s, 1, R * (S / 2), q.tmp, 64, NL_ACT_NONE, S / 2, S / 2, S / 2, 1, 0)); }
  NL_TRY(nl_launch_add2d(q.gcat1, 128, q.tmp, 64, q.gc1, 64, R * (S / 2), 64, st));
  // conv1 (+ MaxPool)
  NL_TRY(ln_bwd(U_CONV1, u.r1, S, 64, q.gc1, 64, 1, q.gr1));
  NL_TRY(conv_wg(U_CONV1, q.gr1, 64, in, W, W, 0, S, true));
  { SegSpec s[1] = {{q.gr1, 64, 64, 0, 1, 3}};
    NL_TRY(run_gemm(xb, G_UB_C1, s, 1, R * S, q.tmp, W, NL_ACT_NONE, S, S, S, 1, 0)); }
  return nl_launch_add2d(g_in, W, q.tmp, W, g_in, W, R * S, W, st);
}

// the part of the heads that needs feature_agg only (not the density): feat_mlp.0, the per-sample blend projection, the blend tail
// term == true: n_alive / tile_list of `h` are valid (nl_launch_termination ran): dead samples are skipped
struct BlendTaps { NlViews vw; const float* viewsdev; const float* pfeat; const float* xyz; };   // bl1 == null: the blend tail recomputes its per-(sample, view) rows
int do_heads_pre(const Ctx& x, int V, const float* FA, const float* bl1, const float* rgbv, int64_t N, bool want_feat, const HdBufs& h, int parts = 7,
                 bool term = false, const BlendTaps* bt = nullptr) {
  const int W = x.c->W;
  if (want_feat && (parts & 1)) {
    SegSpec s0{FA, W, W, 0, 1};
    const TileMap tm{h.tile_list, h.tile_count};
    NL_TRY(run_gemm(x, G_FEAT0, &s0, 1, N, h.fth, W, NL_ACT_LRELU, 0, 0, 0, 1, 0, nullptr, nullptr, term ? &tm : nullptr));
  }
  SegSpec sa{FA, W, W, 0, 1};
  if (parts & 2) NL_TRY(run_gemm(x, G_BLENDA, &sa, 1, N, h.blA, 32, NL_ACT_NONE));
  if ((parts & 4) && !bl1) {
    if (!bt) return NL_ERR_BAD_ARG;
    NL_TRY(nl_launch_blend_taps(bt->vw, bt->viewsdev, bt->pfeat, x.p<float>(x.L.blw), bt->xyz, h.blA, rgbv, N, x.p<float>(x.L.bl2_w), x.p<float>(x.L.bl2_b),
                                x.p<float>(x.L.bl4_w), x.p<float>(x.L.bl4_b), h.rgb_s, x.st, term ? h.n_alive : nullptr, x.c->S));
  } else if (parts & 4) NL_TRY(nl_launch_blend(h.blA, bl1, rgbv, N, V, x.p<float>(x.L.bl2_w), x.p<float>(x.L.bl2_b), x.p<float>(x.L.bl4_w),
                                               x.p<float>(x.L.bl4_b), h.rgb_s, x.st, term ? h.n_alive : nullptr, x.c->S));
  return NL_OK;
}

int do_heads(const Ctx& x, int V, const float* z, const float* FA, const float* geo, const float* bl1, const float* rgbv,
             const int* valid_s, int64_t R, int white, const nl_render_out* out, int64_t ray0, const HdBufs& h, bool have_sigma = false,
             bool pre_done = false, float term_eps = 0.f, int chain_parts = 0, const BlendTaps* bt = nullptr, bool feat_late = false, bool fa_f16 = false) {
  const int W = x.c->W, S = x.c->S, C = x.c->C;
  const int64_t N = R * S;
  if (!have_sigma) NL_TRY(nl_launch_sigma(geo, N, W, x.p<float>(x.L.sig_w), x.p<float>(x.L.sig_b), h.sigma, x.st));
  const bool want_feat = out->feat != nullptr;
  const bool term = term_eps > 0.f && !pre_done;
  if (term) NL_TRY(nl_launch_termination(z, h.sigma, R, S, term_eps, h.n_alive, h.tile_list, h.tile_count, x.st));
  bool feat_done = false;
  if (feat_late) {
    // f16mx, W = 256, FA = the chain kernel's fragment image: feat_mlp.0's hidden rows are never materialised — the compositing pass leaves the samples' weights
    // (in the caller's `weights` output, or in the buffer the hidden rows would have taken) and feat_comp_mx_kernel multiplies, activates, weights and sums in one go
    if (!want_feat || term || pre_done || (chain_parts & 1)) return NL_ERR_BAD_ARG;
    NL_TRY(do_heads_pre(x, V, FA, bl1, rgbv, N, false, h, 6 & ~chain_parts, false, bt));
    float* wts = out->weights ? out->weights + ray0 * S : h.fth;
    NL_TRY(nl_launch_composite(z, h.sigma, h.rgb_s, nullptr, valid_s, R, S, W, white, out, ray0, nullptr, h.wsum, x.st, nullptr, out->weights ? nullptr : h.fth));
    // ... and applies feat_mlp.2 to the rows it has summed (the per-ray GEMM below: 33 us whatever the batch, 3 % of a 512-ray shard's step)
    const bool f2 = x.L.g[G_FEAT2].Npad <= 192 && !dbg_switch("NERFLOC_NO_FEAT2_FUSED");
    NL_TRY(nl_launch_feat_comp_mx(FA, wts, N, S, x.pk + x.L.bsh[G_FEAT0P], x.pk + x.L.mx_feat0, x.p<float>(x.L.bias[G_FEAT0P]), h.hc, x.st,
                                  f2 ? x.p<float>(x.L.b32[G_FEAT2]) : nullptr, x.L.g[G_FEAT2].Npad, C, h.wsum, out->feat + ray0 * C, fa_f16));
    feat_done = f2;
  } else {
  if (!pre_done) NL_TRY(do_heads_pre(x, V, FA, bl1, rgbv, N, want_feat, h, 7 & ~chain_parts, term, bt));   // chain_parts: what the chain kernel already produced
  NL_TRY(nl_launch_composite(z, h.sigma, h.rgb_s, want_feat ? h.fth : nullptr, valid_s, R, S, W, white, out, ray0,
                             want_feat ? h.hc : nullptr, want_feat ? h.wsum : nullptr, x.st, term ? h.n_alive : nullptr));
  }
  if (want_feat && !feat_done) {   // feat = W2 . (sum_s w_s hidden_s) + b2 * sum_s w_s  ==  sum_s w_s (W2 . hidden_s + b2)
    SegSpec s1[2] = {{h.hc, W, W, 0, 1}, {h.wsum, 1, 1, 0, 1}};
    NL_TRY(run_gemm(x, G_FEAT2, s1, 2, R, out->feat + ray0 * C, C, NL_ACT_NONE));
  }
  if (out->sigma) NL_CHECK_HIP(hipMemcpyAsync(out->sigma + ray0 * S, h.sigma, sizeof(float) * N, hipMemcpyDeviceToDevice, x.st));
  return NL_OK;
}

// ---- the whole ray path backwards in one call (nl_render_rays_backward) ----------------------------------------------------------------
// = the four stage backwards above + the heads + compositing, sharing what the separate autograd nodes each recompute: ONE pass of the visibility
// decoders forward and ONE backward for the aggregation's and the blend's uses of visibility / depth difference, one geometry kernel for
// both sets of taps, one neighbour search.
struct RbBufs {
  MvBwdBufs m; PtBwdBufs p; UnBwdBufs q;
  float *xyz, *zc, *FA, *sigma, *Hf, *rgb_s, *hc, *wsum4, *ghc, *gw, *g_sigma, *g_rgb_s, *gFA, *gtmp, *gpre4, *gxyz_m, *gxyz_p, *gdir, *gG, *gqcN, *wts, *bv, *gpre4b;
};
void carve_rb(Bump& b, const nl_config* c, int V, int64_t R, RbBufs& a, bool train) {
  const int W = c->W, S = c->S;
  const size_t N = (size_t)R * S;
  // multi-view buffers: the union of the aggregation's and the blend's sets
  carve_mvb(b, c, V, (int64_t)N, true, a.m, train);
  a.m.t64 = b.take<float>(N * 64); a.m.G = b.take<float>(N * W); a.m.gA = b.take<float>(N * W);
  a.m.gt64 = b.take<float>(N * 64); a.m.gg393 = b.take<float>(N * ldg_of(c->C));
  carve_ptb(b, c, (int64_t)N, 8, a.p, train);
  carve_unb(b, c, R, a.q, train);
  a.xyz = b.take<float>(N * 3); a.zc = b.take<float>(N); a.FA = b.take<float>(N * W); a.sigma = b.take<float>(N); a.Hf = b.take<float>(N * W);
  a.rgb_s = b.take<float>(N * 3); a.hc = b.take<float>((size_t)R * W); a.wsum4 = b.take<float>((size_t)R * 4); a.ghc = b.take<float>((size_t)R * W);
  a.gw = b.take<float>(N); a.g_sigma = b.take<float>(N); a.g_rgb_s = b.take<float>(N * 3); a.gFA = b.take<float>(N * W); a.gtmp = b.take<float>(N * W);
  a.gpre4 = b.take<float>(N * 4); a.gxyz_m = b.take<float>(N * 3); a.gxyz_p = b.take<float>(N * 3); a.gdir = b.take<float>(N * 3);
  a.gG = b.take<float>(N * W); a.gqcN = b.take<float>(N * 3);
  a.wts = b.take<float>(N); a.bv = b.take<float>(N); a.gpre4b = b.take<float>(N * 4);   // the uncertainty head (keep / kept pair)
}
struct RbCot { const float *g_rgb, *g_depth, *g_unc, *g_feat, *g_wts; const int* idx; const float* d2; };
// the staged forward of the whole path into the workspace (everything the way back reads).  want_feat: feat_mlp.0's hidden rows too
int render_forward_staged(const Ctx& x32, const nl_frame* f, const float* qc, const float* qrows, const float* rays_o, const float* rays_d, const float* z, int64_t R,
                          bool want_feat, const int* knn_idx, const float* knn_d2, const RbBufs& a, bool frozen = false) {
  const int W = x32.c->W, S = x32.c->S;
  const int64_t N = R * S;
  hipStream_t st = x32.st;
  const NlViews vw = with_query(f, qc, qrows, S);
  const float eps_ln = 1e-6f;
  NL_TRY(nl_launch_sample_points(rays_o, rays_d, R, S, f->views.near_, f->views.far_, z, a.zc, a.xyz, st));
  NL_TRY(mv_recompute(x32, f, vw, a.xyz, N, a.m));                       // visibility / depth difference, statistics rows, the blend's per-view part
  NL_TRY(mv_outfc_forward(x32, f, N, a.m));                              // -> G
  if (frozen && pt_keep_fused_ok(x32, f, N, 8)) NL_TRY(pt_forward_keep_fused(x32, f, a.xyz, rays_d, 3, S, a.m.G, N, a.p, knn_idx, knn_d2));
  else NL_TRY(pt_forward_staged(x32, f, a.xyz, rays_d, 3, S, a.m.G, N, 8, a.p, knn_idx, knn_d2));
  NL_TRY(nl_launch_ln_agg(a.p.FCo, a.m.G, N, W, x32.p<float>(x32.L.ln_g), x32.p<float>(x32.L.ln_b), eps_ln, a.p.wscale, a.FA, st));
  NL_TRY(do_unet(x32, a.FA, R, a.q.geo, a.q.u));
  NL_TRY(nl_launch_sigma(a.q.geo, N, W, x32.p<float>(x32.L.sig_w), x32.p<float>(x32.L.sig_b), a.sigma, st));
  SegSpec sfa{a.FA, W, W, 0, 1};
  if (want_feat) NL_TRY(run_gemm(x32, G_FEAT0, &sfa, 1, N, a.Hf, W, NL_ACT_LRELU));
  NL_TRY(run_gemm(x32, G_BLENDA, &sfa, 1, N, a.m.blA, 32, NL_ACT_NONE));
  return nl_launch_blend(a.m.blA, a.m.bl1, a.m.rgbv, N, vw.V, x32.p<float>(x32.L.bl2_w), x32.p<float>(x32.L.bl2_b), x32.p<float>(x32.L.bl4_w),
                         x32.p<float>(x32.L.bl4_b), a.rgb_s, st);
}
// the way back from the staged forward's workspace
int render_backward_staged(const Ctx& xb, const Ctx& x32, const nl_frame* f, const float* qc, const float* qrows, const float* rays_d, int64_t R, int white,
                           const RbCot& ct, float* g_o, float* g_d, float* g_qc_rows, const RbBufs& a, const TrainOut* tg, const nl_beta_head* bh = nullptr) {
  const int W = x32.c->W, S = x32.c->S, C = x32.c->C;
  const int64_t N = R * S;
  hipStream_t st = x32.st;
  const NlViews vw = with_query(f, qc, qrows, S);
  const bool want_feat = ct.g_feat != nullptr;
  // ---------------------------------------------------------------- compositing backwards (feat = W2 . sum_s w_s hidden_s + b2 sum_s w_s)
  const float* b2 = x32.p<float>(x32.L.b32[G_FEAT2]) + (size_t)W * x32.L.g[G_FEAT2].Npad;   // the bias row of G_FEAT2's fp32 weights (K row W)
  if (want_feat) {
    SegSpec sgf{ct.g_feat, C, C, 0, 1};
    NL_TRY(run_gemm(xb, G_FEAT2_T, &sgf, 1, R, a.ghc, W, NL_ACT_NONE));
  }
  const bool beta = bh && bh->g_beta;
  // (the uncertainty head's share of the weights' cotangent has to be in before compositing is differentiated; its share of d/d geo joins the density
  // head's below)
  NL_TRY(nl_launch_gw_total(ct.g_wts, ct.g_feat, b2, R, S, C, a.gw, beta ? bh->g_beta : nullptr, a.bv, st));
  NL_TRY(nl_composite_backward(a.zc, a.sigma, a.rgb_s, want_feat ? a.Hf : nullptr, R, S, want_feat ? W : 0, white, ct.g_rgb, ct.g_depth, ct.g_unc,
                               want_feat ? a.ghc : nullptr, a.gw, a.g_sigma, a.g_rgb_s, want_feat ? a.gtmp : nullptr, st));
  // ---------------------------------------------------------------- heads
  bool have_gfa = false;
  if (want_feat) {   // feat_mlp: gtmp = d/d hidden -> LeakyReLU mask -> feat_mlp.0^T
    NL_TRY(nl_launch_lrelu_mask(a.gtmp, a.Hf, (size_t)N * W, st));
    NL_TRY(wgrad_to(tg, st, T_F0W, T_F0B, a.gtmp, W, W, a.FA, W, W, N));
    SegSpec sg{a.gtmp, W, W, 0, 1};
    NL_TRY(run_gemm(xb, G_FEAT0_T, &sg, 1, N, a.gFA, W, NL_ACT_NONE));
    have_gfa = true;
    if (tg && (tg->w[T_F2W] || tg->w[T_F2B])) {
      NL_TRY(nl_launch_ray_feat_sum(a.zc, a.sigma, a.Hf, R, S, W, a.hc, a.wsum4, st));
      if (tg->w[T_F2W]) NL_TRY(nl_launch_wgrad(ct.g_feat, C, C, a.hc, W, W, R, 0, 0, tg->w[T_F2W], W, 1, 0, nullptr, tg->scratch, tg->scratch_floats, st));
      if (tg->w[T_F2B]) NL_TRY(nl_launch_wgrad(ct.g_feat, C, C, a.wsum4, 4, 1, R, 0, 0, tg->w[T_F2B], 1, 1, 0, nullptr, tg->scratch, tg->scratch_floats, st));
    }
  }
  // density head -> g_geo (in q.gout's neighbour: reuse a.gG as scratch is not possible yet; g_geo lives in a.Hf, free from here on)
  float* g_geo = a.Hf;
  NL_TRY(nl_launch_sigma_backward(a.q.geo, N, W, x32.p<float>(x32.L.sig_w), x32.p<float>(x32.L.sig_b), a.g_sigma, g_geo, a.gpre4, st));
  NL_TRY(wgrad_to(tg, st, T_SIGW, T_SIGB, a.gpre4, 4, 1, a.q.geo, W, W, N));
  if (beta) {
    NL_TRY(nl_launch_beta_backward(a.q.geo, N, S, W, bh->weight, bh->bias, a.wts, bh->g_beta, g_geo, a.gpre4b, st));
    if (bh->g_weight) NL_TRY(nl_launch_wgrad(a.gpre4b, 4, 1, a.q.geo, W, W, N, 0, 0, bh->g_weight, W, 1, 0, bh->g_bias, tg ? tg->scratch : nullptr,
                                            tg ? tg->scratch_floats : 0, st));
  }
  // ---------------------------------------------------------------- ray U-Net, colour blend: their shares of d/d feature_agg
  NL_TRY(unet_backward_only(xb, x32, a.FA, R, g_geo, a.gtmp, a.q, tg));
  if (have_gfa) NL_TRY(nl_launch_add(a.gFA, a.gtmp, a.gFA, (size_t)N * W, st));
  else NL_CHECK_HIP(hipMemcpyAsync(a.gFA, a.gtmp, sizeof(float) * (size_t)N * W, hipMemcpyDeviceToDevice, st));
  NL_TRY(blend_tail_backward(xb, x32, f, vw, a.xyz, a.FA, N, a.g_rgb_s, a.gtmp, a.m, tg));
  NL_TRY(nl_launch_add(a.gFA, a.gtmp, a.gFA, (size_t)N * W, st));
  // ---------------------------------------------------------------- neural-point branch, aggregation, geometry + decoders
  NL_TRY(pt_backward_only(xb, x32, f, a.xyz, rays_d, 3, S, a.m.G, N, 8, a.gFA, a.gxyz_p, a.gdir, a.gG, a.p, ct.idx, ct.d2, tg));
  NL_TRY(mv_outfc_backward(xb, x32, f, N, a.gG, a.m, tg));
  NL_TRY(mv_geom_dec_backward(x32, f, vw, a.xyz, N, a.m.gg393, true, a.gxyz_m, g_qc_rows ? a.gqcN : nullptr, a.m, tg));
  return nl_launch_ray_reduce(a.gxyz_m, a.gxyz_p, nullptr, a.gdir, g_qc_rows ? a.gqcN : nullptr, a.zc, R, S, g_o, g_d, g_qc_rows, st);
}
int do_render_backward(const Ctx& xb, const Ctx& x32, const nl_frame* f, const float* qc, const float* qrows, const float* rays_o, const float* rays_d, const float* z,
                       int64_t R, int white, const RbCot& ct, float* g_o, float* g_d, float* g_qc_rows, const RbBufs& a, const TrainOut* tg) {
  NL_TRY(render_forward_staged(x32, f, qc, qrows, rays_o, rays_d, z, R, ct.g_feat != nullptr, ct.idx, ct.d2, a, tg == nullptr));
  return render_backward_staged(xb, x32, f, qc, qrows, rays_d, R, white, ct, g_o, g_d, g_qc_rows, a, tg);
}
// the per-ray outputs from the staged forward's workspace (the gradient path's forward values: split-FP16 arithmetic in the bf16 modes)
int render_outputs_staged(const Ctx& x32, const nl_frame* f, int64_t R, int white, const nl_render_out* out, const RbBufs& a, const nl_beta_head* bh = nullptr) {
  const int W = x32.c->W, S = x32.c->S, C = x32.c->C;
  const bool want_feat = out->feat != nullptr;
  NL_TRY(nl_launch_composite(a.zc, a.sigma, a.rgb_s, want_feat ? a.Hf : nullptr, a.m.valid_s, R, S, W, white, out, 0, want_feat ? a.hc : nullptr,
                             want_feat ? a.gw : nullptr, x32.st));   // (a.gw: R floats of scratch for the weight sums; the way back rewrites it)
  if (want_feat) {
    SegSpec s1[2] = {{a.hc, W, W, 0, 1}, {a.gw, 1, 1, 0, 1}};
    NL_TRY(run_gemm(x32, G_FEAT2, s1, 2, R, out->feat, C, NL_ACT_NONE));
  }
  if (bh) {   // the uncertainty head: softplus(beta_mlp.0(geo)) per sample (the density head's kernel), composited with the weights; both stay for the way back
    NL_TRY(nl_launch_sigma(a.q.geo, R * S, W, bh->weight, bh->bias, a.bv, x32.st));
    NL_CHECK_HIP(hipMemcpyAsync(a.wts, out->weights, sizeof(float) * (size_t)R * S, hipMemcpyDeviceToDevice, x32.st));
    NL_TRY(nl_launch_beta_forward(a.wts, a.bv, R, S, bh->beta_min, bh->beta, x32.st));
  }
  return NL_OK;
}

__global__ void gap_check_kernel(const unsigned char* __restrict__ p, size_t n, unsigned char pat, int* __restrict__ bad) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n && p[i] != pat) atomicAdd(bad, 1);
}

Ctx make_ctx(const nl_config* c, const void* packed, void* stream) {
  Ctx x;
  x.c = c; x.L = make_layout(c); x.pk = (const char*)packed; x.st = (hipStream_t)stream;
  const PackInfo pi = pack_info(packed);
  x.has_bst = pi.bst; x.has_bsh = pi.bsh;
  return x;
}

}  // namespace

// =====================================================================================================
extern "C" {

static_assert(sizeof(nl_render_opts) == 32, "nl_render_opts is part of the C-ABI: 32 bytes");
static_assert(sizeof(nl_train_grads) == 72 && sizeof(nl_render_cotangents) == 64, "training / cotangent blocks are part of the C-ABI");
static_assert(offsetof(nl_render_opts, flags) == 4 && offsetof(nl_render_opts, ray_centers) == 8, "nl_render_opts layout");
int nl_abi_version(void) { return NL_ABI_VERSION; }

int nl_debug_bump_gap(size_t bytes) { g_bump_gap = nl_align_up(bytes, 256); g_bump_gaps.clear(); return NL_OK; }
// counts the recorded gap regions (since the last nl_debug_bump_gap / nl_debug_check_gaps) that hold anything but `pattern`; scratch: 4 device bytes
int nl_debug_check_gaps(int pattern, int* scratch, int* bad_regions, int* checked_regions, void* stream) {
  if (!scratch || !bad_regions) return NL_ERR_BAD_ARG;
  if (checked_regions) *checked_regions = (int)g_bump_gaps.size();
  hipStream_t st = (hipStream_t)stream;
  int bad = 0;
  for (const auto& g : g_bump_gaps) {
    NL_CHECK_HIP(hipMemsetAsync(scratch, 0, sizeof(int), st));
    hipLaunchKernelGGL(gap_check_kernel, dim3((unsigned)nl_cdiv((int64_t)g.second, 256)), dim3(256), 0, st, (const unsigned char*)g.first, g.second, (unsigned char)pattern,
                       scratch);
    int h = 0;
    NL_CHECK_HIP(hipMemcpyAsync(&h, scratch, sizeof(int), hipMemcpyDeviceToHost, st));
    NL_CHECK_HIP(hipStreamSynchronize(st));
    if (h) ++bad;
  }
  *bad_regions = bad;
  g_bump_gaps.clear();
  return NL_OK;
}

int nl_profile_begin(void) { g_prof.on = true; g_prof.used = 0; return NL_OK; }
int nl_profile_end(float* fused_ms, int* launches) {
  float tot = 0.f;
  for (int i = 0; i + 1 < g_prof.used; i += 2) {
    float ms = 0.f;
    if (hipEventSynchronize(g_prof.ev[i + 1]) != hipSuccess || hipEventElapsedTime(&ms, g_prof.ev[i], g_prof.ev[i + 1]) != hipSuccess) return NL_ERR_HIP;
    tot += ms;
  }
  if (fused_ms) *fused_ms = tot;
  if (launches) *launches = g_prof.used / 2;
  g_prof.on = false; g_prof.used = 0;
  return NL_OK;
}

const char* nl_strerror(int s) {
  switch (s) {
    case NL_OK: return "ok";
    case NL_ERR_BAD_ARG: return "bad argument";
    case NL_ERR_UNSUPPORTED: return "unsupported shape or option";
    case NL_ERR_WORKSPACE: return "workspace too small";
    case NL_ERR_HIP: return "HIP runtime error";
    case NL_ERR_NO_DEVICE: return "no HIP device";
    default: return "unknown status";
  }
}

int nl_num_weights(void) { return kNumWeights; }
const char* nl_weight_name(int i) { return (i >= 0 && i < kNumWeights) ? kWeightNames[i] : nullptr; }

size_t nl_packed_weights_bytes(const nl_config* cfg) {
  NL_EFF_CFG(cfg); return cfg_ok(cfg) ? make_layout(cfg).total : 0; }

int nl_pack_weights(const nl_config* cfg, const float* const* t, int n, void* packed, size_t bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg) || !t || n != kNumWeights || !packed) return NL_ERR_BAD_ARG;
  for (int i = 0; i < n; ++i) if (!t[i]) return NL_ERR_BAD_ARG;
  const Layout L = make_layout(cfg);
  if (bytes < L.total) return NL_ERR_WORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  bump_generation(packed);
  NL_CHECK_HIP(hipMemsetAsync(packed, 0, L.total, st));
  Packer P{t, (char*)packed, &L, st};
  const int W = cfg->W, C = cfg->C, F = C + 3;
  P.linear(G_OUTFC0, t[T_OUT0W], t[T_OUT0B]);
  P.linear(G_OUTFC2, t[T_OUT2W], t[T_OUT2B]);
  P.linear(G_BASE0, t[T_B0W], t[T_B0B]);
  P.linear(G_BASE2, t[T_B2W], t[T_B2B]);
  P.linear(G_BASE4, t[T_B4W], t[T_B4B]);
  // KV: columns 0..127 = w_ks rows, 128..255 = w_vs rows
  {
    const GemmDim& d = L.g[G_KV];
    P.mark(G_KV, true);
    for (int half = 0; half < 2; ++half) {
      const float* w = t[half ? T_WV : T_WK];
      int nel = W * 128;
      // reuse pack_block with N=128 into a column window: emulate by offsetting destination pointers
      hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(nel, 256)), dim3(256), 0, st, w, 0, W, 1, W, 128, 0,
                         (float*)((char*)packed + L.b32[G_KV]) + half * 128,
                         (unsigned short*)((char*)packed + L.bhi[G_KV]) + (size_t)half * 128 * d.Kpad,
                         (unsigned short*)((char*)packed + L.blo[G_KV]) + (size_t)half * 128 * d.Kpad, d.Kpad, d.Npad,
                         (unsigned short*)((char*)packed + L.bst[G_KV]), nl_tgemm_nrt(d.N), half * 128, 0, (unsigned short*)((char*)packed + L.bsh[G_KV]));
    }
  }
  P.linear(G_Q, t[T_WQ], nullptr);
  P.linear(G_FC, t[T_FC], nullptr);
  // transposed copies (element [k = output o][n = input i] = w[o][i]): source strides swapped
  P.block(G_FC_T, 0, t[T_FC], 0, 1, 128, W);
  P.block(G_Q_T, 0, t[T_WQ], 0, 1, W, 128);
  {
    const GemmDim& d = L.g[G_KV_T];   // K = [k-projection outputs 128 | v-projection outputs 128]
    P.mark(G_KV_T, false);
    for (int half = 0; half < 2; ++half)
      hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(128 * W, 256)), dim3(256), 0, st, t[half ? T_WV : T_WK], 0, 1, W, 128, W, half * 128,
                         (float*)((char*)packed + L.b32[G_KV_T]), (unsigned short*)((char*)packed + L.bhi[G_KV_T]),
                         (unsigned short*)((char*)packed + L.blo[G_KV_T]), d.Kpad, d.Npad, (unsigned short*)((char*)packed + L.bst[G_KV_T]), nl_tgemm_nrt(d.N), 0);
  }
  P.block(G_BASE4_T, 0, t[T_B4W], 0, 1, W, W);
  P.block(G_BASE2_T, 0, t[T_B2W], 0, 1, W, W);
  {
    const GemmDim& d = L.g[G_BASE0_T];   // columns F .. F+89 of base_mlp.0.weight (W, F + 90); the 6 pad columns stay zero
    P.mark(G_BASE0_T, false);
    hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(W * 90, 256)), dim3(256), 0, st, t[T_B0W], F, 1, F + 90, W, 90, 0,
                       (float*)((char*)packed + L.b32[G_BASE0_T]), (unsigned short*)((char*)packed + L.bhi[G_BASE0_T]),
                       (unsigned short*)((char*)packed + L.blo[G_BASE0_T]), d.Kpad, d.Npad, (unsigned short*)((char*)packed + L.bst[G_BASE0_T]), nl_tgemm_nrt(d.N), 0);
  }
  {
    const GemmDim& d = L.g[G_BASE0_TF];   // columns 0 .. F-1 of base_mlp.0.weight
    P.mark(G_BASE0_TF, false);
    hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(W * F, 256)), dim3(256), 0, st, t[T_B0W], 0, 1, F + 90, W, F, 0,
                       (float*)((char*)packed + L.b32[G_BASE0_TF]), (unsigned short*)((char*)packed + L.bhi[G_BASE0_TF]),
                       (unsigned short*)((char*)packed + L.blo[G_BASE0_TF]), d.Kpad, d.Npad, (unsigned short*)((char*)packed + L.bst[G_BASE0_TF]), nl_tgemm_nrt(d.N), 0);
  }
  P.block(G_BASE0_S, 0, t[T_B0W], F, F + 90, 1, 90);
  P.block(G_OUTFC2_T, 0, t[T_OUT2W], 0, 1, 64, W);
  P.block(G_FEAT0_T, 0, t[T_F0W], 0, 1, W, W);
  P.block(G_FEAT2_T, 0, t[T_F2W], 0, 1, W, C);
  {
    // out_fc.0.weight (64, 2F + 3): element [k = o][n = i].  416 columns at C = 192: generic kernels, no streaming layout; a narrower feature map (C <= 123) puts
    // the product on the streaming kernel, whose weight stream must then exist (it was left zero-filled until tools/grad_fuzz.py: every gradient through the
    // statistics rows vanished for such C in the non-fp32 modes)
    const GemmDim& d = L.g[G_OUTFC0_T];
    const bool stream = d.N <= 256;
    if (stream) P.mark(G_OUTFC0_T, true);
    hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(64 * (2 * F + 3), 256)), dim3(256), 0, st, t[T_OUT0W], 0, 1, 2 * F + 3, 64, 2 * F + 3, 0,
                       (float*)((char*)packed + L.b32[G_OUTFC0_T]), (unsigned short*)((char*)packed + L.bhi[G_OUTFC0_T]),
                       (unsigned short*)((char*)packed + L.blo[G_OUTFC0_T]), d.Kpad, d.Npad,
                       stream ? (unsigned short*)((char*)packed + L.bst[G_OUTFC0_T]) : (unsigned short*)nullptr, nl_tgemm_nrt(d.N), 0, 0,
                       stream ? (unsigned short*)((char*)packed + L.bsh[G_OUTFC0_T]) : (unsigned short*)nullptr);
  }
  {
    const GemmDim& d = L.g[G_BLENDA_T];   // the feature_agg columns of rgb_blending_mlp.0.weight (32, W + F + 5)
    P.mark(G_BLENDA_T, false);
    hipLaunchKernelGGL(pack_block_kernel, dim3((unsigned)nl_cdiv(32 * W, 256)), dim3(256), 0, st, t[T_BL0W], 0, 1, W + F + 5, 32, W, 0,
                       (float*)((char*)packed + L.b32[G_BLENDA_T]), (unsigned short*)((char*)packed + L.bhi[G_BLENDA_T]),
                       (unsigned short*)((char*)packed + L.blo[G_BLENDA_T]), d.Kpad, d.Npad, (unsigned short*)((char*)packed + L.bst[G_BLENDA_T]), nl_tgemm_nrt(d.N), 0);
  }
  const float* const* un = t + T_UNET;
  { const int w1[1] = {W}, w2[1] = {64}, w3[1] = {128};
    P.conv3(G_CONV1, un[0], un[1], W, w1, 1);
    P.conv3(G_CONV1F, un[0], un[1], W, w1, 1, 1u);
    P.conv3(G_CONV2, un[4], un[5], 64, w2, 1);
    P.conv3(G_CONV3, un[8], un[9], 128, w3, 1); }
  P.convT(G_T3E, G_T3O, un[12], un[13], 128, 128);
  P.convT(G_T2E, G_T2O, un[16], un[17], 256, 64);
  P.convT(G_T1E, G_T1O, un[20], un[21], 128, 32);
  P.convT_merged(G_T3M, un[12], un[13], 128, 128);
  P.convT_merged(G_T2M, un[16], un[17], 256, 64);
  P.convT_merged(G_T1M, un[20], un[21], 128, 32);
  { const int wo[2] = {W, 32}; P.conv3(G_CONVOUT, un[24], un[25], W + 32, wo, 2); P.conv3(G_CONVOUTF, un[24], un[25], W + 32, wo, 2, 1u); }
  if (W == 256) {   // NL_PREC_F16MX: conv_out's fp6 images for tgemm_mx_kernel, from the layer's packed fp32 matrix (same K order as its streams)
    const GemmDim& d = L.g[G_CONVOUTF];
    const int nslab = (d.Kpad / 32 + 1) / 2;
    hipLaunchKernelGGL(pack_tgemm_mx6_kernel, dim3((unsigned)nl_cdiv((int64_t)nslab * 1024, 256)), dim3(256), 0, st, (const float*)((char*)packed + L.b32[G_CONVOUTF]), d.Kpad, d.Npad,
                       d.N, nslab, (unsigned char*)packed + L.mx_convout);
  }
  P.conv3_dgrad(G_UB_OUTA, un[24], W, W + 32, 0, W);
  P.conv3_dgrad(G_UB_OUTB, un[24], W, W + 32, W, 32);
  P.convT_dgrad(G_UB_T1, un[20], 128, 32);
  P.convT_dgrad(G_UB_T2, un[16], 256, 64);
  P.convT_dgrad(G_UB_T3, un[12], 128, 128);
  P.conv3_dgrad(G_UB_C3, un[8], 128, 128, 0, 128);
  P.conv3_dgrad(G_UB_C2, un[4], 128, 64, 0, 64);
  P.conv3_dgrad(G_UB_C1, un[0], 64, W, 0, W);
  for (int u = 0; u < U_COUNT; ++u) {
    P.transpose(un[4 * u + 2], L.un_g[u], L.un_c[u], L.un_l[u]);   // (C, L) -> (L, C)
    P.transpose(un[4 * u + 3], L.un_b[u], L.un_c[u], L.un_l[u]);
    P.lane_major(L.un_g[u], L.un_gl[u], L.un_n[u], L.un_so[u]);
    P.lane_major(L.un_b[u], L.un_bl[u], L.un_n[u], L.un_so[u]);
  }
  P.linear(G_FEAT0, t[T_F0W], t[T_F0B]);
  P.block(G_FEAT2, 0, t[T_F2W], 0, W, 1, W);
  P.block(G_FEAT2, W, t[T_F2B], 0, 1, 0, 1);   // bias as the K-row that meets the weight-sum column
  P.block(G_BLENDA, 0, t[T_BL0W], 0, W + F + 5, 1, W);
  if (W % 32 == 0) {   // accumulator-order copies for the chain kernel
    P.block(G_FEAT0P, 0, t[T_F0W], 0, W, 1, W, 1);
    P.copy(t[T_F0B], L.bias[G_FEAT0P], W);
    if (W == 256) {   // NL_PREC_F16MX: feat_mlp.0's fp6 images for feat_comp_mx_kernel (K in accumulator order = the order of feature_agg's fragment image)
      const GemmDim& d = L.g[G_FEAT0P];
      const int nslab = (d.Kpad / 32 + 1) / 2;
      hipLaunchKernelGGL(pack_tgemm_mx6_kernel, dim3((unsigned)nl_cdiv((int64_t)nslab * 1024, 256)), dim3(256), 0, st, (const float*)((char*)packed + L.b32[G_FEAT0P]), d.Kpad, d.Npad,
                         d.N, nslab, (unsigned char*)packed + L.mx_feat0);
    }
    P.block(G_BLENDAP, 0, t[T_BL0W], 0, W + F + 5, 1, W, 1);
    P.block(G_QP, 0, t[T_WQ], 0, W, 1, W, 1);
  }
  P.block(G_BLENDP, 0, t[T_BL0W], W + 3, W + F + 5, 1, C);
  if (nl_pack_ptt(t[T_B0W], t[T_B0B], W, F, L.g[G_PTT].Kpad, L.g[G_PTT].Npad, (float*)((char*)packed + L.b32[G_PTT]),
                  (float*)((char*)packed + L.bias[G_PTT]), st) != NL_OK) return NL_ERR_HIP;
  hipLaunchKernelGGL(pack_blw_kernel, dim3(2), dim3(256), 0, st, t[T_BL0W], t[T_BL0B], (float*)((char*)packed + L.blw), W, F);
  // small VALU-side weights
  P.copy(t[T_RD0W], L.rd_w, 64); P.copy(t[T_RD0B], L.rd_w + 4 * 64, 16);
  P.copy(t[T_RD2W], L.rd_w + 4 * 80, 27 * 16); P.copy(t[T_RD2B], L.rd_w + 4 * (80 + 432), 27);
  for (int d = 0; d < 4; ++d) {
    const float* const* q = t + T_DEC + 6 * d;
    const size_t o = L.dec_w + 4 * (size_t)d * 2178;
    const int nout = d < 2 ? 2 : 1;
    P.copy(q[0], o, 1024); P.copy(q[1], o + 4 * 1024, 32);
    P.copy(q[2], o + 4 * 1056, 1024); P.copy(q[3], o + 4 * 2080, 32);
    P.copy(q[4], o + 4 * 2112, 32 * nout); P.copy(q[5], o + 4 * 2176, nout);
  }
  if (nl_pack_mv_decoder((const float*)((char*)packed + L.dec_w), (char*)packed + L.dec_mfma, st) != NL_OK) return NL_ERR_HIP;
  P.copy(t[T_SIGW], L.sig_w, W); P.copy(t[T_SIGB], L.sig_b, 1);
  P.copy(t[T_BL2W], L.bl2_w, 512); P.copy(t[T_BL2B], L.bl2_b, 16);
  P.copy(t[T_BL4W], L.bl4_w, 16); P.copy(t[T_BL4B], L.bl4_b, 1);
  P.copy(t[T_LNW], L.ln_g, W); P.copy(t[T_LNB], L.ln_b, W);
  P.copy(t[T_B0B], L.pt_bias, W); P.copy(t[T_B2B], L.pt_bias + 4 * (size_t)W, W); P.copy(t[T_B4B], L.pt_bias + 8 * (size_t)W, W);
  if (cfg->C == 192) {
    int rc = nl_pack_mv_front(t[T_OUT0W], t[T_OUT0B], (char*)packed + L.mvf_pack, st);
    if (rc != NL_OK) return rc;
  }
  if (W == 64 || W == 128 || W == 256) {
    int rc = nl_pack_point_stream(t[T_B0W], t[T_B2W], t[T_B4W], t[T_WK], t[T_WV], (char*)packed + L.pt_stream, W, F, st);
    if (rc != NL_OK) return rc;
  }
  if (W == 128 || W == 256) {   // rd_w was filled by the copies above (same stream)
    int rc = nl_pack_point_stream2(t[T_B0W], t[T_B2W], t[T_B4W], t[T_WK], t[T_WV], t[T_B2B], t[T_B4B], (const float*)((char*)packed + L.rd_w),
                                   (char*)packed + L.pt_stream2, W, F, st);
    if (rc != NL_OK) return rc;
    rc = nl_pack_point_stream2(t[T_B0W], t[T_B2W], t[T_B4W], t[T_WK], t[T_WV], t[T_B2B], t[T_B4B], (const float*)((char*)packed + L.rd_w),
                               (char*)packed + L.pt_stream2_mx, W, F, st, 1, (int*)((char*)packed + L.pt_mx_sc));
    if (rc != NL_OK) return rc;
    rc = nl_pack_point_stream2(t[T_B0W], t[T_B2W], t[T_B4W], t[T_WK], t[T_WV], t[T_B2B], t[T_B4B], (const float*)((char*)packed + L.rd_w),
                               (char*)packed + L.pt_stream2_f16, W, F, st, 2, nullptr);
    if (rc != NL_OK) return rc;
    rc = nl_pack_point_bwd_stream(t[T_B0W], t[T_B2W], t[T_B4W], t[T_WK], t[T_WV], (char*)packed + L.pt_bwd_stream, W, F, st);
    if (rc != NL_OK) return rc;
  }
  NL_LAUNCH_CHECK();
  static_assert(G_COUNT <= 64, "one bit per layer");
  set_pack_streams(packed, P.has_bst, P.has_bsh);
  return NL_OK;
}

// ---- frame ---------------------------------------------------------------------------------------------
static bool desc_ok(const nl_config* c, const nl_frame_desc* d) {
  return cfg_ok(c) && d && d->V >= 1 && d->V <= NL_MAX_VIEWS && d->H > 1 && d->Wimg > 1 && d->h > 1 && d->w > 1 && d->vis_h > 1 && d->vis_w > 1 && d->M >= 0 &&
         d->images && d->featmaps && d->vis_featmaps && d->proj_ibr && d->proj_neuray && d->cam_centers &&
         (d->M == 0 || (d->sp_xyz && d->sp_feature && d->sp_confidence && d->sp_direction));
}

size_t nl_frame_bytes(const nl_config* cfg, const nl_frame_desc* d) {
  NL_EFF_CFG(cfg);
  if (!desc_ok(cfg, d)) return 0;
  return nl_align_up((size_t)d->V * d->vis_h * d->vis_w * 32 * 4, 256) + nl_knn_grid_bytes(d->M) + nl_align_up((size_t)(d->M + 1) * cfg->W * 4, 256) +
         nl_align_up((size_t)d->V * d->h * d->w * 32 * 4, 256) + 1024 + nl_align_up((size_t)d->M * cfg->W * 4, 256) +
         nl_align_up((size_t)d->M * nl_align_up(cfg->C + 3, 32) * 4, 256);
}

int nl_frame_create(const nl_config* cfg, const nl_frame_desc* d, void* mem, size_t bytes, void* stream, nl_frame** out) {
  NL_EFF_CFG(cfg);
  if (!desc_ok(cfg, d) || !mem || !out) return NL_ERR_BAD_ARG;
  if (bytes < nl_frame_bytes(cfg, d)) return NL_ERR_WORKSPACE;
  nl_frame* f = new (std::nothrow) nl_frame;
  if (!f) return NL_ERR_BAD_ARG;
  memset(&f->views, 0, sizeof(NlViews));
  f->views.V = d->V; f->views.H = d->H; f->views.Wimg = d->Wimg; f->views.h = d->h; f->views.w = d->w;
  f->views.vh = d->vis_h; f->views.vw = d->vis_w;
  f->views.near_ = d->near_; f->views.far_ = d->far_;
  for (int v = 0; v < d->V; ++v) {
    memcpy(f->views.P1[v], d->proj_ibr + 12 * v, 48);
    memcpy(f->views.P2[v], d->proj_neuray + 12 * v, 48);
    memcpy(f->views.cam[v], d->cam_centers + 3 * v, 12);
  }
  f->C = cfg->C;
  f->images = d->images; f->feat = d->featmaps;
  f->sp_xyz = d->sp_xyz; f->sp_feat = d->sp_feature; f->sp_conf = d->sp_confidence; f->sp_dir = d->sp_direction;
  f->M = d->M;
  hipStream_t st = (hipStream_t)stream;
  f->visf_hwc = (float*)mem;
  int rc = nl_launch_chw_to_hwc(d->vis_featmaps, f->visf_hwc, d->V, 32, d->vis_h * d->vis_w, st);
  if (rc == NL_OK) rc = nl_knn_grid_build(&f->grid, (char*)mem + nl_align_up((size_t)d->V * d->vis_h * d->vis_w * 32 * 4, 256), d->sp_xyz, d->M, st);
  {
    char* p = (char*)mem + nl_align_up((size_t)d->V * d->vis_h * d->vis_w * 32 * 4, 256) + nl_knn_grid_bytes(d->M);
    f->ptt = (float*)p;
    f->ptt_for = nullptr; f->ptt_gen = 0;
    f->pfeat = (float*)(p + nl_align_up((size_t)(d->M + 1) * cfg->W * 4, 256));
    f->pfeat_for = nullptr; f->pfeat_gen = 0;
    f->views_dev = (float*)((char*)f->pfeat + nl_align_up((size_t)d->V * d->h * d->w * 32 * 4, 256));
    f->tr_gT = (float*)((char*)f->views_dev + 1024);
    f->tr_tmp = (float*)((char*)f->tr_gT + nl_align_up((size_t)d->M * cfg->W * 4, 256));
    memset(f->views_host, 0, sizeof(f->views_host));
    for (int v = 0; v < d->V; ++v) {
      memcpy(f->views_host + 12 * v, d->proj_ibr + 12 * v, 48);
      memcpy(f->views_host + 192 + 3 * v, d->cam_centers + 3 * v, 12);
    }
    if (rc == NL_OK && hipMemcpyAsync(f->views_dev, f->views_host, sizeof(f->views_host), hipMemcpyHostToDevice, st) != hipSuccess) rc = NL_ERR_HIP;
    // the 16 floats of slack behind the matrices hold the frame's diagnostics (nl_frame_diagnostics): [248] max |T|, [249] max |attention logit|
    if (rc == NL_OK && hipMemsetAsync(f->views_dev + 240, 0, 64, st) != hipSuccess) rc = NL_ERR_HIP;
  }
  if (rc != NL_OK) { delete f; return rc; }
  // the side stream and its two events (see nl_frame): failure to create them only disables the fork
  f->side = nullptr; f->ev_fork = f->ev_join = nullptr;
  f->side_ok = hipStreamCreateWithFlags(&f->side, hipStreamNonBlocking) == hipSuccess &&
               hipEventCreateWithFlags(&f->ev_fork, hipEventDisableTiming) == hipSuccess &&
               hipEventCreateWithFlags(&f->ev_join, hipEventDisableTiming) == hipSuccess;
  if (!f->side_ok) (void)hipGetLastError();
  f->parts_ok = f->side_ok;
  for (int i = 0; i < nl_frame::kMaxParts; ++i) { f->ev_go[i] = f->ev_done[i] = nullptr; }
  for (int i = 0; i < nl_frame::kMaxParts && f->parts_ok; ++i)
    f->parts_ok = hipEventCreateWithFlags(&f->ev_go[i], hipEventDisableTiming) == hipSuccess && hipEventCreateWithFlags(&f->ev_done[i], hipEventDisableTiming) == hipSuccess;
  if (!f->parts_ok) (void)hipGetLastError();
  *out = f;
  return NL_OK;
}

int nl_frame_destroy(nl_frame* f) {
  if (!f) return NL_OK;
  if (f->side) { (void)hipStreamSynchronize(f->side); (void)hipStreamDestroy(f->side); }
  if (f->ev_fork) (void)hipEventDestroy(f->ev_fork);
  if (f->ev_join) (void)hipEventDestroy(f->ev_join);
  for (int i = 0; i < nl_frame::kMaxParts; ++i) { if (f->ev_go[i]) (void)hipEventDestroy(f->ev_go[i]); if (f->ev_done[i]) (void)hipEventDestroy(f->ev_done[i]); }
  delete f;
  return NL_OK;
}

int nl_frame_diagnostics(const nl_frame* f, float* host_out, int32_t n, void* stream) {
  if (!f || !host_out || n < 1) return NL_ERR_BAD_ARG;
  struct { float tmax, lmax; unsigned long long cyc, ref; } tmp = {0.f, 0.f, 0ull, 0ull};   // floats 248, 249; two 64-bit counters at floats 250 .. 253
  static_assert(sizeof(tmp) == 24, "diagnostics block");
  NL_CHECK_HIP(hipMemcpyAsync(&tmp, f->views_dev + 248, sizeof(tmp), hipMemcpyDeviceToHost, (hipStream_t)stream));
  NL_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
  const float vals[NL_DIAG_COUNT] = {tmp.tmax, tmp.lmax, tmp.ref ? (float)((double)tmp.cyc / ((double)tmp.ref * 10.0)) : 0.f,   // cycles per ns = GHz
                                     (float)f->guard_last_prec, (float)f->guard_escalations};
  for (int i = 0; i < n; ++i) host_out[i] = i < NL_DIAG_COUNT ? vals[i] : 0.f;
  return NL_OK;
}

// ---- stages ----------------------------------------------------------------------------------------------
int nl_knn(const nl_frame* f, const float* xyz, int64_t N, int K, int32_t* idx, float* d2, void* stream) {
  if (N == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!f || !xyz || !idx || !d2 || N < 0 || K < 1 || K > NL_KNN_MAX_K) return NL_ERR_BAD_ARG;
  return nl_knn_search(&f->grid, xyz, N, K, idx, d2, (hipStream_t)stream);
}

int nl_sample_points(const float* o, const float* d, int64_t R, int S, float near_, float far_, const float* z_in, float* z_out,
                     float* xyz, void* stream) {
  if (R == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!o || !d || !xyz || R < 0 || S < 1) return NL_ERR_BAD_ARG;
  return nl_launch_sample_points(o, d, R, S, near_, far_, z_in, z_out, xyz, (hipStream_t)stream);
}

size_t nl_mv_aggregate_workspace_bytes(const nl_config* cfg, int V, int64_t N) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  Bump b{nullptr, 0}; MvBufs m; carve_mv(b, cfg, V, N, m); return b.off;
}

int nl_mv_aggregate(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* xyz, int64_t N,
                    float* mv_feat, float* rgb_feat, float* vis_ang, int32_t* valid_s, float* blend1, float* rgbv, void* ws,
                    size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (N == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!cfg_ok(cfg) || !packed || !f || !xyz || !mv_feat || !valid_s || !ws || N < 0 || (blend1 && !rgbv)) return NL_ERR_BAD_ARG;
  if (ws_bytes < nl_mv_aggregate_workspace_bytes(cfg, f->views.V, N)) return NL_ERR_WORKSPACE;
  Bump b{(char*)ws, 0}; MvBufs m; carve_mv(b, cfg, f->views.V, N, m);
  Ctx x = make_ctx(cfg, packed, stream);
  return do_mv(x, f, qc, xyz, N, mv_feat, rgb_feat, vis_ang, valid_s, blend1, rgbv, m);
}

size_t nl_point_mlp_workspace_bytes(const nl_config* cfg, int64_t N) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  Bump b{nullptr, 0}; PtBufs p; carve_pt(b, cfg, N, 8, p, true); return b.off;
}

int nl_point_mlp(const nl_config* cfg, const void* packed, const nl_frame* f, const float* xyz, const float* dir, int64_t dir_stride,
                 const float* mv_feat, int64_t N, int K, float* feature_agg, int32_t* knn_idx, float* knn_d2, void* ws, size_t ws_bytes,
                 void* stream) {
  NL_EFF_CFG(cfg);
  if (N == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!cfg_ok(cfg) || !packed || !f || !xyz || !mv_feat || !feature_agg || !ws || N < 0 || K < 1 || K > 8) return NL_ERR_BAD_ARG;
  if (ws_bytes < nl_point_mlp_workspace_bytes(cfg, N)) return NL_ERR_WORKSPACE;
  Bump b{(char*)ws, 0}; PtBufs p; carve_pt(b, cfg, N, 8, p, true);
  Ctx x = make_ctx(cfg, packed, stream);
  NL_TRY(do_point(x, f, xyz, dir, (int)dir_stride, 1, mv_feat, N, K, feature_agg, p));
  if (knn_idx) NL_CHECK_HIP(hipMemcpyAsync(knn_idx, p.idx, sizeof(int) * N * K, hipMemcpyDeviceToDevice, x.st));
  if (knn_d2) NL_CHECK_HIP(hipMemcpyAsync(knn_d2, p.d2, sizeof(float) * N * K, hipMemcpyDeviceToDevice, x.st));
  return NL_OK;
}

struct BwdCtx { nl_config c32, cbw; Ctx x32, xb; };
static void make_bwd_ctx(BwdCtx& B, const nl_config* cfg, const void* packed, void* stream) {
  B.c32 = *cfg; B.cbw = *cfg;
  // recomputed forward: exact fp32 in the fp32 mode, three-term split-FP16 (products good to ~2^-22, the speed of split-bf16) otherwise — see
  // nl_point_mlp_backward for why split-bf16 is not enough there; the way back: split-bf16
  B.c32.precision = cfg->precision == NL_PREC_F32 ? NL_PREC_F32 : NL_PREC_F16X3_INTERNAL;
  if (B.cbw.precision == NL_PREC_BF16) B.cbw.precision = NL_PREC_BF16X3;
  B.x32 = make_ctx(&B.c32, packed, stream); B.xb = make_ctx(&B.cbw, packed, stream);
}

static size_t point_bwd_bytes(const nl_config* cfg, int64_t n, bool train = false) { Bump b{nullptr, 0}; PtBwdBufs p; carve_ptb(b, cfg, n, 8, p, train); return b.off; }
// nl_train_grads -> TrainOut (validated)
static int resolve_train(const nl_config* cfg, const nl_train_grads* g, TrainOut& t) {
  memset(&t, 0, sizeof(t));
  if (!g) return NL_OK;
  for (int i = 0; i < 4; ++i) if (g->reserved[i] != 0) return NL_ERR_BAD_ARG;
  if (g->weights) for (int i = 0; i < kNumWeights; ++i) t.w[i] = g->weights[i];
  t.sp_feat = g->support_feature;
  t.feat_maps = g->feat_maps; t.pfeat_maps = g->blend_feat_maps; t.vis_maps = g->vis_featmaps;
  if (!g->scratch || g->scratch_bytes < nl_train_scratch_bytes(cfg) || ((uintptr_t)g->scratch & 15)) return NL_ERR_WORKSPACE;
  t.scratch = (float*)g->scratch; t.scratch_floats = g->scratch_bytes / sizeof(float);
  return NL_OK;
}

size_t nl_point_mlp_backward_workspace_bytes(const nl_config* cfg, int64_t N) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  const int64_t n = N < 1 ? 1 : (N > (1 << 14) ? (1 << 14) : N);   // recommended: chunks of <= 16 384 samples (131 072 neighbour rows, ~1.1 GB at W = 256)
  return point_bwd_bytes(cfg, n);
}

size_t nl_train_scratch_bytes(const nl_config* cfg) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  const int F = cfg->C + 3;
  // the largest weight the split-K kernel is asked for: conv_out (W, 3 (W + 32)) is done one tap at a time -> W x (W + 32); out_fc.0 64 x (2F + 3); base_mlp.0 W x (F + 90)
  size_t mx = (size_t)cfg->W * (F + 90);
  if ((size_t)64 * (2 * F + 3) > mx) mx = (size_t)64 * (2 * F + 3);
  if ((size_t)128 * 128 > mx) mx = (size_t)128 * 128;                       // the U-Net's fixed-width layers (one tap / one phase at a time)
  if ((size_t)cfg->C * cfg->W > mx) mx = (size_t)cfg->C * cfg->W;           // feat_mlp.2
  size_t fl = nl_wgrad_scratch_floats(0, 1, (int)(mx + 256));
  const size_t ln = (size_t)258 * 2 * cfg->S * (cfg->W > 64 ? cfg->W : 64);   // the U-Net's LayerNorm tables: 2 S max(W, 64) sums + up to 256 partial rows of them
  if (ln > fl) fl = ln;
  if (nl_dec_wpart_floats() > fl) fl = nl_dec_wpart_floats();   // the decoder backward's per-wave partial sets
  return sizeof(float) * fl;
}
size_t nl_point_mlp_backward_train_workspace_bytes(const nl_config* cfg, int64_t N) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  const int64_t n = N < 1 ? 1 : (N > (1 << 15) ? (1 << 15) : N);
  return point_bwd_bytes(cfg, n, true);
}
int nl_point_mlp_backward(const nl_config* cfg, const void* packed, const nl_frame* f, const float* xyz, const float* dir, int64_t dir_stride,
                          const float* mv_feat, int64_t N, int K, const int32_t* knn_idx, const float* knn_d2, const float* g_feature_agg, float* g_xyz,
                          float* g_dir, float* g_mv_feat, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  return nl_point_mlp_backward_train(cfg, packed, f, xyz, dir, dir_stride, mv_feat, N, K, knn_idx, knn_d2, g_feature_agg, g_xyz, g_dir, g_mv_feat, nullptr, ws,
                                     ws_bytes, stream);
}
int nl_point_mlp_backward_train(const nl_config* cfg, const void* packed, const nl_frame* f, const float* xyz, const float* dir, int64_t dir_stride,
                                const float* mv_feat, int64_t N, int K, const int32_t* knn_idx, const float* knn_d2, const float* g_feature_agg, float* g_xyz,
                                float* g_dir, float* g_mv_feat, const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (N == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !f || !xyz || !mv_feat || !g_feature_agg || !g_xyz || !ws || N < 0 || K < 1 || K > 8 || (g_dir && !dir)) return NL_ERR_BAD_ARG;
  const bool train = grads != nullptr;
  TrainOut T;
  NL_TRY(resolve_train(cfg, grads, T));   // (validated before anything is dereferenced)
  if (f->M < 1) return NL_ERR_UNSUPPORTED;
  if (ws_bytes < point_bwd_bytes(cfg, 1, train)) return NL_ERR_WORKSPACE;
  // Precision of the two halves (measured):
  //  * the RECOMPUTED FORWARD must be much better than split-bf16: the derivative of a LeakyReLU network is piecewise constant, and a forward that
  //    is 1e-5 off flips the sign of a few pre-activations near zero — every flip changes that neighbour row's gradient by a few percent (2e-2 in
  //    the max-norm of g_xyz with a split-bf16 recompute against 4e-6 with exact fp32; plain fp32 autograd is 4e-3 from the fp64 gradient for the
  //    same reason).  Exact fp32 in the fp32 mode; three-term split-FP16 (~2^-22) otherwise: 40x fewer flips than split-bf16 at the same speed;
  //  * the transposed-weight products of the way back are linear in the incoming gradient and run in split-bf16 (1e-5, no discontinuity).
  int64_t lo = 1, hi = N;
  while (lo < hi) {   // largest sample chunk whose buffers fit the workspace
    const int64_t mid = (lo + hi + 1) / 2;
    if (point_bwd_bytes(cfg, mid, train) <= ws_bytes) lo = mid; else hi = mid - 1;
  }
  const int64_t NC = lo < (1 << 17) ? lo : (1 << 17);
  BwdCtx B; make_bwd_ctx(B, cfg, packed, stream);
  const Ctx &xf = B.x32, &xb = B.xb;   // recomputed forward / way back
  const int W = cfg->W;
  for (int64_t n0 = 0; n0 < N; n0 += NC) {
    const int64_t nc = N - n0 < NC ? N - n0 : NC;
    Bump b{(char*)ws, 0}; PtBwdBufs p; carve_ptb(b, cfg, nc, 8, p, train);
    NL_TRY(do_point_backward(xb, xf, f, xyz + 3 * n0, dir ? dir + dir_stride * n0 : nullptr, (int)dir_stride, mv_feat + n0 * W, nc, K, g_feature_agg + n0 * W,
                             g_xyz + 3 * n0, g_dir ? g_dir + 3 * n0 : nullptr, g_mv_feat ? g_mv_feat + n0 * W : nullptr, p,
                             knn_idx ? knn_idx + n0 * K : nullptr, knn_d2 ? knn_d2 + n0 * K : nullptr, train ? &T : nullptr));
  }
  return NL_OK;
}

static size_t mv_bwd_bytes(const nl_config* cfg, int V, int64_t n, bool blend, bool train = false) {
  Bump b{nullptr, 0}; MvBwdBufs m; carve_mvb(b, cfg, V, n, blend, m, train); return b.off;
}
static int64_t mv_bwd_chunk(const nl_config* cfg, int V, int64_t N, bool blend, size_t ws_bytes, bool train = false) {
  if (ws_bytes < mv_bwd_bytes(cfg, V, 1, blend, train)) return 0;
  int64_t lo = 1, hi = N;
  while (lo < hi) { const int64_t mid = (lo + hi + 1) / 2; if (mv_bwd_bytes(cfg, V, mid, blend, train) <= ws_bytes) lo = mid; else hi = mid - 1; }
  return lo < (1 << 18) ? lo : (1 << 18);
}
size_t nl_mv_aggregate_backward_train_workspace_bytes(const nl_config* cfg, int V, int64_t N) {
  NL_EFF_CFG(cfg);
  return cfg_ok(cfg) && V >= 1 && V <= NL_MAX_VIEWS ? mv_bwd_bytes(cfg, V, N < 1 ? 1 : (N > (1 << 15) ? (1 << 15) : N), false, true) : 0;
}
size_t nl_blend_backward_train_workspace_bytes(const nl_config* cfg, int V, int64_t N) {
  NL_EFF_CFG(cfg);
  return cfg_ok(cfg) && V >= 1 && V <= NL_MAX_VIEWS ? mv_bwd_bytes(cfg, V, N < 1 ? 1 : (N > (1 << 15) ? (1 << 15) : N), true, true) : 0;
}
size_t nl_mv_aggregate_backward_workspace_bytes(const nl_config* cfg, int V, int64_t N) {
  NL_EFF_CFG(cfg);
  return cfg_ok(cfg) && V >= 1 && V <= NL_MAX_VIEWS ? mv_bwd_bytes(cfg, V, N < 1 ? 1 : (N > (1 << 16) ? (1 << 16) : N), false) : 0;
}
int nl_mv_aggregate_backward(const nl_config* cfg, const void* packed, const nl_frame* f, const float* xyz, int64_t N, const float* g_mv_feat, float* g_xyz,
                             void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  return nl_mv_aggregate_backward_train(cfg, packed, f, xyz, N, g_mv_feat, g_xyz, nullptr, ws, ws_bytes, stream);
}
int nl_mv_aggregate_backward_train(const nl_config* cfg, const void* packed, const nl_frame* f, const float* xyz, int64_t N, const float* g_mv_feat, float* g_xyz,
                                   const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (N == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !f || !xyz || !g_mv_feat || !g_xyz || !ws || N < 0) return NL_ERR_BAD_ARG;
  const bool train = grads != nullptr;
  TrainOut T;
  NL_TRY(resolve_train(cfg, grads, T));
  const int V = f->views.V, W = cfg->W;
  const int64_t NC = mv_bwd_chunk(cfg, V, N, false, ws_bytes, train);
  if (NC == 0) return NL_ERR_WORKSPACE;
  BwdCtx B; make_bwd_ctx(B, cfg, packed, stream);
  for (int64_t n0 = 0; n0 < N; n0 += NC) {
    const int64_t nc = N - n0 < NC ? N - n0 : NC;
    Bump b{(char*)ws, 0}; MvBwdBufs m; carve_mvb(b, cfg, V, nc, false, m, train);
    NL_TRY(do_mv_backward(B.xb, B.x32, f, xyz + 3 * n0, nc, g_mv_feat + n0 * W, g_xyz + 3 * n0, m, train ? &T : nullptr));
  }
  return NL_OK;
}

size_t nl_blend_workspace_bytes(const nl_config* cfg, int V, int64_t N) {
  NL_EFF_CFG(cfg);
  return cfg_ok(cfg) && V >= 1 && V <= NL_MAX_VIEWS ? mv_bwd_bytes(cfg, V, N < 1 ? 1 : (N > (1 << 16) ? (1 << 16) : N), true) : 0;
}
int nl_blend(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* xyz, const float* feature_agg, int64_t N, float* rgb_s,
             void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (N == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !f || !qc || !xyz || !feature_agg || !rgb_s || !ws || N < 0) return NL_ERR_BAD_ARG;
  const int V = f->views.V, W = cfg->W;
  const int64_t NC = mv_bwd_chunk(cfg, V, N, true, ws_bytes);
  if (NC == 0) return NL_ERR_WORKSPACE;
  Ctx x = make_ctx(cfg, packed, stream);
  for (int64_t n0 = 0; n0 < N; n0 += NC) {
    const int64_t nc = N - n0 < NC ? N - n0 : NC;
    Bump b{(char*)ws, 0}; MvBwdBufs m; carve_mvb(b, cfg, V, nc, true, m);
    NL_TRY(do_blend_forward(x, f, qc, xyz + 3 * n0, feature_agg + n0 * W, nc, rgb_s + 3 * n0, m));
  }
  return NL_OK;
}
int nl_blend_backward(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* xyz, const float* feature_agg, int64_t N,
                      const float* g_rgb_s, float* g_xyz, float* g_feature_agg, float* g_query_center, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  return nl_blend_backward_train(cfg, packed, f, qc, xyz, feature_agg, N, g_rgb_s, g_xyz, g_feature_agg, g_query_center, nullptr, ws, ws_bytes, stream);
}
int nl_blend_backward_train(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* xyz, const float* feature_agg, int64_t N,
                            const float* g_rgb_s, float* g_xyz, float* g_feature_agg, float* g_query_center, const nl_train_grads* grads, void* ws,
                            size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (N == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !f || !qc || !xyz || !feature_agg || !g_rgb_s || !g_xyz || !ws || N < 0) return NL_ERR_BAD_ARG;
  const bool train = grads != nullptr;
  TrainOut T;
  NL_TRY(resolve_train(cfg, grads, T));
  const int V = f->views.V, W = cfg->W;
  const int64_t NC = mv_bwd_chunk(cfg, V, N, true, ws_bytes, train);
  if (NC == 0) return NL_ERR_WORKSPACE;
  BwdCtx B; make_bwd_ctx(B, cfg, packed, stream);
  for (int64_t n0 = 0; n0 < N; n0 += NC) {
    const int64_t nc = N - n0 < NC ? N - n0 : NC;
    Bump b{(char*)ws, 0}; MvBwdBufs m; carve_mvb(b, cfg, V, nc, true, m, train);
    NL_TRY(do_blend_backward(B.xb, B.x32, f, qc, xyz + 3 * n0, feature_agg + n0 * W, nc, g_rgb_s + 3 * n0, g_xyz + 3 * n0,
                             g_feature_agg ? g_feature_agg + n0 * W : nullptr, g_query_center ? g_query_center + 3 * n0 : nullptr, m, train ? &T : nullptr));
  }
  return NL_OK;
}

static size_t unet_bwd_bytes(const nl_config* cfg, int64_t r, bool train = false) { Bump b{nullptr, 0}; UnBwdBufs q; carve_unb(b, cfg, r, q, train); return b.off; }
static size_t render_bwd_bytes(const nl_config* cfg, int V, int64_t r, bool train) { Bump b{nullptr, 0}; RbBufs a; carve_rb(b, cfg, V, r, a, train); return b.off; }
size_t nl_render_rays_backward_workspace_bytes(const nl_config* cfg, int V, int64_t R, int train) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg) || V < 1 || V > NL_MAX_VIEWS) return 0;
  const int64_t cap = (1 << 16) / cfg->S > 1 ? (1 << 16) / cfg->S : 1;   // recommended chunk: ~64 k samples (~70 KB of workspace per sample at W = 256)
  return render_bwd_bytes(cfg, V, R < 1 ? 1 : (R > cap ? cap : R), train != 0);
}
int nl_render_rays_backward(const nl_config* cfg, const void* packed, const nl_frame* f, const float* query_center, const float* ray_centers, const float* rays_o,
                            const float* rays_d, const float* z_vals, int64_t R, int white_bkgd, const nl_render_cotangents* g, float* g_rays_o, float* g_rays_d,
                            float* g_query_center_rows, const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !f || (!query_center && !ray_centers) || !rays_o || !rays_d || !z_vals || !g || !g_rays_o || !g_rays_d || !ws || R < 0) return NL_ERR_BAD_ARG;
  if (g->reserved[0] != nullptr || (g->knn_idx == nullptr) != (g->knn_d2 == nullptr)) return NL_ERR_BAD_ARG;
  const bool train = grads != nullptr;
  TrainOut T;
  NL_TRY(resolve_train(cfg, grads, T));
  if (f->M < 1) return NL_ERR_UNSUPPORTED;
  const int V = f->views.V, S = cfg->S, C = cfg->C;
  if (ws_bytes < render_bwd_bytes(cfg, V, 1, train)) return NL_ERR_WORKSPACE;
  int64_t lo = 1, hi = R;
  while (lo < hi) { const int64_t mid = (lo + hi + 1) / 2; if (render_bwd_bytes(cfg, V, mid, train) <= ws_bytes) lo = mid; else hi = mid - 1; }
  const int64_t RC = lo;
  BwdCtx B; make_bwd_ctx(B, cfg, packed, stream);
  for (int64_t r0 = 0; r0 < R; r0 += RC) {
    const int64_t rc = R - r0 < RC ? R - r0 : RC;
    Bump b{(char*)ws, 0}; RbBufs a; carve_rb(b, cfg, V, rc, a, train);
    RbCot ct{g->g_rgb ? g->g_rgb + 3 * r0 : nullptr, g->g_depth ? g->g_depth + r0 : nullptr, g->g_depth_uncertainty ? g->g_depth_uncertainty + r0 : nullptr,
             g->g_feat ? g->g_feat + r0 * C : nullptr, g->g_weights ? g->g_weights + r0 * S : nullptr,
             g->knn_idx ? g->knn_idx + r0 * S * 8 : nullptr, g->knn_d2 ? g->knn_d2 + r0 * S * 8 : nullptr};
    NL_TRY(do_render_backward(B.xb, B.x32, f, query_center, ray_centers ? ray_centers + 3 * r0 : nullptr, rays_o + 3 * r0, rays_d + 3 * r0, z_vals + r0 * S, rc,
                              white_bkgd, ct, g_rays_o + 3 * r0,
                              g_rays_d + 3 * r0, g_query_center_rows ? g_query_center_rows + 3 * r0 : nullptr, a, train ? &T : nullptr));
  }
  return NL_OK;
}

// The gradient path's forward and backward as a PAIR that shares one workspace: the forward call leaves the staged activations there, the backward call
// walks back from them without recomputing.  The whole batch must fit the workspace as one chunk (NL_ERR_WORKSPACE otherwise: use nl_render_rays +
// nl_render_rays_backward, which chunk).
size_t nl_render_rays_keep_workspace_bytes(const nl_config* cfg, int V, int64_t R, int train) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg) || V < 1 || V > NL_MAX_VIEWS || R < 1) return 0;
  return render_bwd_bytes(cfg, V, R, train != 0);
}
int nl_render_rays_forward_keep(const nl_config* cfg, const void* packed, const nl_frame* f, const float* query_center, const float* ray_centers, const float* rays_o,
                                const float* rays_d, const float* z_vals, int64_t R, int white_bkgd, const nl_render_out* out, const nl_beta_head* beta, int train,
                                void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !f || (!query_center && !ray_centers) || !rays_o || !rays_d || !z_vals || !out || !ws || R < 0) return NL_ERR_BAD_ARG;
  if (beta && (!beta->weight || !beta->bias || !beta->beta)) return NL_ERR_BAD_ARG;
  if (!out->rgb || !out->depth || !out->weights || !out->mask || !out->depth_uncertainty) return NL_ERR_BAD_ARG;
  if (f->M < 1) return NL_ERR_UNSUPPORTED;
  const int V = f->views.V;
  if (ws_bytes < render_bwd_bytes(cfg, V, R, train != 0)) return NL_ERR_WORKSPACE;
  BwdCtx B; make_bwd_ctx(B, cfg, packed, stream);
  Bump b{(char*)ws, 0}; RbBufs a; carve_rb(b, cfg, V, R, a, train != 0);
  NL_TRY(render_forward_staged(B.x32, f, query_center, ray_centers, rays_o, rays_d, z_vals, R, out->feat != nullptr, nullptr, nullptr, a, train == 0));
  return render_outputs_staged(B.x32, f, R, white_bkgd, out, a, beta);
}
int nl_render_rays_backward_kept(const nl_config* cfg, const void* packed, const nl_frame* f, const float* query_center, const float* ray_centers, const float* rays_d,
                                 int64_t R, int white_bkgd, const nl_render_cotangents* g, const nl_beta_head* beta, float* g_rays_o, float* g_rays_d,
                                 float* g_query_center_rows, const nl_train_grads* grads, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !f || (!query_center && !ray_centers) || !rays_d || !g || !g_rays_o || !g_rays_d || !ws || R < 0) return NL_ERR_BAD_ARG;
  if (beta && (!beta->weight || !beta->bias || (beta->g_weight && !grads))) return NL_ERR_BAD_ARG;   // (the weight gradient's split-K scratch comes with `grads`)
  if (g->reserved[0] != nullptr || g->knn_idx || g->knn_d2) return NL_ERR_BAD_ARG;   // (the neighbours are in the workspace)
  const bool train = grads != nullptr;
  TrainOut T;
  NL_TRY(resolve_train(cfg, grads, T));
  if (f->M < 1) return NL_ERR_UNSUPPORTED;
  const int V = f->views.V;
  if (ws_bytes < render_bwd_bytes(cfg, V, R, train)) return NL_ERR_WORKSPACE;
  BwdCtx B; make_bwd_ctx(B, cfg, packed, stream);
  Bump b{(char*)ws, 0}; RbBufs a; carve_rb(b, cfg, V, R, a, train);
  RbCot ct{g->g_rgb, g->g_depth, g->g_depth_uncertainty, g->g_feat, g->g_weights, nullptr, nullptr};
  return render_backward_staged(B.xb, B.x32, f, query_center, ray_centers, rays_d, R, white_bkgd, ct, g_rays_o, g_rays_d, g_query_center_rows, a, train ? &T : nullptr,
                                beta);
}

size_t nl_ray_unet_backward_train_workspace_bytes(const nl_config* cfg, int64_t R) {
  NL_EFF_CFG(cfg);
  return cfg_ok(cfg) ? unet_bwd_bytes(cfg, R < 1 ? 1 : (R > 1024 ? 1024 : R), true) : 0;
}
size_t nl_ray_unet_backward_workspace_bytes(const nl_config* cfg, int64_t R) {
  NL_EFF_CFG(cfg);
  return cfg_ok(cfg) ? unet_bwd_bytes(cfg, R < 1 ? 1 : (R > 1024 ? 1024 : R)) : 0;   // recommended: chunks of <= 1024 rays
}
int nl_ray_unet_backward(const nl_config* cfg, const void* packed, const float* xin, int64_t R, const float* g_geo, float* g_x, void* ws, size_t ws_bytes,
                         void* stream) {
  NL_EFF_CFG(cfg);
  return nl_ray_unet_backward_train(cfg, packed, xin, R, g_geo, g_x, nullptr, ws, ws_bytes, stream);
}
int nl_ray_unet_backward_train(const nl_config* cfg, const void* packed, const float* xin, int64_t R, const float* g_geo, float* g_x, const nl_train_grads* grads,
                               void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;
  if (!cfg_ok(cfg) || !packed || !xin || !g_geo || !g_x || !ws || R < 0) return NL_ERR_BAD_ARG;
  const bool train = grads != nullptr;
  TrainOut T;
  NL_TRY(resolve_train(cfg, grads, T));
  if (ws_bytes < unet_bwd_bytes(cfg, 1, train)) return NL_ERR_WORKSPACE;
  int64_t lo = 1, hi = R;
  while (lo < hi) { const int64_t mid = (lo + hi + 1) / 2; if (unet_bwd_bytes(cfg, mid, train) <= ws_bytes) lo = mid; else hi = mid - 1; }
  const int64_t RC = lo;
  BwdCtx B; make_bwd_ctx(B, cfg, packed, stream);
  const size_t row = (size_t)cfg->S * cfg->W;
  for (int64_t r0 = 0; r0 < R; r0 += RC) {
    const int64_t rc = R - r0 < RC ? R - r0 : RC;
    Bump b{(char*)ws, 0}; UnBwdBufs q; carve_unb(b, cfg, rc, q, train);
    NL_TRY(do_unet_backward(B.xb, B.x32, xin + r0 * row, rc, g_geo + r0 * row, g_x + r0 * row, q, train ? &T : nullptr));
  }
  return NL_OK;
}

size_t nl_ray_unet_workspace_bytes(const nl_config* cfg, int64_t R) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  Bump b{nullptr, 0}; UnBufs u; carve_un(b, cfg, R, u); return b.off;
}

int nl_ray_unet(const nl_config* cfg, const void* packed, const float* xin, int64_t R, float* geo, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!cfg_ok(cfg) || !packed || !xin || !geo || !ws || R < 0) return NL_ERR_BAD_ARG;
  if (ws_bytes < nl_ray_unet_workspace_bytes(cfg, R)) return NL_ERR_WORKSPACE;
  Bump b{(char*)ws, 0}; UnBufs u; carve_un(b, cfg, R, u);
  Ctx x = make_ctx(cfg, packed, stream);
  return do_unet(x, xin, R, geo, u);
}

size_t nl_heads_composite_workspace_bytes(const nl_config* cfg, int V, int64_t R) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  Bump b{nullptr, 0}; HdBufs h; carve_hd(b, cfg, V, R, h); return b.off;
}

int nl_heads_composite(const nl_config* cfg, const void* packed, int V, const float* z, const float* FA, const float* geo,
                       const float* blend1, const float* rgbv, const int32_t* valid_s, int64_t R, int white,
                       const nl_render_out* out, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!cfg_ok(cfg) || !packed || !z || !FA || !geo || !blend1 || !rgbv || !out || !ws || R < 0 || V < 1 || V > NL_MAX_VIEWS) return NL_ERR_BAD_ARG;
  if (ws_bytes < nl_heads_composite_workspace_bytes(cfg, V, R)) return NL_ERR_WORKSPACE;
  Bump b{(char*)ws, 0}; HdBufs h; carve_hd(b, cfg, V, R, h);
  Ctx x = make_ctx(cfg, packed, stream);
  return do_heads(x, V, z, FA, geo, blend1, rgbv, valid_s, R, white, out, 0, h);
}

// ---- fused path ---------------------------------------------------------------------------------------------
static size_t render_bytes(const nl_config* cfg, int V, int64_t rc) {
  Bump b{nullptr, 0}; RenderBufs rb; carve_render(b, cfg, V, rc, rb); return b.off;
}

size_t nl_render_rays_min_workspace_bytes(const nl_config* cfg, int V) {
  NL_EFF_CFG(cfg); return cfg_ok(cfg) ? render_bytes(cfg, V, 1) : 0; }

size_t nl_render_rays_workspace_bytes(const nl_config* cfg, int V, int64_t R) {
  NL_EFF_CFG(cfg);
  if (!cfg_ok(cfg)) return 0;
  int64_t rc = R < 1 ? 1 : R;
  const int64_t cap = (1 << 20) / cfg->S > 0 ? (1 << 20) / cfg->S : 1;  // ~1M samples per chunk (~12 GB of workspace at W=256, V=10): small grids in the U-Net fill the chip only at this size
  if (rc > cap) rc = cap;
  return render_bytes(cfg, V, rc);
}

int nl_render_rays(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* rays_o,
                   const float* rays_d, const float* z_vals, int64_t R, int white, const nl_render_out* out, void* ws,
                   size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  return nl_render_rays_ex(cfg, packed, f, qc, rays_o, rays_d, z_vals, R, white, out, ws, ws_bytes, stream, nullptr);
}

}  // extern "C" (interrupted for two internal helpers)

namespace {
// exactness order of the precision modes (the enum's numbers are historical): BF16 < F16MX < BF16X3 < F32
int prec_rank(int p) { return p == NL_PREC_BF16 ? 0 : p == NL_PREC_F16MX ? 1 : p == NL_PREC_BF16X3 ? 2 : 3; }
// the |attention logit| up to which a mode stayed within 1e-4 of the CPU oracle on every scene of tools/scale_sweep.py; <= 0: no limit known
float guard_limit(int p) { return p == NL_PREC_F16MX ? NL_GUARD_LOGIT_LIMIT_F16MX : p == NL_PREC_BF16X3 ? NL_GUARD_LOGIT_LIMIT_BF16X3 : 0.f; }
int guard_safer(int p) { return p == NL_PREC_F16MX ? NL_PREC_BF16X3 : NL_PREC_F32; }
int render_rays_impl(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* rays_o, const float* rays_d, const float* z_vals,
                     int64_t R, int white, const nl_render_out* out, void* ws, size_t ws_bytes, void* stream, const nl_render_opts* opts);
}  // namespace

extern "C" {
int nl_render_rays_ex(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* rays_o,
                      const float* rays_d, const float* z_vals, int64_t R, int white, const nl_render_out* out, void* ws,
                      size_t ws_bytes, void* stream, const nl_render_opts* opts) {
  if (!opts || !(opts->flags & NL_RENDER_PRECISION_GUARD) || !cfg || !f || R <= 0)
    return render_rays_impl(cfg, packed, f, qc, rays_o, rays_d, z_vals, R, white, out, ws, ws_bytes, stream, opts);
  // ---- NL_RENDER_PRECISION_GUARD: render, read the frame's conditioning indicator (one 4-byte copy + a stream synchronisation), and while it is beyond the
  // validated range of the mode the outputs were produced in, render THIS batch again in the next more exact mode.  The frame stays in that mode for every
  // later guarded call (a frame whose attention logits are large once has them large in every batch), so the extra pass is paid once per frame.
  nl_config c = *cfg;
  if (f->guard_prec >= 0 && prec_rank(f->guard_prec) > prec_rank(c.precision)) c.precision = f->guard_prec;
  for (;;) {
    const int rc = render_rays_impl(&c, packed, f, qc, rays_o, rays_d, z_vals, R, white, out, ws, ws_bytes, stream, opts);
    if (rc != NL_OK) return rc;
    f->guard_last_prec = c.precision;
    const float limit = guard_limit(c.precision);
    if (limit <= 0.f) return NL_OK;
    float amax = 0.f;
    NL_CHECK_HIP(hipMemcpyAsync(&amax, f->views_dev + 249, sizeof(float), hipMemcpyDeviceToHost, (hipStream_t)stream));
    NL_CHECK_HIP(hipStreamSynchronize((hipStream_t)stream));
    if (amax <= limit) return NL_OK;   // (the kernels record a NaN logit as +inf: it escalates)
    c.precision = guard_safer(c.precision);
    f->guard_prec = c.precision;
    ++f->guard_escalations;
  }
}
}  // extern "C"

namespace {
int render_rays_impl(const nl_config* cfg, const void* packed, const nl_frame* f, const float* qc, const float* rays_o,
                     const float* rays_d, const float* z_vals, int64_t R, int white, const nl_render_out* out, void* ws,
                     size_t ws_bytes, void* stream, const nl_render_opts* opts) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  const float term_eps = opts ? opts->early_term_eps : 0.f;
  if (!(term_eps >= 0.f && term_eps < 1.f)) return NL_ERR_BAD_ARG;   // (written so that NaN is rejected)
  if (opts) {   // fields without a meaning in this ABI version must be zero, so that a later version can give them one
    if (opts->flags & ~(uint32_t)NL_RENDER_FLAGS_ALL) return NL_ERR_BAD_ARG;
    for (int i = 0; i < 4; ++i) if (opts->reserved[i] != 0) return NL_ERR_BAD_ARG;
  }
  const uint32_t flags = opts ? opts->flags : 0u;
  const float* ray_centers = opts ? opts->ray_centers : nullptr;
  if (!cfg_ok(cfg) || !packed || !f || (!qc && !ray_centers) || !rays_o || !rays_d || !out || !ws || R < 0) return NL_ERR_BAD_ARG;
  const int V = f->views.V, S = cfg->S, W = cfg->W;
  // largest ray chunk whose buffers fit the workspace
  const size_t b1 = render_bytes(cfg, V, 1);
  if (ws_bytes < b1) return NL_ERR_WORKSPACE;
  int64_t lo = 1, hi = R > 1 ? R : 1;
  while (lo < hi) {
    int64_t mid = (lo + hi + 1) / 2;
    if (render_bytes(cfg, V, mid) <= ws_bytes) lo = mid; else hi = mid - 1;
  }
  int64_t RC = lo;
  {   // the buffer-addressed kernels use 32-bit byte offsets: at most 2^21 samples per chunk (twice the recommended workspace's chunk)
    const int64_t cap = ((int64_t)1 << 21) / S > 0 ? ((int64_t)1 << 21) / S : 1;
    if (RC > cap) RC = cap;
  }
  Bump b{(char*)ws, 0}; RenderBufs rb; carve_render(b, cfg, V, RC, rb);
  Ctx x = make_ctx(cfg, packed, stream);
  x.mx = nl_mx_;
  const bool fork = f->side_ok && !(flags & NL_RENDER_NO_SIDE_STREAM) && nl_point_fused_supported(W, cfg->precision);
  for (int64_t r0 = 0; r0 < R; r0 += RC) {
    const int64_t rc = (R - r0 < RC) ? R - r0 : RC;
    const int64_t N = rc * S;
    NL_TRY(nl_launch_sample_points(rays_o + 3 * r0, rays_d + 3 * r0, rc, S, f->views.near_, f->views.far_,
                                   z_vals ? z_vals + r0 * S : nullptr, rb.z, rb.xyz, x.st));
    // ---- fork: exact KNN + aggregation scale on the frame's side stream, beside the multi-view gather kernels (both need xyz only).
    // `knn` joins in do_point before the neural-point kernel — or in its destructor on any earlier exit from this iteration.
    SideJoin knn;
    if (fork) {
      NL_CHECK_HIP(hipEventRecord(f->ev_fork, x.st));
      NL_CHECK_HIP(hipStreamWaitEvent(f->side, f->ev_fork, 0));
      knn.arm(x.st, f->side, f->ev_join);
#ifndef NL_KNN_PARTS
#define NL_KNN_PARTS 1   // measured (profiles/r6_knn_parts_{1,2,4}_timeline.txt): 2 / 4 parts shorten the front end by 0.3 / 0.4 ms and lengthen the neural-point span by 0.41 / 0.92 ms — off
#endif
#ifndef NL_KNN_PARTS_MIN_ROWS
#define NL_KNN_PARTS_MIN_ROWS (1 << 18)
#endif
      // the search in parts (do_point): part 0 now, the others as the neural-point launches of the parts before them start
      const int parts = (f->parts_ok && NL_KNN_PARTS > 1 && N >= NL_KNN_PARTS_MIN_ROWS && rc >= NL_KNN_PARTS && !dbg_switch("NERFLOC_NO_KNN_PARTS")) ? NL_KNN_PARTS : 1;
      static_assert(NL_KNN_PARTS <= nl_frame::kMaxParts, "events per frame");
      const int64_t part_rays = (rc + parts - 1) / parts, n_first = parts > 1 ? part_rays * S : N;
      NL_TRY(nl_knn_search(&f->grid, rb.xyz, n_first, 8, rb.pt.idx, rb.pt.d2, f->side));
      NL_TRY(nl_launch_wscale(rb.pt.idx, rb.pt.d2, f->sp_conf, n_first, 8, f->M, rb.pt.wscale, f->side));
      if (parts > 1) { NL_CHECK_HIP(hipEventRecord(f->ev_done[0], f->side)); knn.parts = parts; knn.part_rays = part_rays; }
    }
    // the chain kernels recompute the multiview feature rows G (N x W) from out_fc's 64-wide hidden rows: G is only materialised for
    // the stage output or when the separate launches run instead
    const bool use_chain = !dbg_switch("NERFLOC_NO_CHAIN") && W == 256 && cfg->precision != NL_PREC_F32 && N * 1024 <= 0x7fffffffll &&
                           nl_point_fused_supported(W, cfg->precision);
    const bool front = front_path(cfg, V) && f->C == cfg->C && nl_mv_front_supported(f->C, V, N) && rb.bl1 == nullptr;
    NL_TRY(do_mv(x, f, qc, rb.xyz, N, rb.G, nullptr, nullptr, rb.valid_s, front ? nullptr : rb.bl1, rb.rgbv, rb.mv, use_chain && !out->mv_feature_agg,
                 ray_centers ? ray_centers + 3 * r0 : nullptr, S, front));
    BlendTaps bt{with_query(f, qc, ray_centers ? ray_centers + 3 * r0 : nullptr, S), f->views_dev, f->pfeat, rb.xyz};
    // per-sample viewing direction = its ray's direction (model.py:501-504): row = sample / S
    // with early termination feat_mlp.0 runs later, over the live tiles only; otherwise the chain kernel produces it right here
    bool chain_done = false;
    const bool want_feat = out->feat != nullptr;
    // feature_agg has two consumers left on this path — conv1 and conv_out, three taps each: the chain kernel hands it over as the split-bf16 fragments it
    // holds anyway (same bytes in the same buffer) unless someone wants the fp32 rows: the stage output, or feat_mlp.0 over the live tiles of an early-terminated batch
    const bool fa_frag = use_chain && !dbg_switch("NERFLOC_NO_FRAG") && (N & 31) == 0 && !out->feature_agg && !(want_feat && term_eps > 0.f);
    // f16mx: feat_mlp.0 leaves the chain kernel — it runs after the density, fused with the compositing of its rows (do_heads: feat_late)
    // (the kernel reads the samples' weights as 16-byte rows: a caller's `weights` buffer that is not 16-byte aligned keeps the old path — no alignment was ever asked of it)
    const bool feat_late = want_feat && term_eps == 0.f && fa_frag && x.mx && ((x.has_bsh >> G_FEAT0P) & 1) && nl_feat_comp_mx_supported(W, S, N) &&
                           (((size_t)out->weights) & 15) == 0 && !dbg_switch("NERFLOC_NO_FEAT_COMP");
    // split-FP16 fragments where EVERY consumer of the image multiplies in fp16-based arithmetic: conv1 -> tgemm_conv1_kernel<true, true>, conv_out -> tgemm_mx_kernel,
    // feat_mlp.0 -> feat_comp_mx_kernel (or nobody), the blend projection inside the chain kernel on its fp16 stream (W = 256, S = 128, f16mx)
    const bool fa_f16 = fa_frag && x.mx && W == 256 && S == 128 && (!want_feat || feat_late) && ((x.has_bsh >> G_CONV1F) & 1) && ((x.has_bsh >> G_CONVOUTF) & 1) &&
                        ((x.has_bsh >> G_BLENDAP) & 1) && !dbg_switch("NERFLOC_NO_TGEMM_MX") && !dbg_switch("NERFLOC_NO_F16FRAG");
    const ChainOut chain{(want_feat && term_eps == 0.f && !feat_late) ? rb.hd.fth : nullptr, rb.hd.blA, &chain_done, use_chain ? rb.mv.t64 : nullptr, fa_frag, fa_f16};
    NL_TRY(do_point(x, f, rb.xyz, rays_d + 3 * r0, 3, S, rb.G, N, 8, rb.FA, rb.pt, fork ? &knn : nullptr, &chain));
    const int chain_parts = chain_done ? ((want_feat && term_eps == 0.f && !feat_late ? 1 : 0) | 2) : 0;
    bool have_sigma = false;
    NL_TRY(do_unet(x, rb.FA, rc, rb.geo, rb.un, rb.hd.sigma, &have_sigma, out->geo != nullptr, fa_frag ? (fa_f16 ? 3 : 1) : 0, true));
    // (The colour-blend taps on the frame's side stream beside the ray U-Net's kernels — they need the chain kernel's projection rows, not the density — built and
    // traced: conv1 stretches by what the taps take (conv1 399 us with the taps' 369 us inside it = a 403-us span against 192 + 222 us one after the other;
    // profiles/r6_blend_side_stream.txt).  Not kept.)
    NL_TRY(do_heads(x, V, rb.z, rb.FA, rb.geo, front ? nullptr : rb.bl1, rb.rgbv, rb.valid_s, rc, white, out, r0, rb.hd, have_sigma, false, term_eps, chain_parts, &bt,
                    feat_late && chain_done, fa_f16));
    if (out->feature_agg) NL_CHECK_HIP(hipMemcpyAsync(out->feature_agg + r0 * S * W, rb.FA, sizeof(float) * N * W, hipMemcpyDeviceToDevice, x.st));
    if (out->mv_feature_agg) NL_CHECK_HIP(hipMemcpyAsync(out->mv_feature_agg + r0 * S * W, rb.G, sizeof(float) * N * W, hipMemcpyDeviceToDevice, x.st));
    if (out->geo) NL_CHECK_HIP(hipMemcpyAsync(out->geo + r0 * S * W, rb.geo, sizeof(float) * N * W, hipMemcpyDeviceToDevice, x.st));
    if (out->knn_idx) NL_CHECK_HIP(hipMemcpyAsync(out->knn_idx + r0 * S * 8, rb.pt.idx, sizeof(int) * N * 8, hipMemcpyDeviceToDevice, x.st));
    if (out->knn_d2) NL_CHECK_HIP(hipMemcpyAsync(out->knn_d2 + r0 * S * 8, rb.pt.d2, sizeof(float) * N * 8, hipMemcpyDeviceToDevice, x.st));
  }
  return NL_OK;
}
}  // namespace

extern "C" {
// ---- several frames per call: fork / join over library-owned streams -----------------------------------------------------------------------
namespace {
// One pool of lane streams / events PER DEVICE (keyed by hipGetDevice(): streams and events belong to the device that was current when they were
// created — a process that renders on cuda:0 and later on cuda:1 must not launch cuda:1's work on cuda:0's streams)
struct MultiPool { std::vector<hipStream_t> st; std::vector<hipEvent_t> ev; hipEvent_t fork = nullptr; };
std::mutex g_multi_mu;
std::unordered_map<int, MultiPool> g_multi;
}  // namespace

int nl_render_rays_multi(const nl_config* cfg, const void* packed, const nl_render_job* jobs, int32_t njobs, int32_t white, void* stream) {
  if (njobs == 0) return NL_OK;
  if (!cfg || !packed || !jobs || njobs < 0) return NL_ERR_BAD_ARG;
  for (int i = 0; i < njobs; ++i) if (!jobs[i].frame || !jobs[i].out || !jobs[i].ws) return NL_ERR_BAD_ARG;
  if (njobs == 1) return nl_render_rays_ex(cfg, packed, jobs[0].frame, jobs[0].query_center, jobs[0].rays_o, jobs[0].rays_d, jobs[0].z_vals, jobs[0].R, white,
                                           jobs[0].out, jobs[0].ws, jobs[0].ws_bytes, stream, jobs[0].opts);
  // the precision guard synchronises its stream after every job: with concurrent lanes it would serialise them — callers check nl_frame_diagnostics per frame instead
  for (int i = 0; i < njobs; ++i) if (jobs[i].opts && (jobs[i].opts->flags & NL_RENDER_PRECISION_GUARD)) return NL_ERR_BAD_ARG;
  // a lane (stream) per distinct frame: a frame's side stream and events serve one render call at a time
  std::vector<int> lane_of(njobs);
  std::vector<const nl_frame*> lanes;
  for (int i = 0; i < njobs; ++i) {
    int l = -1;
    for (size_t k = 0; k < lanes.size(); ++k) if (lanes[k] == jobs[i].frame) { l = (int)k; break; }
    if (l < 0) { l = (int)lanes.size(); lanes.push_back(jobs[i].frame); }
    lane_of[i] = l;
  }
  // jobs on DIFFERENT lanes run concurrently: their workspaces must not overlap (jobs of one lane run one after the other and may share one)
  for (int i = 0; i < njobs; ++i)
    for (int k = i + 1; k < njobs; ++k) {
      if (lane_of[i] == lane_of[k]) continue;
      const char *a0 = (const char*)jobs[i].ws, *a1 = a0 + jobs[i].ws_bytes, *b0 = (const char*)jobs[k].ws, *b1 = b0 + jobs[k].ws_bytes;
      if (a0 < b1 && b0 < a1) return NL_ERR_BAD_ARG;
    }
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return NL_ERR_HIP;
  std::lock_guard<std::mutex> lk(g_multi_mu);
  MultiPool& mp = g_multi[dev];
  while (mp.st.size() < lanes.size()) {
    hipStream_t s; hipEvent_t e;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) return NL_ERR_HIP;
    mp.st.push_back(s); mp.ev.push_back(e);
  }
  if (!mp.fork && hipEventCreateWithFlags(&mp.fork, hipEventDisableTiming) != hipSuccess) return NL_ERR_HIP;
  hipStream_t main = (hipStream_t)stream;
  NL_CHECK_HIP(hipEventRecord(mp.fork, main));
  for (size_t k = 0; k < lanes.size(); ++k) NL_CHECK_HIP(hipStreamWaitEvent(mp.st[k], mp.fork, 0));
  int rc = NL_OK;
  for (int i = 0; i < njobs && rc == NL_OK; ++i)
    rc = nl_render_rays_ex(cfg, packed, jobs[i].frame, jobs[i].query_center, jobs[i].rays_o, jobs[i].rays_d, jobs[i].z_vals, jobs[i].R, white, jobs[i].out,
                           jobs[i].ws, jobs[i].ws_bytes, mp.st[lane_of[i]], jobs[i].opts);
  // join unconditionally: whatever was enqueued must be ordered before the caller's next work (and an active capture must stay well-formed)
  for (size_t k = 0; k < lanes.size(); ++k) {
    if (hipEventRecord(mp.ev[k], mp.st[k]) != hipSuccess || hipStreamWaitEvent(main, mp.ev[k], 0) != hipSuccess) rc = rc == NL_OK ? NL_ERR_HIP : rc;
  }
  return rc;
}

size_t nl_coarse_weights_workspace_bytes(int V, int64_t R, int Sc) {
  return 3 * nl_align_up((size_t)(V > 0 ? V : 1) * (R > 0 ? R : 1) * (Sc > 0 ? Sc : 1) * 4, 256);
}

int nl_coarse_weights(const nl_config* cfg, const void* packed, const nl_frame* f, const float* w2c_kinv, const float* pix,
                      const float* zc, int64_t R, int Sc, float* weights, float* depth_coarse, void* ws, size_t ws_bytes, void* stream) {
  NL_EFF_CFG(cfg);
  if (R == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!cfg_ok(cfg) || !packed || !f || !w2c_kinv || !pix || !zc || !weights || !ws || R < 0) return NL_ERR_BAD_ARG;
  if (ws_bytes < nl_coarse_weights_workspace_bytes(f->views.V, R, Sc)) return NL_ERR_WORKSPACE;
  const size_t part = nl_align_up((size_t)f->views.V * R * Sc * 4, 256);
  float* wa = (float*)ws; float* wv = (float*)((char*)ws + part); float* wm = (float*)((char*)ws + 2 * part);
  const Layout L = make_layout(cfg);
  return nl_launch_coarse_weights(f->views, w2c_kinv, f->visf_hwc, (const float*)((const char*)packed + L.dec_w), (const char*)packed + L.dec_mfma,
                                  cfg->precision, pix, zc, R, Sc, wa, wv, wm, weights, depth_coarse, (hipStream_t)stream);
}

int nl_sample_pdf(const float* zc, const float* wc, int Sc, const float* u, int Ni, const float* zb, int Sb, int64_t R, float* z_out,
                  void* stream) {
  if (R == 0) return NL_OK;   // empty batch: nothing to do, data pointers may be null
  if (!zc || !wc || !u || !zb || !z_out || R < 0) return NL_ERR_BAD_ARG;
  return nl_launch_sample_pdf(zc, wc, Sc, u, Ni, zb, Sb, R, z_out, (hipStream_t)stream);
}

}  // extern "C"
