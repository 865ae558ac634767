// extern "C" surface of librmcl_hip.so (declared in include/rmcl.h) + the heads (pooler / MoCo head)
// forward and backward, which are tiny [B,768] problems run on the exact-f32 GEMM.
#include "rmcl_common.h"
#include "kernels.h"
#include "../../include/rmcl.h"
#include <string>

static thread_local std::string g_err;
extern "C" void rmcl_set_error(const char* msg) { g_err = msg ? msg : ""; }

int rmcl_attention_fwd_impl(const void* qkv, const int* mask, void* out, void* probs, float* scores, int B, int N, int H, int dt,
                            int exact, hipStream_t s);
int rmcl_attention_bwd_impl(const void* qkv, const int* mask, const void* probs, const void* dout, const void* out, void* dqkv,
                            float* scores, void* dS, int B, int N, int H, int dt, int exact, hipStream_t s);

// ---- optional in-stream timing of one GEMM class (bench.py roofline leg) ------------------------
#include <vector>
namespace {
struct Prof {
  bool on = false;
  int mask = 0;
  size_t n = 0, cap = 0;
  std::vector<hipEvent_t> ev;   // 2 per launch
  double flops = 0.0;
} g_prof;
}

int rmcl_launch_gemm(const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc, int exact, hipStream_t stream) {
  const bool timed = g_prof.on && (g.tag & g_prof.mask) && g_prof.n < g_prof.cap;
  if (timed) (void)hipEventRecord(g_prof.ev[2 * g_prof.n], stream);
  int rc;
  if (!exact && rmcl_gemm_fast_supported(g, dt_in, dt_out, a_kc, b_kc)) rc = rmcl_launch_gemm_fast(g, dt_out, a_kc, b_kc, stream);
  else rc = rmcl_launch_gemm_exact(g, dt_in, dt_out, a_kc, b_kc, stream);
  if (timed) {
    (void)hipEventRecord(g_prof.ev[2 * g_prof.n + 1], stream);
    g_prof.flops += 2.0 * g.M * g.N * g.K * (g.splitk > 1 ? 1 : g.nb1 * g.nb2);
    ++g_prof.n;
  }
  return rc;
}

void rmcl_gemm_fast_set_cfg(int cfg);

namespace {
struct HeadStash {
  float *cls_in, *pooled, *h1, *mean, *rstd, *h2r, *z, *q, *nrm;
};
size_t carve_heads(const rmcl_dims& d, void* base, HeadStash* hs) {
  char* b = reinterpret_cast<char*>(base);
  size_t off = 0;
  auto take = [&](size_t n) { off = (off + 255) & ~(size_t)255; float* p = b ? reinterpret_cast<float*>(b + off) : nullptr; off += n * 4; return p; };
  HeadStash h;
  const size_t B = d.B, D = d.D;
  h.cls_in = take(B * D); h.pooled = take(B * D); h.h1 = take(B * D);
  h.mean = take(B); h.rstd = take(B); h.h2r = take(B * D);
  h.z = take(B * d.proj); h.q = take(B * d.proj); h.nrm = take(B);
  if (hs) *hs = h;
  return off;
}
GemmArgs ga(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, int ldc) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = 1.f; g.splitk = 1; g.nb1 = 1; g.nb2 = 1;
  return g;
}
}  // namespace

extern "C" {

const char* rmcl_last_error(void) { return g_err.c_str(); }

int rmcl_prof_begin(int tag_mask, int max_launches) {
  RMCL_REQUIRE(max_launches > 0 && max_launches <= (1 << 16), "prof_begin: bad max_launches");
  while (g_prof.ev.size() < (size_t)2 * max_launches) {
    hipEvent_t e;
    hipError_t r = hipEventCreate(&e);
    if (r != hipSuccess) { rmcl_set_error(hipGetErrorString(r)); return (int)r; }
    g_prof.ev.push_back(e);
  }
  g_prof.on = true; g_prof.mask = tag_mask; g_prof.n = 0; g_prof.cap = max_launches; g_prof.flops = 0.0;
  return 0;
}
int rmcl_prof_end(double* ms_total, int64_t* launches, double* flops_total) {
  g_prof.on = false;
  double ms = 0.0;
  for (size_t i = 0; i < g_prof.n; ++i) {
    hipError_t r = hipEventSynchronize(g_prof.ev[2 * i + 1]);
    float t = 0.f;
    if (r == hipSuccess) r = hipEventElapsedTime(&t, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
    if (r != hipSuccess) { rmcl_set_error(hipGetErrorString(r)); return (int)r; }
    ms += t;
  }
  if (ms_total) *ms_total = ms;
  if (launches) *launches = (int64_t)g_prof.n;
  if (flops_total) *flops_total = g_prof.flops;
  return 0;
}
int rmcl_version(void) { return 1; }

int rmcl_dropout_mask_apply(float* x, int64_t n, uint32_t drop_seed, int layer, int site, float drop_p, void* stream) {
  RMCL_REQUIRE(x && drop_p >= 0.f && drop_p < 1.f, "dropout_mask_apply: bad argument");
  if (drop_p == 0.f) return 0;
  return rmcl_dropout_apply(x, n, rmcl_site_seed(drop_seed, layer, site), (uint32_t)((double)drop_p * 4294967296.0), 1.0f / (1.0f - drop_p),
                            (hipStream_t)stream);
}
extern int g_st_reserve_cus;
extern bool g_attn_fused_bwd;
extern bool g_dw_grouped;
extern int g_attn_fwd_waves;
extern int g_infonce_fold;
extern int g_dp_stagger;
extern int g_attn_bwd_persist;
extern int g_attn_bwd_tpw;
extern int g_gemm_share;
extern bool g_lnfold_centred;
extern int g_prefetch_mask, g_prefetch_wgs;
int rmcl_l2_prefetch_experiment(const void* A, int64_t lda_bytes, int M, int rows_per_tile, int tiles_per_xcd, int col_tiles, const void* B,
                                int64_t ldb_bytes, int nB, int nk, int tick, int lead, int per_xcd, int wgs, int* counter, int64_t* stamps, void* stream) {
  RMCL_REQUIRE(A && B && counter && M > 0 && nk > 0 && per_xcd > 0 && wgs > 0 && col_tiles > 0, "l2_prefetch_experiment: bad argument");
  return rmcl_debug_l2_prefetch(A, (long)lda_bytes, M, rows_per_tile, tiles_per_xcd, col_tiles, B, (long)ldb_bytes, nB, nk, tick, lead, per_xcd, wgs, counter,
                                reinterpret_cast<long long*>(stamps), (hipStream_t)stream);
}
int rmcl_tune_set(int key, int value) {
  if (key == 11) { g_lnfold_centred = value != 0; return 0; }
  if (key == 12) { g_prefetch_mask = value & 7; return 0; }                                          // stash prefetch one layer ahead of the backward (encoder.cpp)
  if (key == 13) { g_prefetch_wgs = value < 0 ? 0 : (value > 64 ? 64 : value); return 0; }       // (0: the event traffic only, no touch launches)
  if (key == 7) { g_dp_stagger = value < 0 ? 0 : value; return 0; }                                  // gemm_dp: start delay of every CU's second workgroup (10 ns ticks)
  if (key == 10) { g_gemm_share = value < 1 ? 1 : (value > 8 ? 8 : value); return 0; }                 // chains sharing the chip (GEMM routing sizes a launch against CUs / share)
  if (key == 9) { g_attn_bwd_tpw = value; return 0; }                                             // fused attention backward: key tiles per wave (1, 2, 3)
  if (key == 8) { g_attn_bwd_persist = value < 0 ? 0 : value; return 0; }                           // fused attention backward: workgroups (0: one per problem)
  if (key == 6) { rmcl_gemm_skinny_set_form(value); return 0; }                                   // 0: skinny GEMMs in the row-split form only
  if (key == 0) { rmcl_gemm_fast_set_cfg(value); return 0; }
  if (key == 1) { g_st_reserve_cus = value < 0 ? 0 : (value > 128 ? 128 : value); return 0; }   // CUs left free by the activation GEMMs
  if (key == 2) { g_attn_fused_bwd = value != 0; return 0; }                                    // 0: two-kernel attention backward
  if (key == 5) { g_infonce_fold = value != 0; return 0; }                                        // 0: one-slice-per-workgroup InfoNCE + per-row combine
  if (key == 4) { g_attn_fwd_waves = value; return 0; }                                          // waves per workgroup of the attention forward
  if (key == 3) { g_dw_grouped = value != 0; return 0; }                                        // 0: per-GEMM weight gradients (split-K slabs)
  rmcl_set_error("tune_set: unknown key");
  return -1;
}

int64_t rmcl_heads_stash_bytes(const rmcl_dims* d) { return (int64_t)carve_heads(*d, nullptr, nullptr) + 256; }

int rmcl_heads_forward(const rmcl_dims* d, const float* pool32, const float* head32, const float* xn, void* hstash,
                       float* cls_feats, float* q, void* stream) {
  return rmcl_heads_forward2(d, pool32, head32, xn, hstash, cls_feats, q, 0, stream);
}
int rmcl_heads_forward2(const rmcl_dims* d, const float* pool32, const float* head32, const float* xn, void* hstash,
                        float* cls_feats, float* q, int flags, void* stream) {
  RMCL_REQUIRE(d && pool32 && xn && hstash && cls_feats, "heads_forward: NULL argument");
  RMCL_REQUIRE((flags & ~RMCL_HEADS_NO_WGRAD) == 0, "heads_forward: unknown flag");
  rmcl_layout y;
  rmcl_param_layout(d, &y);
  HeadStash h;
  carve_heads(*d, hstash, &h);
  hipStream_t s = (hipStream_t)stream;
  const int B = d->B, D = d->D, N = d->L + 1 + d->P;
  // hidden_states[:, 0] (heads.py:17): a compact copy for the pooler's weight gradient - or, when the matching backward forms none
  // (key pass, PGD passes), the pooler GEMM reads the rows in place (row stride N * D) and the gather launch is not issued
  const bool stash_in = !(flags & RMCL_HEADS_NO_WGRAD);
  if (stash_in) RMCL_TRY(rmcl_gather_rows(xn, h.cls_in, B, D, 1, N, 0, s));
  {
    GemmArgs g = stash_in ? ga(h.cls_in, pool32 + y.pool_w, h.pooled, B, D, D, D, D, D)
                          : ga(xn, pool32 + y.pool_w, h.pooled, B, D, D, N * D, D, D);
    g.epi = EPI_BIAS | EPI_TANH | EPI_DUP; g.bias = pool32 + y.pool_b; g.C2 = cls_feats;   // (cls_feats: written by the GEMM, no copy launch)
    const bool dup = rmcl_gemm_skinny_supported(g, RMCL_F32, RMCL_F32, 1);                  // (only the skinny kernels know EPI_DUP)
    if (!dup) { g.epi &= ~EPI_DUP; g.C2 = nullptr; }
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 1, s));
    if (!dup) {
      hipError_t e = hipMemcpyAsync(cls_feats, h.pooled, (size_t)B * D * 4, hipMemcpyDeviceToDevice, s);
      if (e != hipSuccess) { rmcl_set_error(hipGetErrorString(e)); return (int)e; }
    }
  }
  if (!q) return 0;
  RMCL_REQUIRE(head32, "heads_forward: head arena is NULL");
  {
    GemmArgs g = ga(h.pooled, head32 + y.mh0_w, h.h1, B, D, D, D, D, D);
    g.epi = EPI_BIAS; g.bias = head32 + y.mh0_b;
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 1, s));
  }
  RMCL_TRY(rmcl_ln_fwd(h.h1, D, head32 + y.mh1_w, head32 + y.mh1_b, 1e-5f, h.h2r, D, RMCL_F32, h.mean, h.rstd, B, D, 1, s));
  {
    GemmArgs g = ga(h.h2r, head32 + y.mh3_w, h.z, B, d->proj, D, D, D, d->proj);
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 1, s));
  }
  RMCL_TRY(rmcl_l2norm_fwd(h.z, h.q, h.nrm, B, d->proj, 1e-12f, s, q));
  return 0;
}

int rmcl_heads_backward(const rmcl_dims* d, const float* pool32, const float* head32, void* hstash, const float* dq,
                        const float* dcls_extra, float* dcls, float* G, void* workspace, void* stream) {
  RMCL_REQUIRE(d && pool32 && hstash && dcls && workspace, "heads_backward: NULL argument");
  rmcl_layout y;
  rmcl_param_layout(d, &y);
  HeadStash h;
  carve_heads(*d, hstash, &h);
  hipStream_t s = (hipStream_t)stream;
  const int B = d->B, D = d->D, Pd = d->proj;
  // scratch (all [B,D] f32) carved from the start of the encoder workspace
  float* t0 = reinterpret_cast<float*>(workspace);
  float* t1 = t0 + (size_t)B * D;
  float* t2 = t1 + (size_t)B * D;
  hipError_t e;
  if (dq) {
    RMCL_REQUIRE(head32, "heads_backward: head arena is NULL");
    RMCL_TRY(rmcl_l2norm_bwd(dq, h.q, h.nrm, t0, B, Pd, s));                            // dz
    if (G) {
      GemmArgs g = ga(t0, h.h2r, G + y.mh3_w, Pd, D, B, Pd, D, D);                       // dW3 += dz^T h2r
      g.epi = EPI_ACCUM;
      RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
    }
    {
      GemmArgs g = ga(t0, head32 + y.mh3_w, t1, B, D, Pd, Pd, D, D);                     // dh2r = dz W3
      RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 0, s));
    }
    RMCL_TRY(rmcl_ln_bwd(t1, D, RMCL_F32, h.h1, D, h.mean, h.rstd, head32 + y.mh1_w, head32 + y.mh1_b, t2, D, 0,
                         G ? G + y.mh1_w : nullptr, G ? G + y.mh1_b : nullptr, B, D, 1, s));   // dh1 (ReLU mask inside)
    if (G) {
      GemmArgs g = ga(t2, h.pooled, G + y.mh0_w, D, D, B, D, D, D);                      // dW0 += dh1^T pooled
      g.epi = EPI_ACCUM;
      const bool cs = rmcl_gemm_tn_shortk_takes(g);                                      // bias gradient as a by-product of the same launch
      if (cs) { g.epi |= EPI_COLSUM; g.colsum = G + y.mh0_b; }
      RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
      if (!cs) RMCL_TRY(rmcl_colsum(t2, D, RMCL_F32, G + y.mh0_b, B, D, s));
    }
    {
      GemmArgs g = ga(t2, head32 + y.mh0_w, t0, B, D, D, D, D, D);                       // dpooled = dh1 W0
      RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 0, s));
    }
    if (dcls_extra) RMCL_TRY(rmcl_scatter_rows(dcls_extra, t0, B, D, B, 0, 0, 1, s));    // += gradient from other heads
  } else {
    RMCL_REQUIRE(dcls_extra, "heads_backward: neither dq nor dcls_extra given");
    e = hipMemcpyAsync(t0, dcls_extra, (size_t)B * D * 4, hipMemcpyDeviceToDevice, s);
    if (e != hipSuccess) { rmcl_set_error(hipGetErrorString(e)); return (int)e; }
  }
  RMCL_TRY(rmcl_tanh_bwd(t0, h.pooled, (long)B * D, s));                                  // through tanh
  if (G) {
    GemmArgs g = ga(t0, h.cls_in, G + y.pool_w, D, D, B, D, D, D);                        // dWp += dpre^T cls_in
    g.epi = EPI_ACCUM;
    const bool cs = rmcl_gemm_tn_shortk_takes(g);
    if (cs) { g.epi |= EPI_COLSUM; g.colsum = G + y.pool_b; }
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
    if (!cs) RMCL_TRY(rmcl_colsum(t0, D, RMCL_F32, G + y.pool_b, B, D, s));
  }
  {
    GemmArgs g = ga(t0, pool32 + y.pool_w, dcls, B, D, D, D, D, D);                       // dcls = dpre Wp
    RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 0, s));
  }
  return 0;
}

int rmcl_im2patch_f32(const float* img, float* patches, int B, int C, int Hh, int Ww, int ps, int to_image, void* stream) {
  return rmcl_im2patch(img, patches, B, C, Hh, Ww, ps, to_image, (hipStream_t)stream);
}
int64_t rmcl_ln_fold_elems(const rmcl_dims* d, int which) {
  const int64_t rows = 3 * (int64_t)d->D + d->mlp;
  return which == 0 ? (int64_t)d->layers * rows * d->D : (int64_t)d->layers * 2 * rows;
}
int rmcl_ln_fold(const rmcl_dims* d, const float* params32, void* wf, float* sc, void* stream) {
  RMCL_REQUIRE(d && params32 && wf && sc, "ln_fold: NULL argument");
  rmcl_layout y;
  rmcl_param_layout(d, &y);
  return rmcl_ln_fold_launch(params32, y.layer0, y.layer_stride, d->layers, y.ln1_w, y.ln1_b, y.qkv_w, y.qkv_b, y.ln2_w, y.ln2_b, y.fc1_w,
                             y.fc1_b, d->D, d->mlp, (unsigned short*)wf, sc, (hipStream_t)stream);
}
int rmcl_weight_transpose_bf16(const rmcl_dims* d, const void* params_lp, void* params_lpT, void* stream) {
  RMCL_REQUIRE(d && params_lp && params_lpT, "weight_transpose: NULL argument");
  rmcl_layout y;
  rmcl_param_layout(d, &y);
  const long offs[4] = {y.qkv_w, y.proj_w, y.fc1_w, y.fc2_w};
  const int rows[4] = {3 * d->D, d->D, d->mlp, d->D}, cols[4] = {d->D, d->D, d->D, d->mlp};
  return rmcl_weight_transpose((const unsigned short*)params_lp, (unsigned short*)params_lpT, y.layer0, y.layer_stride, d->layers, offs, rows, cols,
                               (hipStream_t)stream);
}
int rmcl_linear_rowstat(const void* A, const void* W, const float* bias, const float* residual, float* out, void* out_bf16, float* part,
                        int M, int N, int K, void* stream) {
  return rmcl_linear_rowstat_c(A, W, bias, residual, nullptr, out, out_bf16, part, M, N, K, stream);
}
int rmcl_linear_lnfold(const void* xb, const void* wf, const float* s, const float* c, const float* part, int nparts, void* out,
                       void* preact, int M, int N, int K, int gelu, float eps, float* mean, float* rstd, void* stream) {
  return rmcl_linear_lnfold_c(xb, wf, s, c, part, nparts, nullptr, out, preact, M, N, K, gelu, eps, mean, rstd, stream);
}
int rmcl_linear_rowstat_c(const void* A, const void* W, const float* bias, const float* residual, const float* center, float* out,
                          void* out_bf16, float* part, int M, int N, int K, void* stream) {
  RMCL_REQUIRE(A && W && bias && residual && out && out_bf16 && part, "linear_rowstat: NULL argument");
  GemmArgs g = ga(A, W, out, M, N, K, K, K, N);
  g.epi = EPI_BIAS | EPI_RESIDUAL | EPI_ROWSTAT; g.bias = bias; g.aux = residual; g.ld_aux = N; g.C2 = out_bf16;
  g.ln_part = part; g.ln_nparts = 4 * (N / 192); g.ln_center = center;
  RMCL_REQUIRE(N % 192 == 0 && rmcl_gemm_routes_to_tile192(g, 1, 1), "linear_rowstat: shape does not run on the 192-row tile kernels");
  return rmcl_launch_gemm(g, RMCL_BF16, RMCL_F32, 1, 1, 0, (hipStream_t)stream);
}
int rmcl_linear_lnfold_c(const void* xb, const void* wf, const float* s, const float* c, const float* part, int nparts, const float* center,
                         void* out, void* preact, int M, int N, int K, int gelu, float eps, float* mean, float* rstd, void* stream) {
  RMCL_REQUIRE(xb && wf && s && c && part && out, "linear_lnfold: NULL argument");
  GemmArgs g = ga(xb, wf, out, M, N, K, K, K, N);
  g.epi = EPI_LNFOLD | (gelu ? EPI_GELU : 0) | (preact ? EPI_SAVE_PREACT : 0); g.C2 = preact;
  g.ln_s = s; g.ln_c = c; g.ln_part = const_cast<float*>(part); g.ln_nparts = nparts; g.ln_cols = K; g.ln_eps = eps;
  g.ln_mean = mean; g.ln_rstd = rstd; g.ln_center = center;
  RMCL_REQUIRE(rmcl_gemm_routes_to_tile192(g, 1, 1), "linear_lnfold: shape does not run on the 192-row tile kernels");
  return rmcl_launch_gemm(g, RMCL_BF16, RMCL_BF16, 1, 1, 0, (hipStream_t)stream);
}
int rmcl_patch_select(const float* img, int B, int C, int Hh, int Ww, int ps, int32_t* sel, int32_t* counts, int32_t* hw, void* stream) {
  RMCL_REQUIRE(img && sel && counts && hw, "patch_select: NULL argument");
  return rmcl_patch_select(img, B, C, Hh, Ww, ps, sel, counts, hw, (hipStream_t)stream);
}
int rmcl_im2patch_sel(float* img, float* patches, const int32_t* sel, const int32_t* counts, int sel_ld, int B, int n, int C, int Hh,
                      int Ww, int ps, int to_image, void* stream) {
  RMCL_REQUIRE(img && patches && sel && counts && n > 0, "im2patch_sel: NULL argument");
  return rmcl_im2patch_sel(img, patches, sel, counts, sel_ld, B, n, C, Hh, Ww, ps, to_image, (hipStream_t)stream);
}
int rmcl_image_u8_to_patches(const uint8_t* img, const int32_t* sizes, const int32_t* sel, const int32_t* counts, int sel_ld, int B, int n,
                             int Hmax, int Wmax, int patch_size, const float* lut, float* patches, void* stream) {
  RMCL_REQUIRE(img && sizes && lut && patches, "image_u8_to_patches: NULL argument");
  RMCL_REQUIRE(patch_size == 32, "image_u8_to_patches: patch size 32 (ViLT-B/32)");
  RMCL_REQUIRE((sel == nullptr) == (counts == nullptr), "image_u8_to_patches: sel and counts go together");
  return rmcl_u8_to_patches(img, sizes, sel, counts, sel_ld, B, n, Hmax, Wmax, lut, patches, (hipStream_t)stream);
}
int rmcl_image_resize_u8(const uint8_t* src, const int32_t* src_sizes, int B, int Hs, int Ws, const int32_t* dst_sizes, int Hd, int Wd,
                         const int32_t* hbounds, const int32_t* hk, int ksh, const int32_t* vbounds, const int32_t* vk, int ksv, uint8_t* tmp,
                         uint8_t* dst, void* stream) {
  RMCL_REQUIRE(src && src_sizes && dst_sizes && hbounds && hk && vbounds && vk && tmp && dst, "image_resize_u8: NULL argument");
  return rmcl_resize_u8(src, src_sizes, B, Hs, Ws, dst_sizes, Hd, Wd, hbounds, hk, ksh, vbounds, vk, ksv, tmp, dst, (hipStream_t)stream);
}
int rmcl_shard_sum(const void* pieces, int dtype, int n_pieces, int64_t piece_elems, float* out32, void* out_wire, void* stream) {
  RMCL_REQUIRE(pieces && (out32 || out_wire), "shard_sum: NULL argument");
  RMCL_REQUIRE(dtype == RMCL_F32 || dtype == RMCL_BF16, "shard_sum: dtype");
  return rmcl_k_shard_sum(pieces, dtype, n_pieces, piece_elems, out32, out_wire, (hipStream_t)stream);
}
int rmcl_add_cast_f32(const float* a, const float* d1, const float* d2, void* out, int dtype, int64_t n, void* stream) {
  RMCL_REQUIRE(a && out, "add_cast: NULL argument");
  return rmcl_k_add_cast(a, d1, d2, out, dtype, n, (hipStream_t)stream);
}
int64_t rmcl_infonce_ws_bytes(int B, int64_t Kq) { return rmcl_infonce_workspace_bytes(B, Kq) + 256; }
int rmcl_infonce_f32(const float* q, const float* k, const float* queue, int B, int proj, int64_t Kq, float temperature,
                     float grad_scale, float* dq, float* rows_out, float* loss_sum, void* workspace, void* stream) {
  RMCL_REQUIRE(q && k && queue && rows_out && workspace, "infonce: NULL argument");
  return rmcl_infonce(q, k, queue, B, proj, Kq, temperature, grad_scale, dq, rows_out, loss_sum, workspace, (hipStream_t)stream);
}
int rmcl_infonce_split_bf16(const float* q, const float* k, const float* queue, int B, int proj, int64_t Kq, float temperature,
                            float grad_scale, float* dq, float* rows_out, float* loss_sum, void* workspace, int with_metrics, void* stream) {
  RMCL_REQUIRE(q && k && queue && rows_out && workspace, "infonce: NULL argument");
  return rmcl_infonce(q, k, queue, B, proj, Kq, temperature, grad_scale, dq, rows_out, loss_sum, workspace, (hipStream_t)stream,
                      with_metrics ? 1 : 2);
}
int rmcl_pgd_step(const void* grad, int dtype, float* delta, uint32_t* amax_scratch, int B, int64_t per_sample, float lr,
                  float eps, void* stream) {
  RMCL_REQUIRE(grad && delta && amax_scratch, "pgd_step: NULL argument");
  return rmcl_pgd_update(grad, dtype, delta, amax_scratch, B, per_sample, lr, eps, (hipStream_t)stream);
}
int rmcl_pgd_step_fused(const void* grad, int dtype, float* delta, uint32_t* amax_scratch, int B, int64_t per_sample, float lr,
                        float eps, const float* base, void* operand, int operand_dtype, int flags, void* stream) {
  RMCL_REQUIRE(grad && delta && amax_scratch, "pgd_step_fused: NULL argument");
  RMCL_REQUIRE((flags & ~(RMCL_PGD_DELTA_ZERO | RMCL_PGD_SUM_PREV)) == 0, "pgd_step_fused: unknown flag");
  return rmcl_pgd_update_fused(grad, dtype, delta, amax_scratch, B, per_sample, lr, eps, base, operand, operand_dtype, flags, (hipStream_t)stream);
}
int rmcl_delta_channel_norm(const float* delta, float* out, int64_t rows, int C, int pp, void* stream) {
  return rmcl_delta_chan_norm(delta, out, rows, C, pp, (hipStream_t)stream);
}
int rmcl_ema_f32(float* k, const float* q, void* k_lp, float m, int64_t n, void* stream) {
  RMCL_REQUIRE(k && q, "ema: NULL argument");
  return rmcl_ema(k, q, k_lp, m, n, (hipStream_t)stream);
}
int rmcl_enqueue_f32(float* queue, const float* keys, int n, int proj, int64_t Kq, int64_t ptr, void* stream) {
  RMCL_REQUIRE(queue && keys, "enqueue: NULL argument");
  return rmcl_enqueue(queue, keys, n, proj, Kq, ptr, (hipStream_t)stream);
}
int rmcl_cast_f32(const float* in, void* out, int dtype, int64_t n, void* stream) {
  return rmcl_cast(in, out, dtype, n, (hipStream_t)stream);
}
int rmcl_adamw_f32(float* p, const float* g, float* m, float* v, void* p_lp, const int64_t* seg_end, const float* seg_lr_mult,
                   const float* seg_wd, int nseg, float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                   int64_t n, void* stream) {
  RMCL_REQUIRE(p && g && m && v && seg_end && seg_lr_mult && seg_wd, "adamw: NULL argument");
  return rmcl_adamw(p, g, m, v, p_lp, (const long*)seg_end, seg_lr_mult, seg_wd, nseg, lr, beta1, beta2, eps, step, grad_scale, n,
                    (hipStream_t)stream);
}
int rmcl_ipot_f32(const float* cost, const int32_t* txt_valid, const int32_t* img_valid, float* T, int B, int Lt, int Li,
                  int ld, float beta, int iters, void* stream) {
  RMCL_REQUIRE(cost && txt_valid && img_valid && T, "ipot: NULL argument");
  return rmcl_ipot(cost, txt_valid, img_valid, T, B, Lt, Li, ld, beta, iters, (hipStream_t)stream);
}

int rmcl_gemm_batched(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int ldc, float alpha,
                      int nbatch, int64_t strideA, int64_t strideB, int64_t strideC, int dt_in, int dt_out, int a_kc, int b_kc,
                      void* stream) {
  RMCL_REQUIRE(A && B && C && nbatch >= 1, "gemm_batched: bad argument");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = alpha; g.splitk = 1; g.nb1 = nbatch; g.nb2 = 1; g.sA1 = strideA; g.sB1 = strideB; g.sC1 = strideC;
  return rmcl_launch_gemm_exact(g, dt_in, dt_out, a_kc, b_kc, (hipStream_t)stream);
}
int rmcl_l2norm_rows_fwd(const float* x, float* y, float* norms, int rows, int D, float eps, void* stream) {
  RMCL_REQUIRE(x && y && norms, "l2norm_rows_fwd: NULL argument");
  return rmcl_l2norm_fwd(x, y, norms, rows, D, eps, (hipStream_t)stream);
}
int rmcl_l2norm_rows_bwd(const float* dy, const float* y, const float* norms, float* dx, int rows, int D, void* stream) {
  RMCL_REQUIRE(dy && y && norms && dx, "l2norm_rows_bwd: NULL argument");
  return rmcl_l2norm_bwd(dy, y, norms, dx, rows, D, (hipStream_t)stream);
}
int rmcl_wpa_cost_finish(float* cost, const int32_t* txt_valid, const int32_t* img_valid, int B, int Lt, int Li, int ld, void* stream) {
  RMCL_REQUIRE(cost && txt_valid && img_valid, "wpa_cost_finish: NULL argument");
  return rmcl_cost_finish(cost, txt_valid, img_valid, B, Lt, Li, ld, (hipStream_t)stream);
}
int rmcl_wpa_distance(const float* cost, const float* T, const float* w, float* dist, float* dsim, int B, int Lt, int Li, int ld,
                      void* stream) {
  RMCL_REQUIRE(cost && T && dist, "wpa_distance: NULL argument");
  return rmcl_wpa_dist(cost, T, w, dist, dsim, B, Lt, Li, ld, (hipStream_t)stream);
}
int rmcl_itm_fwd(const float* cls, const float* W, const float* bias, const int32_t* labels, float* logits, float* dlogits,
                 float* loss_sum, int B, int D, float grad_scale, void* stream) {
  RMCL_REQUIRE(cls && W && bias && labels && logits, "itm_fwd: NULL argument");
  return rmcl_itm_head_fwd(cls, W, bias, labels, logits, dlogits, loss_sum, B, D, grad_scale, (hipStream_t)stream);
}
int rmcl_itm_bwd(const float* dlogits, const float* cls, const float* W, float* dcls, float* dW, float* db, int B, int D,
                 float scale, void* stream) {
  RMCL_REQUIRE(dlogits && cls && W && dcls, "itm_bwd: NULL argument");
  return rmcl_itm_head_bwd(dlogits, cls, W, dcls, dW, db, B, D, scale, (hipStream_t)stream);
}

int rmcl_gemm(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M, int N, int K,
              int64_t lda, int64_t ldb, int ldc, int ld_aux, float alpha, int epi, int splitk, int dt_in, int dt_out, int a_kc,
              int b_kc, int exact, void* stream) {
  RMCL_REQUIRE(A && B && C, "gemm: NULL operand");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.C2 = C2; g.bias = bias; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.ld_aux = ld_aux;
  g.alpha = alpha; g.epi = epi; g.splitk = splitk < 1 ? 1 : splitk; g.nb1 = 1; g.nb2 = 1;
  return rmcl_launch_gemm(g, dt_in, dt_out, a_kc, b_kc, exact, (hipStream_t)stream);
}
int rmcl_gemm_chain(const rmcl_chain_stage* st, int n, int M, uint32_t* tickets, uint32_t epoch, int flags, int32_t* xcc, int64_t* stamps,
                    int stamp_wg, void* stream) {
  RMCL_REQUIRE(st && n >= 1 && n <= 4 && tickets && epoch >= 1, "gemm_chain: bad argument");
  ChainArgs a{};
  a.n = n; a.ticket = tickets; a.target = 4u * epoch; a.flags = flags; a.xcc = xcc; a.stamps = (long long*)stamps; a.stamp_wg = stamp_wg;
  for (int i = 0; i < n; ++i) {
    const rmcl_chain_stage& q = st[i];
    GemmArgs g = ga(q.A, q.W, q.out, M, q.N, q.K, q.K, q.K, q.N);
    g.epi = q.epi; g.bias = q.bias; g.aux = q.residual; g.ld_aux = q.N; g.C2 = q.out2;
    g.ln_part = q.part; g.ln_nparts = (q.epi & EPI_ROWSTAT) ? 4 * (q.N / 192) : q.nparts; g.ln_center = q.center;
    g.ln_s = q.ln_s; g.ln_c = q.ln_c; g.ln_cols = q.K; g.ln_eps = q.ln_eps; g.ln_mean = q.mean; g.ln_rstd = q.rstd;
    a.g[i] = g;
  }
  return rmcl_launch_gemm_chain(a, (hipStream_t)stream);
}
int rmcl_gemm_route(int M, int N, int K, int epi, int dt_out, int a_kc, int b_kc) {
  GemmArgs g{};
  g.M = M; g.N = N; g.K = K; g.epi = epi; g.splitk = 1; g.nb1 = 1; g.nb2 = 1; g.alpha = 1.0f;
  g.lda = a_kc ? K : M; g.ldb = b_kc ? K : N; g.ldc = N;
  // (the pointer checks of the folded epilogues: any non-NULL value; nothing is dereferenced here)
  static float dummy;
  g.ln_s = g.ln_c = &dummy; g.ln_part = &dummy; g.C2 = &dummy; g.ln_cols = K;
  g.ln_nparts = (epi & EPI_ROWSTAT) ? 4 * (N / 192) : 4 * (K / 192);
  return rmcl_gemm_route_code(g, dt_out, a_kc, b_kc);
}
int rmcl_gemm_kblk(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M, int N, int K, int ldc,
                   int ld_aux, int epi, int dt_out, int kblk, void* stream) {
  RMCL_REQUIRE(A && B && C, "gemm_kblk: NULL operand");
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C; g.C2 = C2; g.bias = bias; g.aux = aux;
  g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = ldc; g.ld_aux = ld_aux;
  g.alpha = 1.0f; g.epi = epi; g.splitk = 1; g.nb1 = 1; g.nb2 = 1; g.kblk = kblk & 3;
  RMCL_REQUIRE(rmcl_gemm_dp_supported(g, 1, 1) && !((epi & EPI_RESIDUAL) && dt_out != RMCL_F32), "gemm_kblk: shape / epilogue not taken by gemm_dp");
  return rmcl_launch_gemm_dp(g, dt_out, (hipStream_t)stream);
}
int rmcl_layernorm_fwd(const float* x, const float* w, const float* b, float eps, void* y, int dt_out, float* mean, float* rstd,
                       int M, int D, int relu, void* stream) {
  return rmcl_ln_fwd(x, D, w, b, eps, y, D, dt_out, mean, rstd, M, D, relu, (hipStream_t)stream);
}
int rmcl_layernorm_bwd(const void* dy, int dt_dy, const float* x, const float* mean, const float* rstd, const float* w,
                       const float* b, float* dx, int add, float* dgamma, float* dbeta, int M, int D, int relu, void* stream) {
  return rmcl_ln_bwd(dy, D, dt_dy, x, D, mean, rstd, w, b, dx, D, add, dgamma, dbeta, M, D, relu, (hipStream_t)stream);
}
int64_t rmcl_attention_scratch_elems(int B, int H, int N) { return (int64_t)B * H * N * ((N + 7) / 8 * 8); }
int rmcl_attention_fwd(const void* qkv, const int32_t* mask, void* out, void* probs, float* scores, int B, int N, int H, int dtype,
                       int exact, void* stream) {
  RMCL_REQUIRE(qkv && mask && out && probs && scores, "attention_fwd: NULL argument");
  return rmcl_attention_fwd_impl(qkv, mask, out, probs, scores, B, N, H, dtype, exact, (hipStream_t)stream);
}
int rmcl_attention_bwd(const void* qkv, const int32_t* mask, const void* probs, const void* dout, const void* out, void* dqkv,
                       float* scores, void* dscores, int B, int N, int H, int dtype, int exact, void* stream) {
  RMCL_REQUIRE(qkv && mask && probs && dout && dqkv && scores && dscores, "attention_bwd: NULL argument");
  return rmcl_attention_bwd_impl(qkv, mask, probs, dout, out, dqkv, scores, dscores, B, N, H, dtype, exact, (hipStream_t)stream);
}

}  // extern "C"
