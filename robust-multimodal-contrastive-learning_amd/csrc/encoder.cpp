// Host-side orchestration of one ViLT encoder pass (forward, data-gradient backward, full backward)
// as a fixed sequence of kernel launches on one HIP stream: no allocation, no synchronisation, so
// the whole pass can be captured into a hipGraph by the caller.
//
// Reference being replaced: ViLTransformerSS.infer / infer_k (vilt/modules/vilt_module.py:275-418),
// VisionTransformer.visual_embed dense case (vision_transformer.py:559-677), Block / Attention / Mlp
// (vision_transformer.py:262-375) and their autograd backward.
#include "rmcl_common.h"
#include "kernels.h"
#include "../../include/rmcl.h"

namespace {

struct Bump {
  char* base;
  size_t off = 0;
  explicit Bump(void* b) : base(reinterpret_cast<char*>(b)) {}
  template <typename T> T* take(size_t n) {
    off = (off + 255) & ~(size_t)255;
    T* p = base ? reinterpret_cast<T*>(base + off) : nullptr;
    off += n * sizeof(T);
    return p;
  }
  void* take_bytes(size_t n) { return take<char>(n); }
};

inline size_t esz(int dt) { return dt == RMCL_F32 ? 4 : 2; }
inline int ldp_of(int N) { return (N + 7) / 8 * 8; }

#define SLAB_FLOATS(d) ((size_t)4 * (size_t)std::max((d).mlp, (d).patch_k) * (size_t)(d).D)

struct LayerStash {
  float *x_in, *x_mid, *mean1, *rstd1, *mean2, *rstd2;
  void *qkv, *probs, *u, *ao;      // DATA+ (ao: the attention output, for the one-kernel attention backward's delta)
  void *ln1, *ln2, *h;             // FULL (workspace-aliased otherwise)
};

struct Stash {
  LayerStash layer[64];
  float *x_final, *meanF, *rstdF;
  float *text_e, *text_mean, *text_rstd;
};

struct Work {
  float *scores;      // [B,H,N,ldp] f32 (S and dP)
  float *pe;          // [B*P, D] f32
  void *ln, *ao, *h, *qkv, *probs, *u;   // INFER-mode per-layer temporaries
  float *x_a, *x_b, *x_mid, *stat;       // INFER-mode residual ping-pong + stats
  // backward
  float *dx, *dln, *dxn_full, *de;
  void *dxT, *du, *dqkv, *dao, *dS, *dpe;
  void *dxT2, *du2, *dqkv2;   // second copies: weight-gradient GEMMs read them on the side stream
  void *xb_in, *xb_mid;   // LayerNorm fold: bf16 copies of the residual stream (block input / after attention)
  float *part_in, *part_mid;   // and the per-row partial sums [M][16][2] the producer GEMMs emit
  void* dxT3;         // third dx copy (grouped weight gradients: one launch per layer reads both of the layer's dx copies)
  float* ln_rep;      // LayerNorm dgamma/dbeta replica slots (2 * layers + 1), summed into the arena by the grouped kernel
  float* slab;        // split-K partial slabs of the weight-gradient GEMMs
};

// Stash carve.  INFER: nothing is stashed (everything aliases workspace).
size_t carve_stash(const rmcl_dims& d, int mode, void* base, Stash* st) {
  Bump b(base);
  const size_t M = (size_t)d.B * (d.L + 1 + d.P), D = d.D, N = d.L + 1 + d.P;
  const size_t e = esz(d.dtype);
  if (mode == RMCL_MODE_INFER) return 0;
  for (int l = 0; l < d.layers; ++l) {
    LayerStash ls{};
    ls.x_in = b.take<float>(M * D);
    ls.x_mid = b.take<float>(M * D);
    ls.mean1 = b.take<float>(M); ls.rstd1 = b.take<float>(M);
    ls.mean2 = b.take<float>(M); ls.rstd2 = b.take<float>(M);
    ls.qkv = b.take_bytes(M * 3 * D * e);
    ls.probs = b.take_bytes((size_t)d.B * d.H * N * ldp_of((int)N) * e);
    ls.u = b.take_bytes(M * d.mlp * e);
    ls.ao = b.take_bytes(M * D * e);
    if (mode == RMCL_MODE_FULL) {
      ls.ln1 = b.take_bytes(M * D * e);
      ls.ln2 = b.take_bytes(M * D * e);
      ls.h = b.take_bytes(M * d.mlp * e);
    }
    if (st) st->layer[l] = ls;
  }
  float* xf = b.take<float>(M * D);
  float* mf = b.take<float>(M);
  float* rf = b.take<float>(M);
  float *te = nullptr, *tm = nullptr, *tr = nullptr;
  {
    te = b.take<float>((size_t)d.B * d.L * D);
    tm = b.take<float>((size_t)d.B * d.L);
    tr = b.take<float>((size_t)d.B * d.L);
  }
  if (st) { st->x_final = xf; st->meanF = mf; st->rstdF = rf; st->text_e = te; st->text_mean = tm; st->text_rstd = tr; }
  return b.off;
}

size_t carve_work(const rmcl_dims& d, void* base, Work* w) {
  Bump b(base);
  const size_t M = (size_t)d.B * (d.L + 1 + d.P), D = d.D, N = d.L + 1 + d.P;
  const size_t e = esz(d.dtype);
  const size_t zn = (size_t)d.B * d.H * N * ldp_of((int)N);
  Work k{};
  k.scores = b.take<float>(zn);
  k.pe = b.take<float>((size_t)d.B * d.P * D);
  k.ln = b.take_bytes(M * D * e);
  k.ao = b.take_bytes(M * D * e);
  k.h = b.take_bytes(M * d.mlp * e);
  k.qkv = b.take_bytes(M * 3 * D * e);
  k.probs = b.take_bytes(zn * e);
  k.u = b.take_bytes(M * d.mlp * e);
  k.x_a = b.take<float>(M * D);
  k.x_b = b.take<float>(M * D);
  k.x_mid = b.take<float>(M * D);
  k.stat = b.take<float>(4 * M);
  k.dx = b.take<float>(M * D);
  k.dln = b.take<float>(M * D);
  k.dxn_full = b.take<float>(M * D);
  k.de = b.take<float>((size_t)d.B * d.L * D);
  k.dxT = b.take_bytes(M * D * e);
  k.du = b.take_bytes(M * d.mlp * e);
  k.dqkv = b.take_bytes(M * 3 * D * e);
  k.dao = b.take_bytes(M * D * e);
  k.dS = b.take_bytes(zn * e);
  k.dpe = b.take_bytes((size_t)d.B * d.P * D * e);
  k.slab = b.take<float>(SLAB_FLOATS(d));
  k.dxT2 = b.take_bytes(M * D * e);
  k.du2 = b.take_bytes(M * d.mlp * e);
  k.dqkv2 = b.take_bytes(M * 3 * D * e);
  k.dxT3 = b.take_bytes(M * D * e);
  k.xb_in = b.take_bytes(M * D * 2);
  k.xb_mid = b.take_bytes(M * D * 2);
  k.part_in = b.take<float>(M * 2 * 4 * (D / 192 + 1));
  k.part_mid = b.take<float>(M * 2 * 4 * (D / 192 + 1));
  k.ln_rep = b.take<float>((size_t)(2 * d.layers + 1) * RMCL_LN_REP_FLOATS);
  if (w) *w = k;
  return b.off;
}

int check_dims(const rmcl_dims* d) {
  RMCL_REQUIRE(d != nullptr, "dims is NULL");
  RMCL_REQUIRE(d->B > 0 && d->L > 0 && d->P > 0 && d->layers > 0 && d->layers <= 64, "bad dims");
  RMCL_REQUIRE(d->D == d->H * 64, "head dim must be 64");
  RMCL_REQUIRE(d->D % 64 == 0 && d->D <= 1024 && d->mlp % 64 == 0 && d->patch_k % 64 == 0, "D/mlp/patch_k must be multiples of 64 (D <= 1024)");
  RMCL_REQUIRE(d->dtype == RMCL_F32 || d->dtype == RMCL_BF16, "bad dtype");
  RMCL_REQUIRE(d->L + 1 + d->P <= 512, "sequence too long (N <= 512)");
  return 0;
}

GemmArgs gemm_args(const void* A, const void* B, void* C, int M, int N, int K, long lda, long ldb, int ldc) {
  GemmArgs g{};
  g.A = A; g.B = B; g.C = C;
  g.M = M; g.N = N; g.K = K; g.lda = lda; g.ldb = ldb; g.ldc = ldc;
  g.alpha = 1.0f; g.epi = 0; g.splitk = 1; g.nb1 = 1; g.nb2 = 1;
  return g;
}

struct Ctx {
  const rmcl_dims& d;
  const float* P32;
  const void* Plp;
  rmcl_layout lay;
  hipStream_t s;
  int dt;
  // weight matrix (GEMM operand type) and fp32 vector accessors
  const void* W(int64_t off) const {
    return d.dtype == RMCL_BF16 ? (const void*)((const bf16_t*)Plp + off) : (const void*)(P32 + off);
  }
  const float* V(int64_t off) const { return P32 + off; }
  int64_t L(int l, int64_t rel) const { return lay.layer0 + (int64_t)l * lay.layer_stride + rel; }
};

int gemm(const Ctx& c, const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc) {
  return rmcl_launch_gemm(g, dt_in, dt_out, a_kc, b_kc, c.d.exact, c.s);
}

// split-K factor for weight-gradient GEMMs (reduction over tokens): fill ~2 waves of 256 CUs.
int dw_splitk(int Mout, int Nout, int K) {
  const int tiles = cdiv(Mout, 128) * cdiv(Nout, 128);
  int sk = std::max(1, 512 / tiles);
  sk = std::min(sk, std::max(1, K / 256));
  return std::min(sk, 64);
}

// dW[Nout, Kin] += dY[tokens, Nout]^T @ X[tokens, Kin]   (fp32 atomics into the gradient arena)
int gemm_dw(const Ctx& c, const void* dY, long lddy, const void* X, long ldx, float* dW, int Nout, int Kin, int tokens, int dt_in,
            float* slab, size_t slab_floats) {
  GemmArgs g = gemm_args(dY, X, dW, Nout, Kin, tokens, lddy, ldx, Kin);
  if (!c.d.exact && dt_in == RMCL_BF16 && slab) {
    // bf16 MFMA path: split-K over tokens into fp32 slabs + ordered reduce (no float atomics)
    GemmArgs probe = g;
    const int cfg = rmcl_gemm_fast_get_cfg();                       // tune cfg 1..4 pins the 128x128 kernel (tests A/B the two paths)
    if ((cfg < 1 || cfg > 4) && rmcl_gemm_st_supported(probe, 0, 0) && tokens >= 1024) {
      // 192x192 ping-pong tiles, one (tile, K-slice) work item per CU
      const int tiles = (Nout / 192) * (Kin / 192);
      int sk = std::max(1, 256 / tiles);
      sk = std::min<int>(sk, std::max<size_t>(1, slab_floats / ((size_t)Nout * Kin)));
      g.splitk = sk; g.tag = GEMM_TAG_DW;
      return rmcl_launch_gemm_st_slab(g, slab, dW, c.s);
    }
    if (rmcl_gemm_fast_supported(probe, dt_in, RMCL_F32, 0, 0)) {
      const int tiles = (Nout / 128) * (Kin / 128);
      int sk = std::max(1, std::min(8, (256 + tiles - 1) / tiles));
      sk = std::min<int>(sk, std::max<size_t>(1, slab_floats / ((size_t)Nout * Kin)));
      sk = std::min(sk, std::max(1, tokens / 64));
      g.splitk = sk; g.tag = GEMM_TAG_DW;
      return rmcl_launch_gemm_fast_slab(g, slab, dW, c.s);
    }
  }
  g.epi = EPI_ATOMIC; g.tag = GEMM_TAG_DW;
  g.splitk = dw_splitk(Nout, Kin, tokens);
  if (g.splitk == 1) g.epi = EPI_ACCUM;
  return gemm(c, g, dt_in, RMCL_F32, 0, 0);
}

// ---- side stream for the weight-gradient GEMMs (fills the tile-quantisation tails of the dX chain) ----
hipStream_t g_side = nullptr;
hipEvent_t g_ev[128];
int g_nev = 0;
int ensure_events() {
  while (g_nev < 128) {
    hipError_t e = hipEventCreateWithFlags(&g_ev[g_nev], hipEventDisableTiming);
    if (e != hipSuccess) { rmcl_set_error(hipGetErrorString(e)); return (int)e; }
    ++g_nev;
  }
  return 0;
}
#define HIP_TRY(expr) do { hipError_t _e = (expr); if (_e != hipSuccess) { rmcl_set_error(hipGetErrorString(_e)); return (int)_e; } } while (0)

}  // namespace

// stash prefetch one layer ahead of the backward chain (rmcl_tune_set key 12: bit 0 = pre-activation u, bit 1 = qkv + attention output,
// bit 2 = the two residual-stream stashes; key 13: workgroups of the touch kernel).  Runs on the stream set by rmcl_set_prefetch_stream.
int g_prefetch_mask = 0, g_prefetch_wgs = 8;
static hipStream_t g_pref = nullptr;
static hipEvent_t g_pev[64];
static int g_npev = 0;
extern "C" int rmcl_set_prefetch_stream(void* stream) { g_pref = (hipStream_t)stream; return 0; }
bool g_lnfold_centred = true;        // rmcl_tune_set key 11: 0 = the uncentred LayerNorm fold of round 2 (A/B, precision tests)
bool rmcl_lnfold_centred() { return g_lnfold_centred; }
bool g_dw_grouped = true;            // rmcl_tune_set key 3: 0 selects the per-GEMM weight-gradient path (A/B and parity tests)

extern "C" int rmcl_set_side_stream(void* stream) { g_side = (hipStream_t)stream; return 0; }

// ---- gradient-ready events of the last weight-gradient backward (overlapped data-parallel all-reduce) ----
static int g_grad_layers = 0;      // 0: no full backward has run
static bool g_grad_side = false;   // the weight gradients of that backward ran on the side stream
extern "C" int rmcl_grad_ready_wait(int layer, void* stream) {
  RMCL_REQUIRE(g_grad_layers > 0, "grad_ready_wait: no weight-gradient backward has been enqueued");
  RMCL_REQUIRE(layer >= 0 && layer < g_grad_layers && layer < 32, "grad_ready_wait: bad layer");
  HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, g_ev[(3 * 32 + layer) & 127], 0));                 // dX chain + LayerNorm grads
  if (g_grad_side) HIP_TRY(hipStreamWaitEvent((hipStream_t)stream, g_ev[(2 * 32 + layer) & 127], 0));  // weight-gradient GEMMs
  return 0;
}

int rmcl_attention_fwd_impl(const void* qkv, const int* mask, void* out, void* probs, float* scores, int B, int N, int H, int dt,
                            int exact, hipStream_t s) {
  // bf16 fast path: fused kernel; `probs` then holds only the per-row log-sum-exp (fp32 [B,H,NKP])
  if (dt == RMCL_BF16 && !exact && N <= 256) return rmcl_attn_fused_fwd(qkv, mask, out, (float*)probs, B, N, H, s);
  const int D = H * 64, ldp = ldp_of(N);
  const size_t e = esz(dt);
  // S = 0.125 * Q K^T   (batched over (b,h); Q/K rows are 3D apart, heads 64 apart)
  GemmArgs g = gemm_args(qkv, (const char*)qkv + (size_t)D * e, scores, N, N, 64, 3 * D, 3 * D, ldp);
  g.alpha = 0.125f;
  g.nb1 = B; g.nb2 = H;
  g.sA1 = (long)N * 3 * D; g.sA2 = 64; g.sB1 = g.sA1; g.sB2 = 64;
  g.sC1 = (long)H * N * ldp; g.sC2 = (long)N * ldp;
  RMCL_TRY(rmcl_launch_gemm(g, dt, RMCL_F32, 1, 1, exact, s));
  RMCL_TRY(rmcl_softmax_fwd(scores, ldp, mask, probs, ldp, dt, B * H, N, H, s));
  // out[b*N+i, h*64+d] = sum_j P[b,h,i,j] V[b*N+j, 2D + h*64 + d]
  GemmArgs o = gemm_args(probs, (const char*)qkv + (size_t)2 * D * e, out, N, 64, N, ldp, 3 * D, D);
  o.nb1 = B; o.nb2 = H;
  o.sA1 = (long)H * N * ldp; o.sA2 = (long)N * ldp; o.sB1 = (long)N * 3 * D; o.sB2 = 64;
  o.sC1 = (long)N * D; o.sC2 = 64;
  RMCL_TRY(rmcl_launch_gemm(o, dt, dt, 1, 0, exact, s));
  return 0;
}

int rmcl_attention_bwd_impl(const void* qkv, const int* mask, const void* probs, const void* dout, const void* out, void* dqkv,
                            float* scores, void* dS, int B, int N, int H, int dt, int exact, hipStream_t s) {
  // bf16 fast path: probs = saved log-sum-exp, scores = scratch for delta (both fp32 [B,H,NKP]); with the forward's output
  // `out` (stashed in DATA and FULL mode) ONE kernel produces dQ, dK and dV
  if (dt == RMCL_BF16 && !exact && N <= 256)
    return rmcl_attn_fused_bwd(qkv, mask, dout, out, (const float*)probs, scores, dqkv, B, N, H, s);
  const int D = H * 64, ldp = ldp_of(N);
  const size_t e = esz(dt);
  const long sQ1 = (long)N * 3 * D, sP1 = (long)H * N * ldp, sP2 = (long)N * ldp;
  // dP = dO V^T
  GemmArgs g = gemm_args(dout, (const char*)qkv + (size_t)2 * D * e, scores, N, N, 64, D, 3 * D, ldp);
  g.nb1 = B; g.nb2 = H;
  g.sA1 = (long)N * D; g.sA2 = 64; g.sB1 = sQ1; g.sB2 = 64; g.sC1 = sP1; g.sC2 = sP2;
  RMCL_TRY(rmcl_launch_gemm(g, dt, RMCL_F32, 1, 1, exact, s));
  RMCL_TRY(rmcl_softmax_bwd(probs, ldp, scores, ldp, dS, ldp, dt, B * H, N, 0.125f, s));
  // dQ = dS K
  GemmArgs q = gemm_args(dS, (const char*)qkv + (size_t)D * e, dqkv, N, 64, N, ldp, 3 * D, 3 * D);
  q.nb1 = B; q.nb2 = H;
  q.sA1 = sP1; q.sA2 = sP2; q.sB1 = sQ1; q.sB2 = 64; q.sC1 = sQ1; q.sC2 = 64;
  RMCL_TRY(rmcl_launch_gemm(q, dt, dt, 1, 0, exact, s));
  // dK = dS^T Q
  GemmArgs k = gemm_args(dS, qkv, (char*)dqkv + (size_t)D * e, N, 64, N, ldp, 3 * D, 3 * D);
  k.nb1 = B; k.nb2 = H;
  k.sA1 = sP1; k.sA2 = sP2; k.sB1 = sQ1; k.sB2 = 64; k.sC1 = sQ1; k.sC2 = 64;
  RMCL_TRY(rmcl_launch_gemm(k, dt, dt, 0, 0, exact, s));
  // dV = P^T dO
  GemmArgs v = gemm_args(probs, dout, (char*)dqkv + (size_t)2 * D * e, N, 64, N, ldp, D, 3 * D);
  v.nb1 = B; v.nb2 = H;
  v.sA1 = sP1; v.sA2 = sP2; v.sB1 = (long)N * D; v.sB2 = 64; v.sC1 = sQ1; v.sC2 = 64;
  RMCL_TRY(rmcl_launch_gemm(v, dt, dt, 0, 0, exact, s));
  return 0;
}

extern "C" {

void rmcl_param_layout(const rmcl_dims* d, rmcl_layout* o) {
  int64_t off = 0;
  auto take = [&](int64_t n) { int64_t p = off; off += (n + 63) / 64 * 64; return p; };
  const int64_t D = d->D;
  o->word = take((int64_t)d->vocab * D);
  o->pos = take((int64_t)d->L * D);
  o->btype = take(2 * D);
  o->eln_w = take(D); o->eln_b = take(D);
  o->vtype = take(2 * D);
  o->cls = take(D);
  o->pos_img = take((int64_t)((d->Pp > 0 ? d->Pp : d->P) + 1) * D);
  o->patch_w = take(D * d->patch_k);
  o->patch_b = take(D);
  o->layer0 = off;
  {
    int64_t base = off;
    o->ln1_w = take(D) - base; o->ln1_b = take(D) - base;
    o->qkv_w = take(3 * D * D) - base; o->qkv_b = take(3 * D) - base;
    o->proj_w = take(D * D) - base; o->proj_b = take(D) - base;
    o->ln2_w = take(D) - base; o->ln2_b = take(D) - base;
    o->fc1_w = take((int64_t)d->mlp * D) - base; o->fc1_b = take(d->mlp) - base;
    o->fc2_w = take((int64_t)d->mlp * D) - base; o->fc2_b = take(D) - base;
    o->layer_stride = off - base;
    off = base + o->layer_stride * d->layers;
  }
  o->norm_w = take(D); o->norm_b = take(D);
  o->mh0_w = take(D * D); o->mh0_b = take(D);
  o->mh1_w = take(D); o->mh1_b = take(D);
  o->mh3_w = take((int64_t)d->proj * D);
  o->ema_end = off;
  o->pool_w = take(D * D); o->pool_b = take(D);
  o->itm_w = take(2 * D); o->itm_b = take(64);
  o->total = off;
}

int64_t rmcl_stash_bytes(const rmcl_dims* d, int mode) { return (int64_t)carve_stash(*d, mode, nullptr, nullptr) + 256; }
int64_t rmcl_workspace_bytes(const rmcl_dims* d) { return (int64_t)carve_work(*d, nullptr, nullptr) + 256; }

int rmcl_encoder_forward(const rmcl_dims* d, int mode, const float* params32, const void* params_lp, const int64_t* text_ids,
                         const int64_t* text_mask, const void* patches, int32_t* co_mask, void* stash, void* workspace,
                         float* xn, uint32_t drop_seed, float drop_p, const rmcl_ragged* ragged, const rmcl_fold* fold, void* stream) {
  RMCL_TRY(check_dims(d));
  RMCL_REQUIRE(!ragged || (ragged->sel && ragged->counts && ragged->hw && ragged->pos_tok), "encoder_forward: incomplete rmcl_ragged");
  RMCL_REQUIRE(ragged || d->Pp == 0 || d->Pp == d->P, "encoder_forward: P != Pp needs the rmcl_ragged selection");
  RMCL_REQUIRE(params32 && text_ids && text_mask && patches && co_mask && workspace && xn, "encoder_forward: NULL argument");
  RMCL_REQUIRE(d->dtype == RMCL_F32 || params_lp, "encoder_forward: bf16 mode needs the bf16 shadow arena");
  const bool tail_req = (mode & RMCL_MODE_CLS_TAIL) != 0;     // only the cls rows of xn will be read (include/rmcl.h)
  mode &= ~RMCL_MODE_CLS_TAIL;
  RMCL_REQUIRE(mode == RMCL_MODE_INFER || stash, "encoder_forward: stash required unless mode is INFER");
  RMCL_REQUIRE(!tail_req || d->B <= 1024, "encoder_forward: the cls-only tail needs B <= 1024");
  Ctx c{*d, params32, params_lp, {}, (hipStream_t)stream, d->dtype};
  rmcl_param_layout(d, &c.lay);
  const rmcl_layout& y = c.lay;
  Stash st{};
  Work w{};
  carve_stash(*d, mode, stash, &st);
  carve_work(*d, workspace, &w);
  const int B = d->B, L = d->L, P = d->P, N = L + 1 + P, D = d->D, M = B * N, dt = d->dtype;
  const bool keep = mode != RMCL_MODE_INFER, full = mode == RMCL_MODE_FULL;
  hipStream_t s = c.s;
  RMCL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "encoder_forward: dropout probability must be in [0,1)");
  const uint32_t dth = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
  const float dinv = 1.0f / (1.0f - drop_p);
  auto with_drop = [&](GemmArgs& g, int layer, int site, int row_mul = 1) {
    if (dth) { g.epi |= EPI_DROPOUT; g.drop_seed = rmcl_site_seed(drop_seed, layer, site); g.drop_thresh = dth; g.drop_inv_keep = dinv; g.drop_row_mul = row_mul; }
  };

  float* x0 = keep ? st.layer[0].x_in : w.x_a;
  RMCL_TRY(rmcl_text_embed_fwd((const long*)text_ids, c.V(y.word), c.V(y.pos), c.V(y.btype), c.V(y.eln_w), c.V(y.eln_b),
                               c.V(y.vtype), 1e-12f, x0, keep ? st.text_e : nullptr, keep ? st.text_mean : nullptr,
                               keep ? st.text_rstd : nullptr, B, L, N, D, rmcl_site_seed(drop_seed, 0, DROP_SITE_TEXT), dth, dinv, s));
  {
    GemmArgs g = gemm_args(patches, c.W(y.patch_w), w.pe, B * P, D, d->patch_k, d->patch_k, d->patch_k, D);
    g.epi = EPI_BIAS; g.bias = c.V(y.patch_b); g.tag = GEMM_TAG_PATCH;
    RMCL_TRY(gemm(c, g, dt, RMCL_F32, 1, 1));
  }
  if (ragged) {
    // zero-padded batch: per-sample position rows = the table resized to each image's (h, w), gathered at its selected
    // patches (vision_transformer.py:570-600, 645-650); recomputed every pass from the arena of THIS pass (query or momentum)
    RMCL_TRY(rmcl_pos_resize_fwd(c.V(y.pos_img), ragged->sel, ragged->counts, ragged->hw, ragged->sel_ld, ragged->gw, ragged->G0, B, P, D,
                                 ragged->pos_tok, s));
  }
  RMCL_TRY(rmcl_image_assemble_fwd(w.pe, c.V(y.cls), ragged ? ragged->pos_tok : c.V(y.pos_img), c.V(y.vtype) + D, x0, B, P, L, N, D,
                                   rmcl_site_seed(drop_seed, 0, DROP_SITE_IMAGE), dth, dinv, ragged ? 1 : 0, s));
  RMCL_TRY(rmcl_co_mask((const long*)text_mask, patches, dt, co_mask, B, L, P, 3, d->patch_k / 3, s));

  // LayerNorm folded into the consuming GEMMs (gemm.h EPI_LNFOLD / EPI_ROWSTAT): passes that keep no LayerNorm output
  // (INFER, DATA), bf16, dropout off, and only where every GEMM involved runs on the 192-row tile kernels
  // (dropout: the 192-row tile kernels carry the dropout epilogues as separate instantiations since round 4, so the fold stays on)
  bool folded = fold && fold->wf && fold->sc && !full && dt == RMCL_BF16 && !d->exact && D % 192 == 0;
  const int fold_rows = 3 * D + d->mlp, nparts = 4 * (D / 192);
  if (folded) {
    GemmArgs t1 = gemm_args(w.xb_in, fold->wf, w.qkv, M, 3 * D, D, D, D, 3 * D);
    t1.epi = EPI_LNFOLD; t1.ln_s = fold->sc; t1.ln_c = fold->sc; t1.ln_part = w.part_in; t1.ln_nparts = nparts; t1.ln_cols = D;
    GemmArgs t2 = gemm_args(w.xb_mid, fold->wf, w.h, M, d->mlp, D, D, D, d->mlp);
    t2.epi = EPI_LNFOLD | EPI_GELU | EPI_SAVE_PREACT | (dth ? EPI_DROPOUT : 0); t2.C2 = w.u; t2.ln_s = fold->sc; t2.ln_c = fold->sc; t2.ln_part = w.part_mid; t2.ln_nparts = nparts; t2.ln_cols = D;
    GemmArgs t3 = gemm_args(w.ao, c.W(c.L(0, y.proj_w)), w.x_mid, M, D, D, D, D, D);
    t3.epi = EPI_BIAS | EPI_RESIDUAL | EPI_ROWSTAT | (dth ? EPI_DROPOUT : 0); t3.bias = c.V(y.norm_b); t3.aux = w.x_a; t3.ld_aux = D; t3.C2 = w.xb_mid; t3.ln_part = w.part_mid; t3.ln_nparts = nparts;
    GemmArgs t4 = gemm_args(w.h, c.W(c.L(0, y.fc2_w)), w.x_a, M, D, d->mlp, d->mlp, d->mlp, D);
    t4.epi = t3.epi; t4.bias = t3.bias; t4.aux = w.x_mid; t4.ld_aux = D; t4.C2 = w.xb_in; t4.ln_part = w.part_in; t4.ln_nparts = nparts;
    folded = rmcl_gemm_routes_to_tile192(t1, 1, 1) && rmcl_gemm_routes_to_tile192(t2, 1, 1) && rmcl_gemm_routes_to_tile192(t3, 1, 1) &&
             rmcl_gemm_routes_to_tile192(t4, 1, 1);
  }
  const bf16_t* fw = folded ? (const bf16_t*)fold->wf : nullptr;
  // shift-robust form of the fold (gemm.h ln_center): every producer stores bf16(x - c) and the partial sums of (x - c), c = the row mean
  // of the row's PREVIOUS LayerNorm (LN is shift-invariant: exact for any c), so the operand's rounding follows the row's spread instead of
  // its offset.  rmcl_tune_set(11, 0) restores the uncentred round-2 form (A/B, precision tests).
  const bool centred = folded && rmcl_lnfold_centred();
  float* x = x0;
  for (int l = 0; l < d->layers; ++l) {
    LayerStash ls{};
    if (keep) ls = st.layer[l];
    float* x_mid = keep ? ls.x_mid : w.x_mid;
    float* x_out = keep ? (l + 1 < d->layers ? st.layer[l + 1].x_in : st.x_final) : (x == w.x_a ? w.x_b : w.x_a);
    float *m1 = keep ? ls.mean1 : w.stat, *r1 = keep ? ls.rstd1 : w.stat + M;
    float *m2 = keep ? ls.mean2 : w.stat + 2 * M, *r2 = keep ? ls.rstd2 : w.stat + 3 * M;
    void* ln1 = full ? ls.ln1 : w.ln;
    void* ln2 = full ? ls.ln2 : w.ln;
    void* qkv = keep ? ls.qkv : w.qkv;
    void* probs = keep ? ls.probs : w.probs;
    void* ao = keep ? ls.ao : w.ao;
    void* h = full ? ls.h : w.h;
    void* u = keep ? ls.u : nullptr;
    const float* sc_l = folded ? fold->sc + (long)l * 2 * fold_rows : nullptr;        // s[0..rows), c[rows..2 rows)
    // LayerNorm 2's row means of the previous layer = the centre the previous fc2 used for this layer's LayerNorm-1 operand
    const float* m2_prev = l > 0 ? (keep ? st.layer[l - 1].mean2 : w.stat + 2 * M) : nullptr;

    if (folded && l > 0) {
      // qkv = LN1(x) Wqkv^T + b: A = the bf16 copy of x the previous fc2 epilogue wrote, row statistics from its partial sums
      GemmArgs g = gemm_args(w.xb_in, fw + (long)l * fold_rows * D, qkv, M, 3 * D, D, D, D, 3 * D);
      g.epi = EPI_LNFOLD; g.tag = GEMM_TAG_QKV;
      g.ln_s = sc_l; g.ln_c = sc_l + fold_rows; g.ln_part = w.part_in; g.ln_nparts = nparts; g.ln_cols = D; g.ln_eps = 1e-6f;
      // the partials (and the bf16 operand) were taken about the previous LayerNorm's row mean (m2 of layer l - 1, see the fc2 producer);
      // the statistics are written in every mode now: the next producer centres on them
      g.ln_mean = m1; g.ln_rstd = r1; g.ln_center = centred ? m2_prev : nullptr;
      RMCL_TRY(gemm(c, g, dt, dt, 1, 1));
    } else {
      RMCL_TRY(rmcl_ln_fwd(x, D, c.V(c.L(l, y.ln1_w)), c.V(c.L(l, y.ln1_b)), 1e-6f, ln1, D, dt, m1, r1, M, D, 0, s));
      GemmArgs g = gemm_args(ln1, c.W(c.L(l, y.qkv_w)), qkv, M, 3 * D, D, D, D, 3 * D);
      g.epi = EPI_BIAS; g.bias = c.V(c.L(l, y.qkv_b)); g.tag = GEMM_TAG_QKV;
      RMCL_TRY(gemm(c, g, dt, dt, 1, 1));
    }
    RMCL_TRY(rmcl_attention_fwd_impl(qkv, co_mask, ao, probs, w.scores, B, N, d->H, dt, d->exact, s));
    if (tail_req && l + 1 == d->layers) {
      // cls-only tail: of the last block's output only row 0 of every sample is read (the pooler takes hidden_states[:, 0],
      // heads.py:17), and everything behind the attention is row-wise - proj, LayerNorm 2, the MLP and the final LayerNorm run
      // on the B cls rows (fp32, exact skinny GEMMs) instead of on all B * N tokens.  Compact results live in the FIRST B rows
      // of the buffers the dense path would have filled (x_mid, u, h, ln2, x_final, the LN statistics) - the backward with
      // cls_only = 2 reads them there.  Dropout: compact row b draws the mask of the dense row b * N it stands for (gemm.h
      // drop_row_mul), so the tail gives the dense block's numbers under dropout too.
      float* ao_c = reinterpret_cast<float*>(w.dao);          // backward scratch, free during a forward
      float* ln2_c = reinterpret_cast<float*>(full ? ls.ln2 : w.ln);
      float* u_c = reinterpret_cast<float*>(keep ? ls.u : w.u);
      float* h_c = reinterpret_cast<float*>(full ? ls.h : w.h);
      float* xo_c = keep ? st.x_final : x_out;
      RMCL_TRY(rmcl_rows_gather_cast(ao, dt, ao_c, B, D, N, 0, s));
      {
        // residual operand: the cls rows of the dense residual stream, read in place (row b at b * N * D: no gather launch)
        GemmArgs g = gemm_args(ao_c, c.V(c.L(l, y.proj_w)), x_mid, B, D, D, D, D, D);
        g.epi = EPI_BIAS | EPI_RESIDUAL; g.bias = c.V(c.L(l, y.proj_b)); g.aux = x; g.ld_aux = N * D;
        with_drop(g, l, DROP_SITE_PROJ, N);
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 1, s));
      }
      RMCL_TRY(rmcl_ln_fwd(x_mid, D, c.V(c.L(l, y.ln2_w)), c.V(c.L(l, y.ln2_b)), 1e-6f, ln2_c, D, RMCL_F32, m2, r2, B, D, 0, s));
      {
        GemmArgs g = gemm_args(ln2_c, c.V(c.L(l, y.fc1_w)), h_c, B, d->mlp, D, D, D, d->mlp);
        g.epi = EPI_BIAS | EPI_GELU | EPI_SAVE_PREACT; g.bias = c.V(c.L(l, y.fc1_b)); g.C2 = u_c;
        with_drop(g, l, DROP_SITE_HIDDEN, N);
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 1, s));
      }
      {
        GemmArgs g = gemm_args(h_c, c.V(c.L(l, y.fc2_w)), xo_c, B, D, d->mlp, d->mlp, d->mlp, D);
        g.epi = EPI_BIAS | EPI_RESIDUAL; g.bias = c.V(c.L(l, y.fc2_b)); g.aux = x_mid; g.ld_aux = D;
        with_drop(g, l, DROP_SITE_FC2, N);
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 1, s));
      }
      // final LayerNorm of the cls rows, written to their places in xn (row b * N); the other rows of xn are NOT written
      RMCL_TRY(rmcl_ln_fwd(xo_c, D, c.V(y.norm_w), c.V(y.norm_b), 1e-6f, xn, (long)N * D, RMCL_F32, keep ? st.meanF : w.stat,
                           keep ? st.rstdF : w.stat + M, B, D, 0, s));
      return 0;
    }
    {
      GemmArgs g = gemm_args(ao, c.W(c.L(l, y.proj_w)), x_mid, M, D, D, D, D, D);
      g.epi = EPI_BIAS | EPI_RESIDUAL; g.bias = c.V(c.L(l, y.proj_b)); g.aux = x; g.ld_aux = D; g.tag = GEMM_TAG_PROJ;
      if (folded) { g.epi |= EPI_ROWSTAT; g.C2 = w.xb_mid; g.ln_part = w.part_mid; g.ln_nparts = nparts; g.ln_center = centred ? m1 : nullptr; }
      with_drop(g, l, DROP_SITE_PROJ);
      RMCL_TRY(gemm(c, g, dt, RMCL_F32, 1, 1));
    }
    if (folded) {
      GemmArgs g = gemm_args(w.xb_mid, fw + ((long)l * fold_rows + 3 * D) * D, h, M, d->mlp, D, D, D, d->mlp);
      g.epi = EPI_LNFOLD | EPI_GELU | (u ? EPI_SAVE_PREACT : 0); g.C2 = u; g.tag = GEMM_TAG_FC1;
      g.ln_s = sc_l + 3 * D; g.ln_c = sc_l + fold_rows + 3 * D; g.ln_part = w.part_mid; g.ln_nparts = nparts; g.ln_cols = D; g.ln_eps = 1e-6f;
      g.ln_mean = m2; g.ln_rstd = r2; g.ln_center = centred ? m1 : nullptr;
      with_drop(g, l, DROP_SITE_HIDDEN);
      RMCL_TRY(gemm(c, g, dt, dt, 1, 1));
    } else {
      RMCL_TRY(rmcl_ln_fwd(x_mid, D, c.V(c.L(l, y.ln2_w)), c.V(c.L(l, y.ln2_b)), 1e-6f, ln2, D, dt, m2, r2, M, D, 0, s));
      GemmArgs g = gemm_args(ln2, c.W(c.L(l, y.fc1_w)), h, M, d->mlp, D, D, D, d->mlp);
      g.epi = EPI_BIAS | EPI_GELU | (u ? EPI_SAVE_PREACT : 0); g.bias = c.V(c.L(l, y.fc1_b)); g.C2 = u; g.tag = GEMM_TAG_FC1;
      with_drop(g, l, DROP_SITE_HIDDEN);
      RMCL_TRY(gemm(c, g, dt, dt, 1, 1));
    }
    {
      GemmArgs g = gemm_args(h, c.W(c.L(l, y.fc2_w)), x_out, M, D, d->mlp, d->mlp, d->mlp, D);
      g.epi = EPI_BIAS | EPI_RESIDUAL; g.bias = c.V(c.L(l, y.fc2_b)); g.aux = x_mid; g.ld_aux = D; g.tag = GEMM_TAG_FC2;
      if (folded && l + 1 < d->layers) { g.epi |= EPI_ROWSTAT; g.C2 = w.xb_in; g.ln_part = w.part_in; g.ln_nparts = nparts; g.ln_center = centred ? m2 : nullptr; }
      with_drop(g, l, DROP_SITE_FC2);
      RMCL_TRY(gemm(c, g, dt, RMCL_F32, 1, 1));
    }
    x = x_out;
  }
  RMCL_TRY(rmcl_ln_fwd(x, D, c.V(y.norm_w), c.V(y.norm_b), 1e-6f, xn, D, RMCL_F32, keep ? st.meanF : w.stat,
                       keep ? st.rstdF : w.stat + M, M, D, 0, s));
  return 0;
}

int rmcl_encoder_backward(const rmcl_dims* d, int mode, const float* params32, const void* params_lp, const int64_t* text_ids,
                          const void* patches, const int32_t* co_mask, void* stash, void* workspace, const float* dxn,
                          int cls_only, void* dpatches, float* dtext, float* G, uint32_t drop_seed, float drop_p,
                          const rmcl_ragged* ragged, const void* params_lpT, void* stream) {
  RMCL_TRY(check_dims(d));
  RMCL_REQUIRE(!ragged || mode != RMCL_MODE_FULL || ragged->dpos_tok, "encoder_backward: rmcl_ragged.dpos_tok needed in FULL mode");
  RMCL_REQUIRE(mode == RMCL_MODE_DATA || mode == RMCL_MODE_FULL, "encoder_backward: mode must be DATA or FULL");
  RMCL_REQUIRE(params32 && stash && workspace && dxn && co_mask, "encoder_backward: NULL argument");
  RMCL_REQUIRE(mode != RMCL_MODE_FULL || (G && text_ids && patches), "encoder_backward: FULL mode needs grads32, text_ids, patches");
  RMCL_REQUIRE(d->dtype == RMCL_F32 || params_lp, "encoder_backward: bf16 mode needs the bf16 shadow arena");
  Ctx c{*d, params32, params_lp, {}, (hipStream_t)stream, d->dtype};
  rmcl_param_layout(d, &c.lay);
  const rmcl_layout& y = c.lay;
  Stash st{};
  Work w{};
  carve_stash(*d, mode, stash, &st);
  carve_work(*d, workspace, &w);
  const int B = d->B, L = d->L, P = d->P, N = L + 1 + P, D = d->D, M = B * N, dt = d->dtype, mlp = d->mlp;
  const bool full = mode == RMCL_MODE_FULL;
  hipStream_t s = c.s;
  auto Gp = [&](int64_t off) { return G + off; };

  // final LayerNorm backward -> dx (the residual-stream gradient, fp32)
  const float* dy = dxn;
  const bool tail = cls_only == 2;               // the forward of this stash used the cls-only tail (compact last-layer rows)
  RMCL_REQUIRE(!tail || d->B <= 1024, "encoder_backward: the cls-only tail needs B <= 1024");
  if (cls_only && !tail) {
    hipError_t e = hipMemsetAsync(w.dxn_full, 0, (size_t)M * D * sizeof(float), s);
    if (e != hipSuccess) { rmcl_set_error(hipGetErrorString(e)); return (int)e; }
    RMCL_TRY(rmcl_scatter_rows(dxn, w.dxn_full, B, D, 1, N, 0, 0, s));
    dy = w.dxn_full;
  }
  // bf16 copy of dx written by every LN backward (ping-pong T[0]/T[1]); with a side stream the weight
  // gradients of layer l run concurrently with the data-gradient chain, reading the copy the chain no
  // longer writes (events order the reuse of T / du / dqkv two sub-layers later).
  RMCL_REQUIRE(drop_p >= 0.f && drop_p < 1.f, "encoder_backward: dropout probability must be in [0,1)");
  const uint32_t dth = drop_p > 0.f ? (uint32_t)((double)drop_p * 4294967296.0) : 0u;
  const float dinv = 1.0f / (1.0f - drop_p);
  const bool lpm = dt != RMCL_F32 || dth != 0;      // a (masked) copy of dx in the GEMM operand type is needed
  // transposed bf16 weight shadows: dX = dY W as [rows][K] x [cols][K] (W^T stored [in][out], k = out contiguous)
  const bool WT = params_lpT != nullptr && dt == RMCL_BF16 && !d->exact;
  auto WTp = [&](int64_t off) { return (const void*)((const bf16_t*)params_lpT + off); };
  // GROUPED weight gradients (bf16 fast path at the step's shapes): ONE launch per layer computes the four dW, the four bias
  // gradients and finishes the layer's LayerNorm dgamma/dbeta (gemm_dw_group_kernel); otherwise four split-K GEMMs + slab
  // reduces + column sums per layer.
  const int Lr = d->layers;
  const bool grouped = full && dt == RMCL_BF16 && !d->exact && M % 64 == 0 && M >= 256 && D % 768 == 0 && mlp % 768 == 0 &&
                       2 * Lr + 1 <= 64 && rmcl_gemm_fast_get_cfg() != 2 && g_dw_grouped;
  const bool use_side = full && lpm && g_side != nullptr;
  if (use_side || full) RMCL_TRY(ensure_events());
  // copies of dx in the operand type, written by every LN backward.  Legacy path: ping-pong T[0]/T[1] (with a side stream the
  // weight gradients of layer l read the copy the chain no longer writes).  Grouped path: rotation over three copies - layer l
  // reads T[ia] (fc2) and T[ib] (proj) in its one weight-gradient launch while LN backward 1 already writes T[ic] for layer l-1.
  void* T[3] = {lpm ? w.dxT : nullptr, lpm ? ((use_side || grouped) ? w.dxT2 : w.dxT) : nullptr, lpm ? w.dxT3 : nullptr};
  void* DU[2] = {w.du, (use_side || grouped) ? w.du2 : w.du};
  void* DQ[2] = {w.dqkv, (use_side || grouped) ? w.dqkv2 : w.dqkv};
  Ctx cs{c.d, c.P32, c.Plp, c.lay, use_side ? g_side : c.s, c.dt};
  auto EV = [&](int kind, int l) { return g_ev[(kind * 32 + (l & 31)) & 127]; };   // kind 0: fork, 1: done1, 2: done2 / side done, 3: main done
  auto rep_slot = [&](int idx) { return grouped ? w.ln_rep + (size_t)idx * RMCL_LN_REP_FLOATS : nullptr; };   // 0: final norm, 1+2l: ln2, 2+2l: ln1
  if (grouped) HIP_TRY(hipMemsetAsync(w.ln_rep, 0, (size_t)(2 * Lr + 1) * RMCL_LN_REP_FLOATS * sizeof(float), s));
  // stash prefetch: while layer l's backward runs, the idle CUs touch layer l - 1's cold operands into the Infinity Cache
  const bool pref = g_prefetch_mask != 0 && g_pref != nullptr && dt == RMCL_BF16 && !d->exact && M >= 4096;
  if (pref) {
    while (g_npev < 64) {
      HIP_TRY(hipEventCreateWithFlags(&g_pev[g_npev], hipEventDisableTiming));
      ++g_npev;
    }
  }
  static int pev_rot = 0;
  auto prefetch_layer = [&](int l) -> int {
    if (!pref || l < 0) return 0;
    const LayerStash& q = st.layer[l];
    hipEvent_t ev = g_pev[(pev_rot++) & 63];
    HIP_TRY(hipEventRecord(ev, s));
    HIP_TRY(hipStreamWaitEvent(g_pref, ev, 0));
    const size_t e2 = esz(dt);
    if (g_prefetch_mask & 1) RMCL_TRY(rmcl_touch(q.u, (size_t)M * mlp * e2, g_prefetch_wgs, g_pref));
    if (g_prefetch_mask & 2) {
      RMCL_TRY(rmcl_touch(q.qkv, (size_t)M * 3 * D * e2, g_prefetch_wgs, g_pref));
      RMCL_TRY(rmcl_touch(q.ao, (size_t)M * D * e2, g_prefetch_wgs, g_pref));
    }
    if (g_prefetch_mask & 4) {
      RMCL_TRY(rmcl_touch(q.x_mid, (size_t)M * D * 4, g_prefetch_wgs, g_pref));
      RMCL_TRY(rmcl_touch(q.x_in, (size_t)M * D * 4, g_prefetch_wgs, g_pref));
    }
    return 0;
  };
  int cur = 0;
  float* dxc = w.dxn_full;                       // tail: gradient of the B cls rows [B, D] f32
  if (tail) {
    // forward ran with RMCL_MODE_CLS_TAIL: final LayerNorm backward on the B compact rows
    RMCL_TRY(rmcl_ln_bwd(dxn, D, RMCL_F32, st.x_final, D, st.meanF, st.rstdF, c.V(y.norm_w), c.V(y.norm_b), dxc, D, 0,
                         full ? Gp(y.norm_w) : nullptr, full ? Gp(y.norm_b) : nullptr, B, D, 0, s));
  } else {
    RMCL_TRY(rmcl_ln_bwd_lp(dy, D, RMCL_F32, st.x_final, D, st.meanF, st.rstdF, c.V(y.norm_w), c.V(y.norm_b), w.dx, D, 0,
                            full ? Gp(y.norm_w) : nullptr, full ? Gp(y.norm_b) : nullptr, M, D, 0, T[0], dt,
                            rmcl_site_seed(drop_seed, d->layers - 1, DROP_SITE_FC2), dth, dinv, rep_slot(0), s));
  }
  // gradient w.r.t. the LayerNorm outputs (dX GEMM -> LN backward): in the operand dtype, like du / dqkv / dao (bf16 mode
  // halves the 36 MB write + read per LayerNorm); fp32 mode is unchanged
  const int dln_dt = lpm ? dt : RMCL_F32;
  for (int l = Lr - 1; l >= 0; --l) {
    const LayerStash& ls = st.layer[l];
    RMCL_TRY(prefetch_layer(l - 1));
    void* du = DU[l & 1];
    void* dqkv = DQ[l & 1];
    const int ia = cur, ib = grouped ? (cur + 1) % 3 : cur ^ 1, ic = grouped ? (cur + 2) % 3 : cur;
    const bool tail_l = tail && l == Lr - 1;
    if (tail_l) {
      // ---- last layer, cls rows only: MLP, LayerNorm 2 and proj backward on [B, .] fp32 (exact skinny GEMMs); weight / bias
      // gradients of fc2, fc1, proj reduce over B rows instead of B * N; then the two row sets the dense chain continues
      // from are rebuilt: the residual-stream gradient w.dx and d(attention output) w.dao, zero except on the cls rows.
      float* u_c = reinterpret_cast<float*>(ls.u);
      float* h_c = reinterpret_cast<float*>(ls.h);
      float* ln2_c = reinterpret_cast<float*>(ls.ln2);
      float* du_c = reinterpret_cast<float*>(w.du);
      float* dln_c = w.dln;
      float* ao_c = reinterpret_cast<float*>(w.dqkv2);
      float* dao_c = reinterpret_cast<float*>(w.du2);
      // dropout: the gradient entering fc2 / proj is the residual-stream gradient times the mask of that branch's output - the mask of
      // the DENSE cls row b * N (forward: drop_row_mul = N); dxc itself keeps flowing down the residual path unmasked
      const float* dy2 = dxc;                                 // d(fc2 output)
      if (dth) {
        float* m2 = reinterpret_cast<float*>(w.dS);           // (attention-backward scratch: free until the attention backward below)
        RMCL_TRY(rmcl_dropout_rows(dxc, m2, B, D, N, rmcl_site_seed(drop_seed, l, DROP_SITE_FC2), dth, dinv, s));
        dy2 = m2;
      }
      {
        GemmArgs g = gemm_args(dy2, c.V(c.L(l, y.fc2_w)), du_c, B, mlp, D, D, mlp, mlp);               // du = (dx W2) * gelu'(u) [* hidden mask]
        g.epi = EPI_DGELU; g.aux = u_c; g.ld_aux = mlp;
        if (dth) { g.epi |= EPI_DROP_BWD; g.drop_seed = rmcl_site_seed(drop_seed, l, DROP_SITE_HIDDEN); g.drop_thresh = dth; g.drop_inv_keep = dinv; g.drop_row_mul = N; }
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 0, s));
      }
      if (full) {
        GemmArgs g = gemm_args(dy2, h_c, Gp(c.L(l, y.fc2_w)), D, mlp, B, D, mlp, mlp);                   // dW2 += dx^T h
        g.epi = EPI_ACCUM;
        const bool cs = rmcl_gemm_tn_shortk_takes(g);                                                    // bias gradient from the same launch
        if (cs) { g.epi |= EPI_COLSUM; g.colsum = Gp(c.L(l, y.fc2_b)); }
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
        if (!cs) RMCL_TRY(rmcl_colsum(dy2, D, RMCL_F32, Gp(c.L(l, y.fc2_b)), B, D, s));
      }
      {
        GemmArgs g = gemm_args(du_c, c.V(c.L(l, y.fc1_w)), dln_c, B, D, mlp, mlp, D, D);                 // dln2 = du W1
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 0, s));
      }
      if (full) {
        GemmArgs g = gemm_args(du_c, ln2_c, Gp(c.L(l, y.fc1_w)), mlp, D, B, mlp, D, D);                  // dW1 += du^T ln2
        g.epi = EPI_ACCUM;
        const bool cs = rmcl_gemm_tn_shortk_takes(g);
        if (cs) { g.epi |= EPI_COLSUM; g.colsum = Gp(c.L(l, y.fc1_b)); }
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
        if (!cs) RMCL_TRY(rmcl_colsum(du_c, mlp, RMCL_F32, Gp(c.L(l, y.fc1_b)), B, mlp, s));
      }
      RMCL_TRY(rmcl_ln_bwd(dln_c, D, RMCL_F32, ls.x_mid, D, ls.mean2, ls.rstd2, c.V(c.L(l, y.ln2_w)), c.V(c.L(l, y.ln2_b)), dxc, D, 1,
                           full ? Gp(c.L(l, y.ln2_w)) : nullptr, full ? Gp(c.L(l, y.ln2_b)) : nullptr, B, D, 0, s));   // dxc = d x_mid
      const float* dyp = dxc;                                 // d(proj output)
      if (dth) {
        float* mp = reinterpret_cast<float*>(w.dS) + (size_t)B * D;
        RMCL_TRY(rmcl_dropout_rows(dxc, mp, B, D, N, rmcl_site_seed(drop_seed, l, DROP_SITE_PROJ), dth, dinv, s));
        dyp = mp;
      }
      {
        GemmArgs g = gemm_args(dyp, c.V(c.L(l, y.proj_w)), dao_c, B, D, D, D, D, D);                     // dao = dx Wproj
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 1, 0, s));
      }
      if (full) {
        RMCL_TRY(rmcl_rows_gather_cast(ls.ao, dt, ao_c, B, D, N, 0, s));
        GemmArgs g = gemm_args(dyp, ao_c, Gp(c.L(l, y.proj_w)), D, D, B, D, D, D);                       // dWproj += dx^T ao
        g.epi = EPI_ACCUM;
        const bool cs = rmcl_gemm_tn_shortk_takes(g);
        if (cs) { g.epi |= EPI_COLSUM; g.colsum = Gp(c.L(l, y.proj_b)); }
        RMCL_TRY(rmcl_launch_gemm_exact(g, RMCL_F32, RMCL_F32, 0, 0, s));
        if (!cs) RMCL_TRY(rmcl_colsum(dyp, D, RMCL_F32, Gp(c.L(l, y.proj_b)), B, D, s));
      }
      HIP_TRY(hipMemsetAsync(w.dx, 0, (size_t)M * D * sizeof(float), s));
      RMCL_TRY(rmcl_scatter_rows(dxc, w.dx, B, D, 1, N, 0, 0, s));
      HIP_TRY(hipMemsetAsync(w.dao, 0, (size_t)M * D * esz(dt), s));
      RMCL_TRY(rmcl_rows_scatter_cast(dao_c, w.dao, dt, B, D, N, 0, s));
      if (full && !grouped && use_side) HIP_TRY(hipEventRecord(EV(1, l), cs.s));   // (no MLP weight-gradient work on the side stream)
    }
    // ---- MLP ----
    const void* dxT = lpm ? T[ia] : (const void*)w.dx;
    const void* dxT_b = lpm ? T[ib] : (const void*)w.dx;
    if (!tail_l) {
    if (use_side && !grouped && l + 2 < Lr) HIP_TRY(hipStreamWaitEvent(s, EV(1, l + 2), 0));   // du buffer free again
    {
      GemmArgs g = gemm_args(dxT, WT ? WTp(c.L(l, y.fc2_w)) : c.W(c.L(l, y.fc2_w)), du, M, mlp, D, D, WT ? D : mlp, mlp);   // du = (dx W2) * gelu'(u)
      g.epi = EPI_DGELU; g.aux = ls.u; g.ld_aux = mlp; g.tag = GEMM_TAG_DX;
      if (dth) { g.epi |= EPI_DROP_BWD; g.drop_seed = rmcl_site_seed(drop_seed, l, DROP_SITE_HIDDEN); g.drop_thresh = dth; g.drop_inv_keep = dinv; }
      RMCL_TRY(gemm(c, g, dt, dt, 1, WT ? 1 : 0));
    }
    if (full && !grouped) {
      if (use_side) { HIP_TRY(hipEventRecord(EV(0, 2 * l), s)); HIP_TRY(hipStreamWaitEvent(cs.s, EV(0, 2 * l), 0)); }
      RMCL_TRY(gemm_dw(cs, dxT, D, ls.h, mlp, Gp(c.L(l, y.fc2_w)), D, mlp, M, dt, w.slab, SLAB_FLOATS(*d)));
      RMCL_TRY(rmcl_colsum(dxT, D, dt, Gp(c.L(l, y.fc2_b)), M, D, cs.s));
      RMCL_TRY(gemm_dw(cs, du, mlp, ls.ln2, D, Gp(c.L(l, y.fc1_w)), mlp, D, M, dt, w.slab, SLAB_FLOATS(*d)));
      RMCL_TRY(rmcl_colsum(du, mlp, dt, Gp(c.L(l, y.fc1_b)), M, mlp, cs.s));
      if (use_side) HIP_TRY(hipEventRecord(EV(1, l), cs.s));
    }
    {
      GemmArgs g = gemm_args(du, WT ? WTp(c.L(l, y.fc1_w)) : c.W(c.L(l, y.fc1_w)), w.dln, M, D, mlp, mlp, WT ? mlp : D, D);   // dln2 = du W1
      g.tag = GEMM_TAG_DX;
      RMCL_TRY(gemm(c, g, dt, dln_dt, 1, WT ? 1 : 0));
    }
    if (use_side && l + 1 < Lr) HIP_TRY(hipStreamWaitEvent(s, EV(2, l + 1), 0));           // T[ib] free again (grouped: layer l+1's whole
                                                                                           // weight-gradient launch has finished)
    RMCL_TRY(rmcl_ln_bwd_lp(w.dln, D, dln_dt, ls.x_mid, D, ls.mean2, ls.rstd2, c.V(c.L(l, y.ln2_w)), c.V(c.L(l, y.ln2_b)), w.dx, D, 1,
                            full ? Gp(c.L(l, y.ln2_w)) : nullptr, full ? Gp(c.L(l, y.ln2_b)) : nullptr, M, D, 0, T[ib], dt,
                            rmcl_site_seed(drop_seed, l, DROP_SITE_PROJ), dth, dinv, rep_slot(1 + 2 * l), s));
    // ---- attention ----
    {
      GemmArgs g = gemm_args(dxT_b, WT ? WTp(c.L(l, y.proj_w)) : c.W(c.L(l, y.proj_w)), w.dao, M, D, D, D, D, D);   // dao = dx Wproj
      g.tag = GEMM_TAG_DX;
      RMCL_TRY(gemm(c, g, dt, dt, 1, WT ? 1 : 0));
    }
    }   // !tail_l
    if (use_side && !grouped && l + 2 < Lr) HIP_TRY(hipStreamWaitEvent(s, EV(2, l + 2), 0));   // dqkv buffer free again
    RMCL_TRY(rmcl_attention_bwd_impl(ls.qkv, co_mask, ls.probs, w.dao, ls.ao, dqkv, w.scores, w.dS, B, N, d->H, dt, d->exact, s));
    if (full && !grouped) {
      if (use_side) { HIP_TRY(hipEventRecord(EV(0, 2 * l + 1), s)); HIP_TRY(hipStreamWaitEvent(cs.s, EV(0, 2 * l + 1), 0)); }
      if (!tail_l) {
        RMCL_TRY(gemm_dw(cs, dxT_b, D, ls.ao, D, Gp(c.L(l, y.proj_w)), D, D, M, dt, w.slab, SLAB_FLOATS(*d)));
        RMCL_TRY(rmcl_colsum(dxT_b, D, dt, Gp(c.L(l, y.proj_b)), M, D, cs.s));
      }
      RMCL_TRY(gemm_dw(cs, dqkv, 3 * D, ls.ln1, D, Gp(c.L(l, y.qkv_w)), 3 * D, D, M, dt, w.slab, SLAB_FLOATS(*d)));
      RMCL_TRY(rmcl_colsum(dqkv, 3 * D, dt, Gp(c.L(l, y.qkv_b)), M, 3 * D, cs.s));
      if (use_side) HIP_TRY(hipEventRecord(EV(2, l), cs.s));
    }
    {
      GemmArgs g = gemm_args(dqkv, WT ? WTp(c.L(l, y.qkv_w)) : c.W(c.L(l, y.qkv_w)), w.dln, M, D, 3 * D, 3 * D, WT ? 3 * D : D, D);   // dln1 = dqkv Wqkv
      g.tag = GEMM_TAG_DX;
      RMCL_TRY(gemm(c, g, dt, dln_dt, 1, WT ? 1 : 0));
    }
    if (use_side && !grouped) HIP_TRY(hipStreamWaitEvent(s, EV(1, l), 0));                 // T[ic] free again
    RMCL_TRY(rmcl_ln_bwd_lp(w.dln, D, dln_dt, ls.x_in, D, ls.mean1, ls.rstd1, c.V(c.L(l, y.ln1_w)), c.V(c.L(l, y.ln1_b)), w.dx, D, 1,
                            full ? Gp(c.L(l, y.ln1_w)) : nullptr, full ? Gp(c.L(l, y.ln1_b)) : nullptr, M, D, 0, T[ic], dt,
                            rmcl_site_seed(drop_seed, l - 1, DROP_SITE_FC2), l > 0 ? dth : 0u, dinv, tail_l ? nullptr : rep_slot(2 + 2 * l), s));
    cur = ic;
    // EV(3, l): everything of layer l that runs on the main stream is enqueued (with the grouped launch on the SAME stream that
    // includes the launch, recorded below)
    if (full && (use_side || !grouped)) HIP_TRY(hipEventRecord(EV(3, l), s));
    if (grouped && tail_l) {
      // tail layer: fc2 / fc1 / proj gradients were formed from the B cls rows above; only qkv reduces over all tokens
      if (use_side) HIP_TRY(hipStreamWaitEvent(cs.s, EV(3, l), 0));
      RMCL_TRY(gemm_dw(cs, dqkv, 3 * D, ls.ln1, D, Gp(c.L(l, y.qkv_w)), 3 * D, D, M, dt, w.slab, SLAB_FLOATS(*d)));
      RMCL_TRY(rmcl_colsum(dqkv, 3 * D, dt, Gp(c.L(l, y.qkv_b)), M, 3 * D, cs.s));
      HIP_TRY(hipEventRecord(EV(use_side ? 2 : 3, l), cs.s));
    } else if (grouped) {
      // everything the layer's weight gradients read is final: both dx copies, du, dqkv, the stash, the LN replicas
      if (use_side) HIP_TRY(hipStreamWaitEvent(cs.s, EV(3, l), 0));
      DwGroupArgs a{};
      const void* As[4] = {dxT, du, dxT_b, dqkv};                                          // dY of fc2, fc1, proj, qkv
      const void* Bs[4] = {ls.h, ls.ln2, ls.ao, ls.ln1};                                   // their inputs X
      const int64_t wo[4] = {y.fc2_w, y.fc1_w, y.proj_w, y.qkv_w}, bo[4] = {y.fc2_b, y.fc1_b, y.proj_b, y.qkv_b};
      const int nout[4] = {D, mlp, D, 3 * D}, kin[4] = {mlp, D, D, D};
      a.tile_base[0] = 0;
      for (int q = 0; q < 4; ++q) {
        a.A[q] = (const unsigned short*)As[q]; a.B[q] = (const unsigned short*)Bs[q];
        a.C[q] = Gp(c.L(l, wo[q])); a.bias[q] = Gp(c.L(l, bo[q]));
        a.lda[q] = nout[q]; a.ldb[q] = kin[q]; a.ldc[q] = kin[q]; a.tiles_n[q] = kin[q] / 192;
        a.tile_base[q + 1] = a.tile_base[q] + (nout[q] / 192) * (kin[q] / 192);
      }
      a.K = M;
      a.rep = w.ln_rep; a.G = G; a.D = D;
      a.nslots = 0;
      auto add_slot = [&](int idx, int64_t gw, int64_t gb) { a.slot[a.nslots] = idx; a.g_gamma[a.nslots] = gw; a.g_beta[a.nslots] = gb; ++a.nslots; };
      add_slot(1 + 2 * l, c.L(l, y.ln2_w), c.L(l, y.ln2_b));
      add_slot(2 + 2 * l, c.L(l, y.ln1_w), c.L(l, y.ln1_b));
      if (l == Lr - 1) add_slot(0, y.norm_w, y.norm_b);
      RMCL_TRY(rmcl_launch_dw_group(a, cs.s));
      HIP_TRY(hipEventRecord(EV(use_side ? 2 : 3, l), cs.s));
    }
  }
  if (full) { g_grad_layers = Lr; g_grad_side = use_side; }
  if (use_side) HIP_TRY(hipStreamWaitEvent(s, EV(2, 0), 0));                               // join: all side work done

  // ---- embeddings ----
  RMCL_TRY(rmcl_image_assemble_bwd(w.dx, w.dpe, dt, full ? Gp(y.pos_img) : nullptr, full ? Gp(y.cls) : nullptr,
                                   full ? Gp(y.vtype) + D : nullptr, B, P, L, N, D, rmcl_site_seed(drop_seed, 0, DROP_SITE_IMAGE), dth, dinv,
                                   (ragged && full) ? ragged->dpos_tok : nullptr, s));
  if (ragged && full)                                          // position-table gradient through the per-sample resize
    RMCL_TRY(rmcl_pos_resize_bwd(ragged->dpos_tok, ragged->sel, ragged->counts, ragged->hw, ragged->sel_ld, ragged->gw, ragged->G0, B, P, D,
                                 Gp(y.pos_img), s));
  if (dpatches) {
    GemmArgs g = gemm_args(w.dpe, c.W(y.patch_w), dpatches, B * P, d->patch_k, D, D, d->patch_k, d->patch_k);
    RMCL_TRY(gemm(c, g, dt, dt, 1, 0));
  }
  if (full) {
    RMCL_TRY(gemm_dw(c, w.dpe, D, patches, d->patch_k, Gp(y.patch_w), D, d->patch_k, B * P, dt, w.slab, SLAB_FLOATS(*d)));
    RMCL_TRY(rmcl_colsum(w.dpe, D, dt, Gp(y.patch_b), B * P, D, s));
  }
  if (full || dtext) {
    // text rows: x[b*N+t] = LN(e) + vtype[0];  de = gradient wrt the embedding sum e, i.e. wrt the output of
    // word_embeddings (the tensor the reference's saliency hook captures, greedy_attack_vilt.py:414-452)
    float* de = dtext ? dtext : w.de;
    RMCL_TRY(rmcl_gather_rows(w.dx, w.dln, B * L, D, L, N, 0, s));
    if (full) RMCL_TRY(rmcl_colsum(w.dln, D, RMCL_F32, Gp(y.vtype), B * L, D, s));
    RMCL_TRY(rmcl_dropout_apply(w.dln, (long)B * L * D, rmcl_site_seed(drop_seed, 0, DROP_SITE_TEXT), dth, dinv, s));
    RMCL_TRY(rmcl_ln_bwd(w.dln, D, RMCL_F32, st.text_e, D, st.text_mean, st.text_rstd, c.V(y.eln_w), c.V(y.eln_b), de, D, 0,
                         full ? Gp(y.eln_w) : nullptr, full ? Gp(y.eln_b) : nullptr, B * L, D, 0, s));
    if (full) RMCL_TRY(rmcl_text_embed_scatter((const long*)text_ids, de, Gp(y.word), Gp(y.pos), Gp(y.btype), B, L, D, 0, s));
  }
  return 0;
}

}  // extern "C"
