// Internal launcher declarations (host side, C++ linkage).  Every launcher enqueues on `s`,
// allocates nothing, never synchronises, and returns 0 or an error code after rmcl_set_error().
#pragma once
#include <hip/hip_runtime.h>
#include <algorithm>
#include "gemm.h"

int rmcl_launch_gemm_exact(const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc, hipStream_t s);
int rmcl_launch_gemm_fast(const GemmArgs& g, int dt_out, int a_kc, int b_kc, hipStream_t s);  // bf16 in
bool rmcl_gemm_fast_supported(const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc);
int rmcl_launch_gemm_fast_slab(const GemmArgs& g, float* slab, float* out, hipStream_t s);
int rmcl_slab_reduce(const float* slab, float* out, long n, int nz, hipStream_t s);
int rmcl_gemm_fast_get_cfg();
void rmcl_gemm_skinny_set_form(int v);     // gemm_exact.hip: 0 = row-split skinny kernel only
bool rmcl_gemm_routes_to_tile192(const GemmArgs& g, int a_kc, int b_kc);
bool rmcl_lnfold_centred();          // rmcl_tune_set key 11 (api.cpp): the shift-robust form of the LayerNorm fold is on (default)
int rmcl_gemm_route_code(const GemmArgs& g, int dt_out, int a_kc, int b_kc);   // 0 128x128, 1 gemm_st, 2 gemm_sw, 3 gemm_dp, 4 / 5 256x256
// LayerNorm fold (gemm.h EPI_LNFOLD): per layer W' = bf16(W * gamma) for qkv (3D rows) then fc1 (mlp rows), and s / c vectors
int rmcl_ln_fold_launch(const float* p32, long layer0, long stride, int layers, long ln1_w, long ln1_b, long qkv_w, long qkv_b, long ln2_w,
                        long ln2_b, long fc1_w, long fc1_b, int D, int mlp, unsigned short* wf, float* sc, hipStream_t s);
// gemm_dp.hip: 192x192x32 tiles, two 4-wave workgroups per CU
int rmcl_launch_gemm_dp(const GemmArgs& g, int dt_out, hipStream_t s);
bool rmcl_gemm_dp_supported(const GemmArgs& g, int a_kc, int b_kc);
double rmcl_gemm_dp_fill(const GemmArgs& g, int cus);
bool rmcl_gemm_st_supported(const GemmArgs& g, int a_kc, int b_kc);
int rmcl_launch_gemm_st(const GemmArgs& g, int dt_out, int a_kc, int b_kc, hipStream_t s);
int rmcl_launch_gemm_st_slab(const GemmArgs& g, float* slab, float* out, hipStream_t s);

int rmcl_ln_fwd(const float* x, long ldx, const float* w, const float* b, float eps, void* y, long ldy, int dt_out,
                float* mean, float* rstd, int M, int D, int relu, hipStream_t s);
int rmcl_ln_bwd(const void* dy, long lddy, int dt_dy, const float* x, long ldx, const float* mean, const float* rstd,
                const float* w, const float* b, float* dx, long lddx, int add, float* dgamma, float* dbeta, int M, int D,
                int relu, hipStream_t s);
int rmcl_ln_bwd_lp(const void* dy, long lddy, int dt_dy, const float* x, long ldx, const float* mean, const float* rstd,
                   const float* w, const float* b, float* dx, long lddx, int add, float* dgamma, float* dbeta, int M, int D,
                   int relu, void* dx_copy, int copy_dt, uint32_t dseed, uint32_t dthresh, float dinv, float* rep_slot, hipStream_t s);
#define RMCL_LN_REP_FLOATS (32 * 2 * 1024)      // one LayerNorm's dgamma/dbeta replica slot (norm_softmax.hip LN_REP x 2 x LN_REP_LD)

// all weight gradients of one encoder layer in one launch (gemm_st.hip)
struct DwGroupArgs {
  const unsigned short* A[4];
  const unsigned short* B[4];
  float* C[4];
  float* bias[4];
  int lda[4], ldb[4], ldc[4], tiles_n[4], tile_base[5];
  int K;
  const float* rep;
  float* G;
  int nslots, D;
  int slot[3];
  long g_gamma[3], g_beta[3];
};
int rmcl_launch_dw_group(const DwGroupArgs& a, hipStream_t s);
// gemm_st.hip, prototype: up to four GEMMs of one row tile chained inside one launch (tickets: [4][64] counters + the give-up word)
enum { CHAIN_TICKETS = 64, CHAIN_ERR = 4 * 64 };
struct ChainArgs {
  GemmArgs g[4];
  int n;                       // stages
  int tiles_m, rows_per_tile;  // (set by the launcher)
  unsigned* ticket;            // [4][CHAIN_TICKETS] + 1 words, zeroed ONCE by the caller; they only grow
  unsigned target;             // 4 x (launches made with this ticket buffer, this one included)
  int flags;                   // bit 0: agent-scope release before every ticket (placement-independent hand-off)
  int* xcc;                    // optional [grid]: XCC_ID of every block (placement check)
  long long* stamps;           // optional [4][4]: wall clock (100 MHz) of block `stamp_wg` per stage: entered, ticket seen, body done, published
  int stamp_wg;
};
int rmcl_launch_gemm_chain(const ChainArgs& a, hipStream_t s);
int rmcl_softmax_fwd(const float* S, long lds, const int* mask, void* P, long ldp, int dt, int Z, int N, int H, hipStream_t s);
int rmcl_softmax_bwd(const void* P, long ldp, const float* dP, long lddp, void* dS, long ldds, int dt, int Z, int N,
                     float scale, hipStream_t s);
int rmcl_attn_fused_fwd(const void* qkv, const int* mask, void* out, float* lse, int B, int N, int H, hipStream_t s);
int rmcl_attn_fused_bwd(const void* qkv, const int* mask, const void* dout, const void* out, const float* lse, float* delta, void* dqkv,
                        int B, int N, int H, hipStream_t s);
int rmcl_colsum(const void* X, long ld, int dt, float* out, int M, int N, hipStream_t s);

int rmcl_text_embed_fwd(const long* ids, const float* word, const float* pos, const float* btype0, const float* g,
                        const float* beta, const float* vtype0, float eps, float* x, float* e_save, float* mean, float* rstd,
                        int B, int L, int N, int D, uint32_t dseed, uint32_t dthresh, float dinv, hipStream_t s);
int rmcl_dropout_apply(float* x, long n, uint32_t dseed, uint32_t dthresh, float dinv, hipStream_t s);
int rmcl_resize_u8(const unsigned char* src, const int* src_sizes, int B, int Hs, int Ws, const int* dst_sizes, int Hd, int Wd, const int* hb,
                   const int* hk, int ksh, const int* vb, const int* vk, int ksv, unsigned char* tmp, unsigned char* dst, hipStream_t s);
int rmcl_touch(const void* p, size_t bytes, int wgs, hipStream_t s);
int rmcl_debug_l2_prefetch(const void* A, long lda_b, int M, int rows_per_tile, int tiles_per_xcd, int col_tiles, const void* B, long ldb_b, int nB,
                           int nk, int tick, int lead, int per_xcd, int wgs, int* counter, long long* stamps, hipStream_t s);
int rmcl_dropout_rows(const float* in, float* out, int rows, int cols, long row_mul, uint32_t dseed, uint32_t dthresh, float dinv, hipStream_t s);
int rmcl_text_embed_scatter(const long* ids, const float* de, float* dword, float* dpos, float* dbtype0, int B, int L, int D,
                            long pad_id, hipStream_t s);
int rmcl_gather_rows(const float* in, float* out, int R, int D, int rows_per, long stride_outer, long off, hipStream_t s);
int rmcl_scatter_rows(const float* in, float* out, int R, int D, int rows_per, long stride_outer, long off, int add, hipStream_t s);
int rmcl_rows_gather_cast(const void* in, int dt, float* out, int R, int D, long stride, long off, hipStream_t s);
int rmcl_rows_scatter_cast(const float* in, void* out, int dt, int R, int D, long stride, long off, hipStream_t s);
int rmcl_image_assemble_fwd(const float* pe, const float* cls, const float* pos, const float* vtype1, float* x, int B, int P,
                            int L, int N, int D, uint32_t dseed, uint32_t dthresh, float dinv, int pos_per_sample, hipStream_t s);
int rmcl_image_assemble_bwd(const float* dx, void* dpe, int dt, float* dpos, float* dcls, float* dvtype1, int B, int P, int L,
                            int N, int D, uint32_t dseed, uint32_t dthresh, float dinv, float* dpos_tok, hipStream_t s);
int rmcl_weight_transpose(const unsigned short* src, unsigned short* dst, long layer0, long stride, int layers, const long* offs, const int* rows,
                          const int* cols, hipStream_t s);
int rmcl_patch_select(const float* img, int B, int C, int Hh, int Ww, int ps, int* sel, int* counts, int* hw, hipStream_t s);
int rmcl_im2patch_sel(float* img, float* pat, const int* sel, const int* counts, int sel_ld, int B, int n, int C, int Hh, int Ww, int ps,
                      int to_image, hipStream_t s);
int rmcl_pos_resize_fwd(const float* table, const int* sel, const int* counts, const int* hw, int sel_ld, int gw, int G0, int B, int n, int D,
                        float* out, hipStream_t s);
int rmcl_pos_resize_bwd(const float* dtok, const int* sel, const int* counts, const int* hw, int sel_ld, int gw, int G0, int B, int n, int D,
                        float* dtable, hipStream_t s);
int rmcl_u8_to_patches(const unsigned char* img, const int* sizes, const int* sel, const int* counts, int sel_ld, int B, int n, int Hmax, int Wmax,
                       const float* lut, float* pat, hipStream_t s);
int rmcl_im2patch(const float* img, float* pat, int B, int C, int Hh, int Ww, int ps, int to_image, hipStream_t s);
int rmcl_k_add_cast(const float* a, const float* d1, const float* d2, void* out, int dt, long n, hipStream_t s);
int rmcl_k_shard_sum(const void* pieces, int dt, int W, long n, float* out32, void* out_wire, hipStream_t s);
int rmcl_cast(const float* in, void* out, int dt, long n, hipStream_t s);
int rmcl_co_mask(const long* text_mask, const void* pat, int dt, int* co, int B, int L, int P, int C, int pp, hipStream_t s);
int rmcl_pgd_update(const void* g, int dt, float* delta, unsigned* amax_bits, int B, long per_sample, float lr, float eps, hipStream_t s);
int rmcl_pgd_update_fused(const void* g, int dt, float* delta, unsigned* amax_bits, int B, long per_sample, float lr, float eps,
                          const float* base, void* out, int dt_out, int flags, hipStream_t s);
int rmcl_delta_chan_norm(const float* d, float* out, long rows, int C, int pp, hipStream_t s);
int rmcl_ema(float* k, const float* q, void* k_lp, float m, long n, hipStream_t s);
int rmcl_enqueue(float* queue, const float* keys, int n, int Pd, long Kq, long ptr, hipStream_t s);
bool rmcl_gemm_tn_shortk_takes(const GemmArgs& g);                                              // (gemm_exact.hip)
bool rmcl_gemm_skinny_supported(const GemmArgs& g, int dt_in, int dt_out, int a_kc);     // (gemm_exact.hip)
int rmcl_l2norm_fwd(const float* z, float* q, float* nrm, int R, int D, float eps, hipStream_t s, float* q2 = nullptr);
int rmcl_l2norm_bwd(const float* dq, const float* q, const float* nrm, float* dz, int R, int D, hipStream_t s);
int rmcl_tanh_bwd(float* g, const float* y, long n, hipStream_t s);
int rmcl_adamw(float* p, const float* g, float* m, float* v, void* p_lp, const long* seg_end, const float* seg_lr_mult,
               const float* seg_wd, int nseg, float lr, float b1, float b2, float eps, int step, float grad_scale, long n,
               hipStream_t s);

long rmcl_infonce_workspace_bytes(int B, long Kq);
// form 0: exact-f32 matrix cores (the fp32 parity engine); 1: split-bf16 matrix cores (x = hi + lo, three products: 2^-16 relative);
// 2: as 1 without the queue-distance metrics (rows_out[6..8] = 0)
int rmcl_infonce(const float* q, const float* k, const float* queue, int B, int Pd, long Kq, float T, float gscale, float* dq,
                 float* rows_out, float* loss_sum, void* workspace, hipStream_t s, int form = 0);

int rmcl_cost_finish(float* cost, const int* txt_valid, const int* img_valid, int B, int Lt, int Li, int ld, hipStream_t s);
int rmcl_wpa_dist(const float* cost, const float* T, const float* w, float* dist, float* dsim, int B, int Lt, int Li, int ld,
                  hipStream_t s);
int rmcl_ipot(const float* cost, const int* txt_valid, const int* img_valid, float* T, int B, int Lt, int Li, int ld, float beta,
              int iters, hipStream_t s);

int rmcl_itm_head_fwd(const float* cls, const float* W, const float* bias, const int* labels, float* logits, float* dlogits,
                      float* loss_sum, int B, int D, float gscale, hipStream_t s);
int rmcl_itm_head_bwd(const float* dl, const float* cls, const float* W, float* dcls, float* dW, float* db, int B, int D, float scale,
                      hipStream_t s);
