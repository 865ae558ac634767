/* rmcl.h - C ABI of librmcl_hip.so: the MI355X (gfx950) hot path of the RMCL training step.
 *
 * The reference (stanFurrer/Robust-Multimodal-Contrastive-Learning) is 100 % Python and has no
 * FFI; its boundary for this path is the Python protocol `vilt.modules.ViLTransformerSS`
 * (vilt/modules/vilt_module.py:20-507) + the free functions in vilt/modules/objectives.py and
 * attack/pgd_attack_vilt.py.  This header is the C boundary a maintainer binds underneath that
 * protocol (ctypes stub: INTEGRATION.md).  Each entry point names the reference code it replaces.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer unless noted; no entry point allocates, frees or synchronises;
 *    work is enqueued on `stream` (a hipStream_t passed as void*); the caller keeps buffers alive
 *    until the stream has drained.
 *  - return value: 0 on success, otherwise a hipError_t value or -1 (argument check);
 *    rmcl_last_error() returns the message (thread-local, host string).
 *  - dtype codes: RMCL_F32 = 0, RMCL_BF16 = 1 (raw bfloat16 bits in uint16_t).
 *  - "arena": all parameters of one encoder live in ONE flat fp32 buffer whose element offsets are
 *    given by rmcl_param_layout(); the optional low-precision shadow arena has the same offsets.
 */
#ifndef RMCL_H
#define RMCL_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RMCL_F32 0
#define RMCL_BF16 1

#define RMCL_MODE_INFER 0 /* no activations kept (momentum encoder, clean query) */
#define RMCL_MODE_DATA 1  /* keep what the data-gradient (PGD) backward needs     */
#define RMCL_MODE_FULL 2  /* keep what the weight-gradient backward needs too     */
/* OR-ed into the mode of rmcl_encoder_forward: the caller will read ONLY row 0 (the cls token) of every sample of xn - what
 * the pooler does (heads.py:17) in the contrastive objectives.  Everything behind the last block's attention is row-wise
 * (proj, LayerNorm 2, the MLP, the final LayerNorm), so it then runs on B rows instead of B * N; the other rows of xn are
 * left unwritten.  The matching rmcl_encoder_backward must be called with cls_only = 2 (dxn = the [B, D] gradient of those
 * rows): it back-propagates the compact tail and reduces the last layer's fc2 / fc1 / proj weight gradients over B rows.
 * Requires dropout off and B <= 1024.  Results equal the dense pass on the cls rows up to fp32 summation order (the tail
 * uses the fp32 master weights).                                                                                     */
#define RMCL_MODE_CLS_TAIL 16

typedef struct rmcl_dims {
  int B;        /* samples in this pass                                              */
  int L;        /* text tokens per sample (max_text_len, 40)                         */
  int P;        /* image patches per sample ((384/32)^2 = 144)                       */
  int D;        /* hidden size 768                                                   */
  int H;        /* heads 12 (head dim D/H must be 64)                                */
  int layers;   /* 12                                                                */
  int mlp;      /* 3072                                                              */
  int patch_k;  /* 3*32*32 = 3072 (K of the patch-embedding GEMM)                    */
  int proj;     /* MoCo projection dim 128                                           */
  int vocab;    /* 30522                                                             */
  int dtype;    /* RMCL_F32 or RMCL_BF16: operand type of the encoder GEMMs          */
  int exact;    /* 1: force the exact-f32 matrix-core GEMM (bf16 operands widened)   */
  int Pp;       /* patches of the position table ((image_size/patch)^2 = 144); 0 = P. Differs from P for zero-padded
                   batches of smaller images, where P = selected patches per sample (rmcl_ragged)                  */
} rmcl_dims;

/* Zero-padded batch of smaller images (VisionTransformer.visual_embed, vision_transformer.py:559-677): `patches` then
 * holds the P SELECTED patches of every sample (rmcl_patch_select + rmcl_im2patch_sel; pad slots are zero rows, which the
 * key mask drops like the reference's), and the position embedding of a sample is the G0 x G0 table resized to its (h, w).
 * All arrays are device pointers owned by the caller.                                                              */
typedef struct rmcl_ragged {
  const int32_t* sel;      /* [B, sel_ld] flat patch index (row * gw + col) of every selected slot                   */
  const int32_t* counts;   /* [B] valid slots per sample (slots >= counts[b] are pads)                              */
  const int32_t* hw;       /* [B, 2] valid patch rows / columns of every sample (x_h, x_w)                           */
  int sel_ld;              /* row pitch of sel                                                                        */
  int gw;                  /* patch columns of the padded batch (Wmax / patch)                                       */
  int G0;                  /* side of the position table (12)                                                         */
  float* pos_tok;          /* scratch [B, P+1, D] f32: resized position rows (forward writes, every pass)            */
  float* dpos_tok;         /* scratch [B, P+1, D] f32: their gradient (backward, mode FULL)                          */
} rmcl_ragged;

/* LayerNorm folded into the GEMM that consumes it (forward passes that keep no LayerNorm output: INFER and DATA mode, bf16,
 * dropout off): y = LN(x) W^T + b = rstd * (bf16(x) W'^T - mean * s) + c.  rmcl_ln_fold derives, from an fp32 arena,
 *   wf [layers][3D + mlp][D] bf16 : W' = W * gamma for qkv (norm1) then fc1 (norm2)
 *   sc [layers][2][3D + mlp] f32  : s = row sums of W', c = W beta + bias
 * and must be re-run whenever that arena changes (optimizer step, momentum update, checkpoint load).  Passing the pair to
 * rmcl_encoder_forward removes the 2 LayerNorm launches per layer (the producer GEMMs emit the bf16 copy of the residual
 * stream and per-row partial sums in their epilogues); NULL keeps the separate LayerNorm kernels.                       */
typedef struct rmcl_fold {
  const void* wf;
  const float* sc;
} rmcl_fold;
int64_t rmcl_ln_fold_elems(const rmcl_dims* d, int which);   /* which = 0: elements of wf (bf16), 1: elements of sc (f32) */
int rmcl_ln_fold(const rmcl_dims* d, const float* params32, void* wf, float* sc, void* stream);

/* Transposed bf16 shadows of qkv / proj / fc1 / fc2 of every layer (same offsets as params_lp, each matrix stored
 * [in][out]): optional operand of rmcl_encoder_backward - the data-gradient GEMMs then read k-contiguous rows.  Re-run after
 * every change of params_lp.                                                                                         */
int rmcl_weight_transpose_bf16(const rmcl_dims* d, const void* params_lp, void* params_lpT, void* stream);

/* The two GEMM forms of the fold as single operators (the encoder pass uses them internally; exported for tests / reuse).
 * rmcl_linear_rowstat: out[M,N] f32 = A[M,K] W[N,K]^T + bias + residual, plus out_bf16 (same values) and, per row, 4 partial
 *   (sum, sum of squares) per 192 output columns: part[M][4*N/192][2].  N % 192 == 0; 192-row-tile shapes only.
 * rmcl_linear_lnfold: out[M,N] bf16 = act( rstd_m * (xb[M,K] wf[N,K]^T - mean_m * s_n) + c_n ), mean / rstd from `part`
 *   ([M][nparts][2], sums over K_ln = 768... columns); preact (optional, bf16) receives the value before the GELU.
 * Both return an error when the shape would not run on the 192-row tile kernels (no fallback).                      */
int rmcl_linear_rowstat(const void* A, const void* W, const float* bias, const float* residual, float* out, void* out_bf16, float* part,
                        int M, int N, int K, void* stream);
int rmcl_linear_lnfold(const void* xb, const void* wf, const float* s, const float* c, const float* part, int nparts, void* out,
                       void* preact, int M, int N, int K, int gelu, float eps, float* mean, float* rstd, void* stream);
/* The SHIFT-ROBUST pair (round 4; what the encoder passes run).  LayerNorm is shift-invariant, LN(x) = LN(x - c) for any per-row c:
 * with `center` [M] the producer stores out_bf16 = bf16(out - c_m) and the partial sums of (out - c_m), so the bf16 operand's rounding
 * follows the row's spread instead of its offset (a trained checkpoint's residual stream is not zero-mean; reference norms:
 * vision_transformer.py:351,362); the consumer's arithmetic is unchanged and it adds c_m back to the `mean` it writes.  The encoder
 * uses c = the row mean of the row's previous LayerNorm, which already sits in the stash.  center = NULL: the two functions above.  */
int rmcl_linear_rowstat_c(const void* A, const void* W, const float* bias, const float* residual, const float* center, float* out,
                          void* out_bf16, float* part, int M, int N, int K, void* stream);
int rmcl_linear_lnfold_c(const void* xb, const void* wf, const float* s, const float* c, const float* part, int nparts, const float* center,
                         void* out, void* preact, int M, int N, int K, int gelu, float eps, float* mean, float* rstd, void* stream);

/* PROTOTYPE (round 4): up to four bf16 GEMMs out[M,N_i] = epi(A_i[M,K_i] W_i[N_i,K_i]^T ...) of one activation row tile chained inside
 * ONE launch - the row-wise part of an encoder layer between two attention calls (reference op chain: vision_transformer.py:279-285
 * Mlp, :330-331 proj, :371-375 Block).  Stage i + 1 reads what stage i wrote (its A / residual pointers name stage i's outputs); groups
 * of four workgroups on one XCD carry a 191-row tile through all stages and synchronise through `tickets` (4 * 64 + 1 uint32, zeroed
 * once by the caller; `epoch` = 1, 2, 3, ... counts the launches made with that buffer).  epi: EPI_BIAS [| EPI_GELU | EPI_SAVE_PREACT]
 * (bf16 output), EPI_BIAS | EPI_RESIDUAL | EPI_ROWSTAT (fp32 output + bf16 copy `out2` + row partials `part`), EPI_LNFOLD (as
 * rmcl_linear_lnfold_c).  flags bit 0: placement-independent hand-off (agent-scope release per stage); 0 relies on the four blocks
 * b, b + 8, b + 16, b + 24 sharing an XCD - check with `xcc` ([grid] int32, optional).  stamps: optional [4][4] int64 wall-clock stamps of
 * block `stamp_wg`.  Measured against separate launches in tools/chain_bench.py; not used by the encoder passes.   */
typedef struct rmcl_chain_stage {
  const void* A; const void* W; const float* bias; const float* residual; void* out; void* out2; float* part; const float* center;
  const float* ln_s; const float* ln_c; float* mean; float* rstd;
  int N, K, epi, nparts; float ln_eps;
} rmcl_chain_stage;
int rmcl_gemm_chain(const rmcl_chain_stage* stages, int n, int M, uint32_t* tickets, uint32_t epoch, int flags, int32_t* xcc, int64_t* stamps,
                    int stamp_wg, void* stream);

/* EXPERIMENT (round 4, tools/l2_prefetch_bench.py): an L2 prefetch agent beside a 192-row-tile GEMM C[M,N] = A[M,K] W[N,K]^T - `wgs`
 * single-wave workgroups, of which the first `per_xcd` to arrive on each XCD (hardware XCC_ID; `counter`: 8 int32 zeroed by the caller) pull
 * one dword of every 128-byte line of the k-tiles the XCD's GEMM workgroups are about to stream (A rows of its row panels: tile ids
 * xcd * tiles_per_xcd ..., `col_tiles` column tiles per row panel; all nB rows of W), `lead` k-tiles ahead of a clock-paced schedule of
 * `tick` 10-ns ticks per k-tile.  Nothing is written but `stamps` ([8][2] int64 start / end wall clock, optional).  Not used by the
 * encoder passes; DESIGN.md section 3 records what it measured.                                                                        */
int rmcl_l2_prefetch_experiment(const void* A, int64_t lda_bytes, int M, int rows_per_tile, int tiles_per_xcd, int col_tiles, const void* W,
                                int64_t ldw_bytes, int nB, int nk, int tick, int lead, int per_xcd, int wgs, int* counter, int64_t* stamps,
                                void* stream);

/* Element offsets into a parameter arena.  Names follow the reference state dict. */
typedef struct rmcl_layout {
  int64_t word, pos, btype, eln_w, eln_b;      /* text_embeddings.{word,position,token_type}_embeddings, LayerNorm */
  int64_t vtype;                               /* token_type_embeddings.weight [2,D]                               */
  int64_t cls, pos_img, patch_w, patch_b;      /* transformer.{cls_token,pos_embed,patch_embed.proj.*}             */
  int64_t layer0, layer_stride;                /* transformer.blocks.<i> base = layer0 + i*layer_stride            */
  int64_t ln1_w, ln1_b, qkv_w, qkv_b, proj_w, proj_b, ln2_w, ln2_b, fc1_w, fc1_b, fc2_w, fc2_b; /* rel. to block */
  int64_t norm_w, norm_b;                      /* transformer.norm                                                 */
  int64_t mh0_w, mh0_b, mh1_w, mh1_b, mh3_w;   /* moco_head.projector.{0,1,3}                                      */
  int64_t ema_end;                             /* [0, ema_end) is the range the momentum update covers             */
  int64_t pool_w, pool_b;                      /* pooler.dense (query side only; vilt_module.py:405)               */
  int64_t itm_w, itm_b;                        /* itm_score.fc                                                     */
  int64_t total;
} rmcl_layout;

const char* rmcl_last_error(void);
int rmcl_version(void);

/* In-stream timing of one class of GEMM launches (tag mask: csrc/gemm.h GEMM_TAG_*) with hipEvents
 * recorded around each launch on the launch stream; used by bench.py for the roofline object.
 * rmcl_prof_end synchronises the recorded events and returns total ms, launches and algorithmic FLOPs. */
/* Tuning knobs (developer use).  key 0: bf16 GEMM tile/pipeline configuration (-1 = automatic).
 * key 1: number of CUs the persistent activation GEMMs leave to other kernels (default 8, so that RCCL's
 * channel workgroups and small side-stream kernels do not displace GEMM workgroups - at M = 64*185 the 62 x {4,12,16} tiles
 * are whole multiples of a 248-workgroup grid, so this costs nothing).
 * Further keys (csrc/api.cpp rmcl_tune_set has the list): 2 two-kernel attention backward, 3 per-GEMM weight gradients, 4 waves of the attention
 * forward, 5 InfoNCE form, 6 skinny-GEMM form, 7 gemm_dp stagger, 8 / 9 attention-backward experiments, 10 chains sharing the chip (half-batch
 * lanes), 11 (round 4) 0 = the UNCENTRED LayerNorm fold of round 2, 12 / 13 (round 4) stash prefetch of the backward: buffer mask / touch
 * workgroups - measured negative, 0 by default.  These are process-global: callers that change one around a region restore it in a finally
 * block (attack/pgd_attack_vilt.py, vilt/modules/objectives.py).                                                                          */
int rmcl_tune_set(int key, int value);

/* Optional second HIP stream: the weight-gradient GEMMs of rmcl_encoder_backward (mode FULL, bf16) then run
 * concurrently with the data-gradient chain, fork/joined with events on `stream`.  NULL disables it.     */
/* Data-parallel overlap: make `stream` wait until the gradients of encoder layer `layer` written by the most recently
 * enqueued weight-gradient backward (rmcl_encoder_backward, full mode) are complete - the bucket of that layer can then be
 * all-reduced while the backward of the layers below is still running (replaces DistributedDataParallel's bucket hooks,
 * run.py:96 / pytorch_lightning ddp).  Layers finish in the order layers-1 .. 0.                                        */
int rmcl_grad_ready_wait(int layer, void* stream);
int rmcl_set_side_stream(void* stream);
/* Stream of the stash prefetch (round 4, rmcl_tune_set(12, mask) != 0): rmcl_encoder_backward touches layer l - 1's cold stash buffers into
 * the Infinity Cache from a few workgroups on this stream while layer l's backward chain runs.  NULL / never set: no prefetch.            */
int rmcl_set_prefetch_stream(void* stream);

/* Test hook: x[i] *= dropout_mask(site seed of (drop_seed, layer, site), i), i < n (x pre-filled with ones gives the mask).
 * sites: 0 proj, 1 mlp hidden, 2 fc2, 3 text embeddings, 4 image embeddings (layer 0 for the last two).   */
int rmcl_dropout_mask_apply(float* x, int64_t n, uint32_t drop_seed, int layer, int site, float drop_p, void* stream);

int rmcl_prof_begin(int tag_mask, int max_launches);
int rmcl_prof_end(double* ms_total, int64_t* launches, double* flops_total);

void rmcl_param_layout(const rmcl_dims* d, rmcl_layout* out);
int64_t rmcl_stash_bytes(const rmcl_dims* d, int mode);
int64_t rmcl_workspace_bytes(const rmcl_dims* d);
int64_t rmcl_heads_stash_bytes(const rmcl_dims* d);

/* ---- path-level entry points --------------------------------------------------------------- */

/* image [B,3,Hh,Ww] f32 -> patch rows [B*(Hh/ps)*(Ww/ps), 3*ps*ps] f32 (to_image=0) or back (1).
 * GEMM view of PatchEmbed's Conv2d (vilt/modules/vision_transformer.py:397-409).                */
int rmcl_im2patch_f32(const float* img, float* patches, int B, int C, int Hh, int Ww, int ps, int to_image, void* stream);

/* Patch geometry of a zero-padded batch [B,C,Hh,Ww] (vision_transformer.py:563-567,605-651): per sample the selection list
 * sel[b, :] (valid patches row-major, then the first non-valid patch repeated; pitch (Hh/ps)*(Ww/ps)), counts[b] valid
 * patches and hw[b] = (x_h, x_w).  The caller picks n = min(max counts, max_image_len) and, for a sample with MORE than n
 * valid patches, overwrites its row with a random subset like the reference's multinomial draw.                      */
int rmcl_patch_select(const float* img, int B, int C, int Hh, int Ww, int ps, int32_t* sel, int32_t* counts, int32_t* hw, void* stream);
/* image <-> compact rows [B*n, C*ps*ps] of the selected patches (pad slots: zero rows / not written back; to_image=1 zeroes
 * the image first).                                                                                                  */
int rmcl_im2patch_sel(float* img, float* patches, const int32_t* sel, const int32_t* counts, int sel_ld, int B, int n, int C, int Hh,
                      int Ww, int ps, int to_image, void* stream);
/* Feed path (row f3): a decoded batch as BYTES - uint8 [B, Hmax, Wmax, 3] (HWC, every sample in its top-left corner, sizes[b] = its
 * (h, w), multiples of 32) -> normalised fp32 patch rows [B*n, 3*32*32]: ToTensor + Normalize(.5, .5) (transforms/pixelbert.py:9-17)
 * through the 256-entry table `lut`, the exact zeros of the collate padding (datasets/base_dataset.py:192-206) outside a sample,
 * and the patch cut of rmcl_im2patch_f32 / rmcl_im2patch_sel, in one pass.  sel / counts: selection of rmcl_patch_select (NULL, NULL:
 * the whole grid in row-major order, n = grid size).                                                                              */
int rmcl_image_u8_to_patches(const uint8_t* img, const int32_t* sizes, const int32_t* sel, const int32_t* counts, int sel_ld, int B, int n,
                             int Hmax, int Wmax, int patch_size, const float* lut, float* patches, void* stream);

/* MinMaxResize on the device (reference: vilt/transforms/utils.py:5-26 -> PIL Image.resize(size, BICUBIC), called from
 * transforms/pixelbert.py:9-18 in every loader worker).  src: decoded bytes of a batch, uint8 [B, Hs, Ws, 3], every sample in its top-left
 * corner, src_sizes [B,2] = (h, w) per sample; dst: uint8 [B, Hd, Wd, 3] with dst_sizes [B,2] (multiples of 32: min_max_resize_size), zero
 * outside a sample - exactly the batch rmcl_image_u8_to_patches takes; tmp: uint8 [B, Hs, Wd, 3] scratch.  The integer tables are PIL's
 * (Pillow src/libImaging/Resample.c precompute_coeffs / normalize_coeffs_8bpc, built by vilt/transforms/resample.py): hbounds [B, Wd, 2] =
 * (first source column, taps) and hk [B, Wd, ksh] = 22-bit fixed-point weights per output column, vbounds [B, Hd, 2] / vk [B, Hd, ksv]
 * per output row.  Integer arithmetic throughout: the output bytes equal PIL's (tests/test_feed_gpu.py).                                  */
int rmcl_image_resize_u8(const uint8_t* src, const int32_t* src_sizes, int B, int Hs, int Ws, const int32_t* dst_sizes, int Hd, int Wd,
                         const int32_t* hbounds, const int32_t* hk, int ksh, const int32_t* vbounds, const int32_t* vk, int ksv, uint8_t* tmp,
                         uint8_t* dst, void* stream);

/* Owner-side sum of a direct reduce-scatter over the xGMI links (replaces the reduction DDP's all-reduce performs inside the
 * collective, run.py:96): pieces [n_pieces][piece_elems] of `dtype` (F32 or BF16), this rank's slice as every rank sent
 * it; out32 = their sum (fp32 accumulation, rank order); out_wire (may be NULL) = the sum rounded to `dtype`, the operand
 * of the all-gather that follows.  piece_elems % 4 == 0.                                                            */
int rmcl_shard_sum(const void* pieces, int dtype, int n_pieces, int64_t piece_elems, float* out32, void* out_wire, void* stream);

/* out[dtype] = a + d1 + d2 (d1/d2 may be NULL): `img_init + img_delta` (attack/pgd_attack_vilt.py:144)
 * and the attacked view of objectives.py:176, fused with the cast to the GEMM operand type.      */
int rmcl_add_cast_f32(const float* a, const float* d1, const float* d2, void* out, int dtype, int64_t n, void* stream);

/* ---- Barlow-Twins variant ---------------------------------------------------------------------------
 * BarlowTwinsHead (vilt/modules/heads.py:88-107; built with [8192, 8192], 8192 at vilt_module.py:115): Linear(D,H1) -
 * BatchNorm1d - ReLU - Linear(H1,H2) - BatchNorm1d - ReLU - Linear(H2,H3) (no biases), then BatchNorm1d(H3, affine=False).
 * w1..w3, g1/b1, g2/b2: element offsets of the weights and the BatchNorm gamma/beta in ONE fp32 arena (parameters and, at
 * the same offsets, gradients).  All head arithmetic is fp32.                                                          */
typedef struct rmcl_bt_head {
  int32_t D, H1, H2, H3;
  int64_t w1, g1, b1, w2, g2, b2, w3;
} rmcl_bt_head;
int64_t rmcl_bt_stash_floats(const rmcl_bt_head* h, int B);
/* z [B,H3] = head(cls_feats [B,D]).  training != 0: batch statistics (and, when `running` is given, the running estimates
 * [mean1,var1,mean2,var2,mean3,var3] are updated with `momentum`, unbiased variance - nn.BatchNorm1d); training == 0:
 * normalise with `running`.  `stash` (rmcl_bt_stash_floats) keeps what rmcl_bt_head_backward needs.                  */
int rmcl_bt_head_forward(const rmcl_bt_head* h, const float* params, const float* cls_feats, int B, int training, float* running,
                         float momentum, float* stash, float* z, void* stream);
/* dcls [B,D] = d loss / d cls_feats given dz [B,H3]; G != NULL: weight / gamma / beta gradients are ACCUMULATED into G at the
 * head's offsets (NULL: data gradient only - the PGD inner loop, attack/pgd_attack_vilt.py:205-222).                     */
int rmcl_bt_head_backward(const rmcl_bt_head* h, const float* params, float* stash, const float* dz, int B, int training, float* G,
                          float* dcls, void* stream);
/* Cross-correlation loss of compute_barlowtwins_contrastive (vilt/modules/objectives.py:476-484, PGD form
 * attack/pgd_attack_vilt.py:219-224), in three steps so the caller can all-reduce c between the first two (:480):
 *   rmcl_bt_corr: c [N,N] = zq^T zk * inv_bs;
 *   rmcl_bt_loss: loss2 = (sum_i (c_ii - 1)^2, sum_{i != j} c_ij^2) and, IN PLACE, c <- grad_scale * d(on + lambda off)/dc;
 *                 ws: rmcl_bt_loss_ws_floats(N) floats;
 *   rmcl_bt_dz:   dzq [B,N] = zk G^T * inv_bs with G the matrix rmcl_bt_loss left in c.                                */
int rmcl_bt_corr(const float* zq, const float* zk, int B, int N, float inv_bs, float* c, void* stream);
int64_t rmcl_bt_loss_ws_floats(int N);
int rmcl_bt_loss(float* c, int N, float lambda, float grad_scale, float* ws, float* loss2, void* stream);
int rmcl_bt_dz(const float* zk, const float* G, int B, int N, float inv_bs, float* dzq, void* stream);
/* rows [B,3] = (||q_b - k_b||, cosine(q_b, k_b) eps 1e-6, q_b . k_b): the distance logs of objectives.py:496-498 */
int rmcl_bt_pair_metrics(const float* q, const float* k, int B, int N, float* rows, void* stream);

/* One joint text+image encoder forward up to transformer.norm: replaces ViLTransformerSS.infer /
 * infer_k (vilt_module.py:275-418) minus the pooler.  params32: fp32 arena; params_lp: bf16
 * shadow arena (NULL when dtype is F32).  text_ids/text_mask [B,L] int64; patches [B*P,patch_k]
 * in `dtype`; co_mask out [B,N] int32 (N = L+1+P); xn out [B*N, D] f32.
 * drop_p > 0 enables the reference's dropout sites (BertEmbeddings dropout, pos_drop, proj_drop, both Mlp
 * drops; vision_transformer.py:279-285,330-331,667) with a counter-based RNG: masks are a pure function
 * of (drop_seed, site, element), so the backward, given the same seed, regenerates them.
 * ragged: NULL for full-size images (dense fixed-order patches), else the selection of a zero-padded batch.  */
int rmcl_encoder_forward(const rmcl_dims* d, int mode, const float* params32, const void* params_lp,
                         const int64_t* text_ids, const int64_t* text_mask, const void* patches,
                         int32_t* co_mask, void* stash, void* workspace, float* xn,
                         uint32_t drop_seed, float drop_p, const rmcl_ragged* ragged, const rmcl_fold* fold, void* stream);

/* Backward of the above.  dxn: gradient wrt xn, [B*N,D] f32, or [B,D] (row 0 of every sample)
 * when cls_only=1; cls_only=2: the same [B,D] gradient for a stash whose forward ran with RMCL_MODE_CLS_TAIL (compact last
 * layer).  dpatches (optional) receives d loss/d patches [B*P,patch_k] in `dtype`
 * (the PGD data gradient, attack/pgd_attack_vilt.py:160-162).  dtext (optional) receives d loss/d
 * word-embedding output [B*L, D] f32 (the text-attack saliency, greedy_attack_vilt.py:414-452).
 * grads32 (mode FULL): gradient
 * arena, accumulated into (+=), same layout as the parameter arena.  params_lpT (optional, bf16 mode): transposed
 * weight shadows from rmcl_weight_transpose_bf16.                                                 */
int rmcl_encoder_backward(const rmcl_dims* d, int mode, const float* params32, const void* params_lp,
                          const int64_t* text_ids, const void* patches, const int32_t* co_mask,
                          void* stash, void* workspace, const float* dxn, int cls_only,
                          void* dpatches, float* dtext, float* grads32, uint32_t drop_seed, float drop_p,
                          const rmcl_ragged* ragged, const void* params_lpT, void* stream);

/* Pooler + MoCo head + L2 normalise (vilt/modules/heads.py:10-20,129-143; objectives.py:264-269).
 * pool32: arena that owns the pooler (always the query arena); head32: arena that owns the
 * moco head (query or momentum).  Outputs cls_feats [B,D] and q [B,proj] (f32).                  */
int rmcl_heads_forward(const rmcl_dims* d, const float* pool32, const float* head32, const float* xn,
                       void* hstash, float* cls_feats, float* q, void* stream);
/* The same with flags.  RMCL_HEADS_NO_WGRAD: the matching rmcl_heads_backward will be called with grads32 = NULL (key pass, PGD
 * and text-attack passes: data gradients only) - the pooler input is then not stashed and one launch less is issued.         */
#define RMCL_HEADS_NO_WGRAD 1
int rmcl_heads_forward2(const rmcl_dims* d, const float* pool32, const float* head32, const float* xn,
                        void* hstash, float* cls_feats, float* q, int flags, void* stream);
/* dq [B,proj] (may be NULL) and dcls_extra [B,D] (may be NULL, e.g. from the ITM head) -> dcls [B,D];
 * grads32 (may be NULL) accumulates pooler and moco-head weight gradients.                       */
int rmcl_heads_backward(const rmcl_dims* d, const float* pool32, const float* head32, void* hstash,
                        const float* dq, const float* dcls_extra, float* dcls, float* grads32, void* workspace, void* stream);

/* Fused InfoNCE forward + dq + queue metrics (objectives.py:328-351, pgd_attack_vilt.py:152-158).
 * rows_out [B,10]: loss_i, argmax, l_pos, pos_dist, pos_cos, pos_dot, neg_dist, neg_cos, neg_dot, lse.
 * loss_sum (optional, 1 float, += mean loss).  workspace: rmcl_infonce_ws_bytes(B,Kq) bytes.      */
int64_t rmcl_infonce_ws_bytes(int B, int64_t Kq);
int rmcl_infonce_f32(const float* q, const float* k, const float* queue, int B, int proj, int64_t Kq, float temperature,
                     float grad_scale, float* dq, float* rows_out, float* loss_sum, void* workspace, void* stream);
/* The same pass for the bf16 engine: logits and dq on the bf16 matrix cores with SPLIT operands (x = bf16(x) + bf16(x - bf16(x)),
 * three products per pair: 2^-16 relative per product, fp32 softmax / accumulation) - the fp32 queue is read as it is.
 * with_metrics = 0 (the PGD passes consume dq only): rows_out[6..8] (queue-distance means) are written as 0.              */
int rmcl_infonce_split_bf16(const float* q, const float* k, const float* queue, int B, int proj, int64_t Kq, float temperature,
                            float grad_scale, float* dq, float* rows_out, float* loss_sum, void* workspace, int with_metrics,
                            void* stream);

/* PGD ascent step in patch layout (attack/pgd_attack_vilt.py:162-173).  amax_scratch: 64 * B uint32 (per-block partial maxima of
 * |grad| per sample; need not be cleared).                                                          */
int rmcl_pgd_step(const void* grad, int dtype, float* delta, uint32_t* amax_scratch, int B, int64_t per_sample,
                  float lr, float eps, void* stream);
/* The same step fused with the loop's next operand (attack/pgd_attack_vilt.py:144,162-173): delta is updated in place and, when
 * `operand` is given, operand = cast(base + delta_new) - the next encoder forward's patch rows - in the same pass.
 * RMCL_PGD_DELTA_ZERO: the incoming delta is the all-zero delta_0 and is not read (the buffer need not be cleared).
 * RMCL_PGD_SUM_PREV: operand = cast((base + delta_old) + delta_new): the attacked view img + delta_{K-1} + delta_K that
 * objectives.py:176 builds on the batch image pgd_attack left behind.  base: the clean patch rows (fp32).                 */
#define RMCL_PGD_DELTA_ZERO 1
#define RMCL_PGD_SUM_PREV 2
int rmcl_pgd_step_fused(const void* grad, int dtype, float* delta, uint32_t* amax_scratch, int B, int64_t per_sample, float lr,
                        float eps, const float* base, void* operand, int operand_dtype, int flags, void* stream);
/* sum over (row, pixel) of the channel-wise L2 norm of delta (objectives.py:184); out += sum     */
int rmcl_delta_channel_norm(const float* delta, float* out, int64_t rows, int C, int pp, void* stream);

/* Momentum update k = m*k + (1-m)*q over n arena elements (objectives.py:219-224,257-260);
 * k_lp (optional) refreshed bf16 shadow.                                                         */
int rmcl_ema_f32(float* k, const float* q, void* k_lp, float m, int64_t n, void* stream);
/* queue[:, ptr:ptr+n] = keys^T (objectives.py:244-246); queue [proj,Kq] f32, keys [n,proj] f32.  */
int rmcl_enqueue_f32(float* queue, const float* keys, int n, int proj, int64_t Kq, int64_t ptr, void* stream);
/* f32 -> dtype cast of n elements (bf16 weight shadow refresh).                                  */
int rmcl_cast_f32(const float* in, void* out, int dtype, int64_t n, void* stream);

/* Fused AdamW over flat arenas ("next" row f1; vilt/modules/vilt_utils.py:395-398, HF AdamW).    */
int rmcl_adamw_f32(float* p, const float* g, float* m, float* v, void* p_lp, const int64_t* seg_end,
                   const float* seg_lr_mult, const float* seg_wd, int nseg, float lr, float beta1, float beta2,
                   float eps, int step, float grad_scale, int64_t n, void* stream);

/* ITM + word-patch alignment (objectives.py:24-76,714-787).  cost / dsim are [B, Lt, ld] f32 with
 * ld >= Li a multiple of 4 (GEMM operand pitch); T is [B, Li, Lt] like the reference's ipot().      */
int rmcl_gemm_batched(const void* A, const void* B, void* C, int M, int N, int K, int64_t lda, int64_t ldb, int ldc, float alpha,
                      int nbatch, int64_t strideA, int64_t strideB, int64_t strideC, int dt_in, int dt_out, int a_kc, int b_kc,
                      void* stream);
int rmcl_l2norm_rows_fwd(const float* x, float* y, float* norms, int rows, int D, float eps, void* stream);   /* cost_matrix_cosine :31-32 */
int rmcl_l2norm_rows_bwd(const float* dy, const float* y, const float* norms, float* dx, int rows, int D, void* stream);
int rmcl_wpa_cost_finish(float* cost, const int32_t* txt_valid, const int32_t* img_valid, int B, int Lt, int Li, int ld, void* stream);
int rmcl_ipot_f32(const float* cost, const int32_t* txt_valid, const int32_t* img_valid, float* T, int B, int Lt, int Li,
                  int ld, float beta, int iters, void* stream);
/* dist[b] = trace(cost_b @ T_b) (objectives.py:761); dsim = -w[b] * T^T = d(sum_b w_b dist_b)/d(cosine sim)  */
int rmcl_wpa_distance(const float* cost, const float* T, const float* w, float* dist, float* dsim, int B, int Lt, int Li, int ld,
                      void* stream);
/* ITMHead + 2-class CE (heads.py:173-180, objectives.py:764-765): logits, mean loss (+=), dlogits = grad_scale*(softmax-onehot) */
int rmcl_itm_fwd(const float* cls, const float* W, const float* bias, const int32_t* labels, float* logits, float* dlogits,
                 float* loss_sum, int B, int D, float grad_scale, void* stream);
int rmcl_itm_bwd(const float* dlogits, const float* cls, const float* W, float* dcls, float* dW, float* db, int B, int D,
                 float scale, void* stream);

/* ---- kernel-level entry points (unit parity tests) ----------------------------------------- */
/* C = epilogue(alpha * op(A) op(B)); layout kinds and epilogue flags: csrc/gemm.h                 */
int rmcl_gemm(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M, int N, int K,
              int64_t lda, int64_t ldb, int ldc, int ld_aux, float alpha, int epi, int splitk, int dt_in, int dt_out,
              int a_kc, int b_kc, int exact, void* stream);
/* Kernel family the bf16 fast path gives a GEMM of this shape / epilogue / layout to under the current tuning: 0 = 128x128 tiles,
 * 1 = 192x192 one workgroup per CU (gemm_st.hip), 2 = 192x384 (gemm_sw.hip), 3 = 192x192x32 two workgroups per CU (gemm_dp.hip),
 * 4 / 5 = 256x256.  Lets a parity test state WHICH kernels it compared with the oracle.                                          */
int rmcl_gemm_route(int M, int N, int K, int epi, int dt_out, int a_kc, int b_kc);
/* rmcl_gemm with bf16 operands stored K-BLOCKED, [K/32][rows][32] (kblk bit 0: A, bit 1: B): the layout the two-workgroups-per-CU
 * kernel (csrc/gemm_dp.hip) streams as whole 128-byte lines; A [M,K] x B [N,K]^T only; fails when no kernel takes the layout  */
int rmcl_gemm_kblk(const void* A, const void* B, void* C, void* C2, const float* bias, const void* aux, int M, int N, int K,
                   int ldc, int ld_aux, int epi, int dt_out, int kblk, void* stream);
int rmcl_layernorm_fwd(const float* x, const float* w, const float* b, float eps, void* y, int dt_out, float* mean,
                       float* rstd, int M, int D, int relu, void* stream);
int rmcl_layernorm_bwd(const void* dy, int dt_dy, const float* x, const float* mean, const float* rstd, const float* w,
                       const float* b, float* dx, int add, float* dgamma, float* dbeta, int M, int D, int relu, void* stream);
/* Masked multi-head self-attention on a packed qkv [B*N, 3*H*64] (Attention.forward,
 * vision_transformer.py:309-332).  out [B*N, H*64]; probs (stash) and scores (scratch) sized by
 * rmcl_attention_scratch_elems.  With dtype bf16 and exact=0 the fused flash-style kernels run and
 * `probs` holds only the per-row log-sum-exp.                                                                  */
int64_t rmcl_attention_scratch_elems(int B, int H, int N);
int rmcl_attention_fwd(const void* qkv, const int32_t* mask, void* out, void* probs, float* scores, int B, int N, int H,
                       int dtype, int exact, void* stream);
/* `out`: the forward's output (same buffer rmcl_attention_fwd wrote) - with it the bf16 path runs ONE backward kernel
 * (delta = rowsum(dO * O)); NULL selects the two-kernel form that recomputes delta from P and dP.                 */
int rmcl_attention_bwd(const void* qkv, const int32_t* mask, const void* probs, const void* dout, const void* out, void* dqkv,
                       float* scores, void* dscores, int B, int N, int H, int dtype, int exact, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RMCL_H */
