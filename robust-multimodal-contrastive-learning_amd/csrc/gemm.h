// GEMM argument block shared by the generic (exact-f32 MFMA) and fast (bf16 MFMA) kernels.
//
//   C[m,n] = epilogue( alpha * sum_k A(m,k) * B(k,n) )         (per batch z)
//
// Operand addressing is by "layout kind":
//   A_KC : A(m,k) = A[m*lda + k]   (k contiguous, e.g. activations [M,K])
//   A_MC : A(m,k) = A[k*lda + m]   (m contiguous, e.g. dY^T read from dY[tokens, N])
//   B_KC : B(k,n) = B[n*ldb + k]   (k contiguous, e.g. nn.Linear weight [N,K] used as W^T)
//   B_MC : B(k,n) = B[k*ldb + n]   (n contiguous, e.g. nn.Linear weight [N',K'] used as-is)
// so  forward  Y = X W^T      -> A_KC, B_KC   ("NT")
//     data grad dX = dY W     -> A_KC, B_MC   ("NN")
//     weight grad dW = dY^T X -> A_MC, B_MC   ("TN", reduction over tokens)
#pragma once
#include <stdint.h>

enum {
  EPI_BIAS = 1,         // + bias[n]
  EPI_GELU = 2,         // exact-erf GELU on the result
  EPI_SAVE_PREACT = 4,  // also store the pre-GELU value to C2 (same type/ld as C)
  EPI_RESIDUAL = 8,     // + aux_f32[m*ld_aux + n]   (fp32 residual stream)
  EPI_DGELU = 16,       // * gelu'(aux_T[m*ld_aux + n])  (aux has the INPUT element type)
  EPI_ATOMIC = 32,      // split-K: atomicAdd into fp32 C (C must be fp32, pre-zeroed or accumulating)
  EPI_ACCUM = 64,       // C += result (fp32 or bf16 read-modify-write; not with split-K)
  EPI_TANH = 128,       // tanh on the result
  EPI_COLSUM = 256,
  EPI_DROPOUT = 512,    // dropout on the result (after bias / GELU, before the residual add): mask(drop_seed, m*ldc+n)
  EPI_DROP_BWD = 1024,  // with EPI_DGELU: also multiply by the forward dropout mask of the hidden activation     // TN only: blocks with blockIdx.y==0 atomically add column sums of A (= bias grad) into bias_grad[m]
  // LayerNorm folded into the GEMM that consumes it (bf16 192x192 / 192x384 kernels only, N_in = ln_cols):
  //   y = LN(x) W^T + b  =  rstd_m * (bf16(x) W'^T - mean_m * s_n) + c_n,  W' = W * gamma (columns), s_n = sum_k W'_nk, c_n = W beta + b
  EPI_LNFOLD = 2048,    // consumer: A = bf16 copy of the residual stream, B = W'; row statistics from ln_part; per-column ln_s / ln_c
  EPI_DUP = 8192,       // skinny fp32 GEMMs: the final value is also stored to C2 (same ld as C) - a second consumer's copy without a memcpy launch
  EPI_ROWSTAT = 4096,   // producer (fp32 output = the residual stream): also store its bf16 copy to C2 and per-row partial
                        // (sum, sum of squares) of this tile's columns to ln_part[m][tile_n * 4 + wave column]
};

enum {
  GEMM_TAG_FC1 = 1, GEMM_TAG_FC2 = 2, GEMM_TAG_QKV = 4, GEMM_TAG_PROJ = 8,   // encoder forward GEMMs
  GEMM_TAG_DX = 16, GEMM_TAG_DW = 32, GEMM_TAG_PATCH = 64, GEMM_TAG_ATTN = 128, GEMM_TAG_HEAD = 256,
};

struct GemmArgs {
  const void* A;
  const void* B;
  void* C;
  void* C2;
  const float* bias;
  const void* aux;
  float* colsum;          // EPI_COLSUM target (fp32 [M])
  int M, N, K;
  long lda, ldb;
  int ldc, ld_aux;
  float alpha;
  int epi;
  int splitk;             // >1: blockIdx.z = K slice (then nbatch must be 1)
  int nb1, nb2;           // batch = nb1*nb2 (blockIdx.z = b1*nb2 + b2) when splitk == 1
  long sA1, sA2, sB1, sB2, sC1, sC2;  // element strides per batch level
  int tag;                // profiling class (GEMM_TAG_*), 0 = untagged
  uint32_t drop_seed, drop_thresh;   // dropout site seed, p * 2^32
  float drop_inv_keep;               // 1 / (1 - p)
  int drop_row_mul;                  // exact skinny GEMMs on COMPACT rows (cls-only tail): row r of this GEMM is dense row r * drop_row_mul, and the
                                     // mask index is (r * drop_row_mul) * ld + n - the same mask the dense block draws for that row.  0 = 1
  // LayerNorm fold (EPI_LNFOLD / EPI_ROWSTAT)
  const float* ln_s;                 // [N] column sums of W'
  const float* ln_c;                 // [N] W beta + bias
  float* ln_part;                    // [M][ln_nparts][2] partial (sum, sum of squares) of the LN input rows
  float* ln_mean;                    // consumer, optional: [M] mean / rstd written by the tile_n == 0 tiles (stash for the backward)
  float* ln_rstd;
  const float* ln_center;            // optional [M] per-row centre c_m (LN(x) = LN(x - c) exactly): the producer stores bf16(x - c) and the
                                     // partial sums of (x - c), so the operand's rounding scales with the row's SPREAD, not with its offset;
                                     // the consumer adds c back to the mean it stashes.  The encoder passes the row mean of the row's previous
                                     // LayerNorm (already in the stash).  NULL: c = 0 (the round-2 form)
  int ln_nparts;                     // partials per row (4 per 192-column tile of the producer)
  int ln_cols;                       // row width of the LN input (768)
  float ln_eps;
  int kblk;                          // gemm_dp.hip only: bit 0 / 1 = A / B stored k-blocked [K/32][rows][32] (a k-step of an operand tile is then
                                     // 12 KiB of whole 128-byte lines instead of 192 half lines); bit 2 = the EPI_ROWSTAT producer writes C2 k-blocked
};

#ifdef __cplusplus
// dt_in / dt_out: RMCL_F32 or RMCL_BF16.  a_kc / b_kc: layout kinds above.
// exact = 1 forces the f32-MFMA kernel (bf16 inputs are widened to f32 in LDS).
int rmcl_launch_gemm(const GemmArgs& g, int dt_in, int dt_out, int a_kc, int b_kc, int exact, hipStream_t stream);
#endif
