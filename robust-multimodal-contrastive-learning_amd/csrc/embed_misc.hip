// Embedding, patch layout, PGD update, EMA, queue and optimiser kernels (all HBM-bound, f32 math).
#include "rmcl_common.h"
#include "kernels.h"

// ---------------------------------------------------------------------------------------------
// Text embedding (HF BertEmbeddings + ViLT token type, vilt_module.py:293,315-321):
//   e = word[id] + pos[t] + btype[0];  x[b*N + t] = LN_{eps}(e) * g + beta + vtype[0]
// one wave per token row.  Optionally saves e / mean / rstd for the backward.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void text_embed_fwd_kernel(const long* __restrict__ ids, const float* __restrict__ word,
                                                             const float* __restrict__ pos, const float* __restrict__ btype0,
                                                             const float* __restrict__ g, const float* __restrict__ beta,
                                                             const float* __restrict__ vtype0, float eps, float* __restrict__ x,
                                                             float* __restrict__ e_save, float* __restrict__ mean,
                                                             float* __restrict__ rstd, int B, int L, int N, int D,
                                                             uint32_t dseed, uint32_t dthresh, float dinv) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= B * L) return;
  const int b = r / L, t = r % L;
  const long id = ids[r];
  float4 v[4];
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      const float4 a = *reinterpret_cast<const float4*>(word + id * D + c);
      const float4 p = *reinterpret_cast<const float4*>(pos + (long)t * D + c);
      const float4 q = *reinterpret_cast<const float4*>(btype0 + c);
      v[i] = make_float4(a.x + q.x + p.x, a.y + q.y + p.y, a.z + q.z + p.z, a.w + q.w + p.w);
      s += v[i].x + v[i].y + v[i].z + v[i].w;
      if (e_save) *reinterpret_cast<float4*>(e_save + (long)r * D + c) = v[i];
    }
  }
  const float mu = wave_sum(s) / D;
  float q2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      const float a = v[i].x - mu, bb = v[i].y - mu, cc = v[i].z - mu, d = v[i].w - mu;
      q2 += a * a + bb * bb + cc * cc + d * d;
    }
  }
  const float rs = rsqrtf(wave_sum(q2) / D + eps);
  if (lane == 0 && mean) { mean[r] = mu; rstd[r] = rs; }
  float* xr = x + ((long)b * N + t) * D;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int c = (lane + 64 * i) * 4;
    if (c < D) {
      const float4 ww = *reinterpret_cast<const float4*>(g + c), bb = *reinterpret_cast<const float4*>(beta + c);
      const float4 tt = *reinterpret_cast<const float4*>(vtype0 + c);
      float4 o = make_float4((v[i].x - mu) * rs * ww.x + bb.x, (v[i].y - mu) * rs * ww.y + bb.y,
                             (v[i].z - mu) * rs * ww.z + bb.z, (v[i].w - mu) * rs * ww.w + bb.w);
      if (dthresh) {                                          // BertEmbeddings.dropout, before the ViLT token type
        const uint32_t di = (uint32_t)((long)r * D + c);
        drop_scale4(dseed, di, dthresh, dinv, o.x, o.y, o.z, o.w);
      }
      *reinterpret_cast<float4*>(xr + c) = make_float4(o.x + tt.x, o.y + tt.y, o.z + tt.z, o.w + tt.w);
    }
  }
}

int rmcl_text_embed_fwd(const long* ids, const float* word, const float* pos, const float* btype0, const float* g,
                        const float* beta, const float* vtype0, float eps, float* x, float* e_save, float* mean, float* rstd,
                        int B, int L, int N, int D, uint32_t dseed, uint32_t dthresh, float dinv, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0 && D <= 1024, "text_embed: D must be a multiple of 4 and <= 1024");
  RMCL_LAUNCH(text_embed_fwd_kernel, dim3(cdiv((long)B * L, 4)), dim3(256), 0, s, ids, word, pos, btype0, g, beta, vtype0,
                     eps, x, e_save, mean, rstd, B, L, N, D, dseed, dthresh, dinv);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// Scatter-add of the embedding-sum gradient de [B*L, D] into the word / position / bert-type tables.
// Index `pad_id` receives no gradient (nn.Embedding padding_idx of HF BertEmbeddings).
__global__ __launch_bounds__(256) void text_embed_scatter_kernel(const long* __restrict__ ids, const float* __restrict__ de,
                                                                 float* __restrict__ dword, float* __restrict__ dpos,
                                                                 float* __restrict__ dbtype0, int B, int L, int D, long pad_id) {
  if ((int)blockIdx.x < B * L) {                             // word table: one token row per block (ids rarely collide)
    const int r = blockIdx.x;
    const long id = ids[r];
    if (id == pad_id) return;
    for (int c = threadIdx.x; c < D; c += 256) atomicAdd(dword + id * D + c, de[(long)r * D + c]);
  } else {                                                   // position t: ordered sum over the batch, then L adders per type-0 column
    const int t = blockIdx.x - B * L;                        // (was: every one of the B*L rows adding into the same D addresses)
    for (int c = threadIdx.x; c < D; c += 256) {
      float acc = 0.f;
      for (int b = 0; b < B; ++b) acc += de[((long)b * L + t) * D + c];
      atomicAdd(dpos + (long)t * D + c, acc);
      atomicAdd(dbtype0 + c, acc);
    }
  }
}

int rmcl_text_embed_scatter(const long* ids, const float* de, float* dword, float* dpos, float* dbtype0, int B, int L, int D,
                            long pad_id, hipStream_t s) {
  RMCL_LAUNCH(text_embed_scatter_kernel, dim3(B * L + L), dim3(256), 0, s, ids, de, dword, dpos, dbtype0, B, L, D, pad_id);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// gather rows: out[r, :] = in[(r/rows_per)*stride_outer + (r%rows_per) + off, :]   (f32, D%4==0)
__global__ __launch_bounds__(256) void gather_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int D,
                                                          int rows_per, long stride_outer, long off) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int dv = D / 4;
  if (i >= (long)R * dv) return;
  const int r = (int)(i / dv), c = (int)(i % dv) * 4;
  const long src = (long)(r / rows_per) * stride_outer + (r % rows_per) + off;
  *reinterpret_cast<float4*>(out + (long)r * D + c) = *reinterpret_cast<const float4*>(in + src * D + c);
}
int rmcl_gather_rows(const float* in, float* out, int R, int D, int rows_per, long stride_outer, long off, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0, "gather_rows: D%4");
  RMCL_LAUNCH(gather_rows_kernel, dim3(cdiv((long)R * (D / 4), 256)), dim3(256), 0, s, in, out, R, D, rows_per, stride_outer, off);
  RMCL_CHECK_LAUNCH();
  return 0;
}
// scatter rows (inverse of gather_rows): out[src(r), :] (=|+=) in[r, :]
__global__ __launch_bounds__(256) void scatter_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int R, int D,
                                                           int rows_per, long stride_outer, long off, int add) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int dv = D / 4;
  if (i >= (long)R * dv) return;
  const int r = (int)(i / dv), c = (int)(i % dv) * 4;
  const long dst = (long)(r / rows_per) * stride_outer + (r % rows_per) + off;
  float4 v = *reinterpret_cast<const float4*>(in + (long)r * D + c);
  float* p = out + dst * D + c;
  if (add) {
    const float4 o = *reinterpret_cast<const float4*>(p);
    v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
  }
  *reinterpret_cast<float4*>(p) = v;
}
int rmcl_scatter_rows(const float* in, float* out, int R, int D, int rows_per, long stride_outer, long off, int add, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0, "scatter_rows: D%4");
  RMCL_LAUNCH(scatter_rows_kernel, dim3(cdiv((long)R * (D / 4), 256)), dim3(256), 0, s, in, out, R, D, rows_per, stride_outer, off, add);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// Row gather / scatter between a [rows, D] tensor of the GEMM operand type and compact fp32 rows (the cls-only tail of the last
// encoder layer, encoder.cpp): out[r, :] = f32(in[r * stride + off, :]) and its inverse (no accumulation).
template <typename T>
__global__ __launch_bounds__(256) void rows_gather_cast_kernel(const T* __restrict__ in, float* __restrict__ out, int R, int D, long stride, long off) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * D) return;
  const int r = (int)(i / D), c = (int)(i % D);
  out[i] = to_f32<T>(in[((long)r * stride + off) * D + c]);
}
template <typename T>
__global__ __launch_bounds__(256) void rows_scatter_cast_kernel(const float* __restrict__ in, T* __restrict__ out, int R, int D, long stride, long off) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)R * D) return;
  const int r = (int)(i / D), c = (int)(i % D);
  out[((long)r * stride + off) * D + c] = from_f32<T>(in[i]);
}
int rmcl_rows_gather_cast(const void* in, int dt, float* out, int R, int D, long stride, long off, hipStream_t s) {
  const dim3 grid(cdiv((long)R * D, 256));
  if (dt == RMCL_F32) RMCL_LAUNCH(rows_gather_cast_kernel<float>, grid, dim3(256), 0, s, (const float*)in, out, R, D, stride, off);
  else RMCL_LAUNCH(rows_gather_cast_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)in, out, R, D, stride, off);
  RMCL_CHECK_LAUNCH();
  return 0;
}
int rmcl_rows_scatter_cast(const float* in, void* out, int dt, int R, int D, long stride, long off, hipStream_t s) {
  const dim3 grid(cdiv((long)R * D, 256));
  if (dt == RMCL_F32) RMCL_LAUNCH(rows_scatter_cast_kernel<float>, grid, dim3(256), 0, s, in, (float*)out, R, D, stride, off);
  else RMCL_LAUNCH(rows_scatter_cast_kernel<bf16_t>, grid, dim3(256), 0, s, in, (bf16_t*)out, R, D, stride, off);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Image token assembly (vision_transformer.py:661-667 + vilt_module.py:315-321):
//   x[b*N + L]         = cls + pos[0] + vtype[1]
//   x[b*N + L + 1 + p] = pe[b*P + p] + pos[1+p] + vtype[1]      (pe already holds conv bias)
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void image_assemble_fwd_kernel(const float* __restrict__ pe, const float* __restrict__ cls,
                                                                 const float* __restrict__ pos, const float* __restrict__ vtype1,
                                                                 float* __restrict__ x, int B, int P, int L, int N, int D,
                                                                 uint32_t dseed, uint32_t dthresh, float dinv, long pos_bstride) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int dv = D / 4;
  if (i >= (long)B * (P + 1) * dv) return;
  const int c = (int)(i % dv) * 4;
  const int tok = (int)((i / dv) % (P + 1)), b = (int)(i / ((long)dv * (P + 1)));
  const float4 a = tok == 0 ? *reinterpret_cast<const float4*>(cls + c)
                            : *reinterpret_cast<const float4*>(pe + ((long)b * P + tok - 1) * D + c);
  // pos_bstride = 0: the shared position table; (P+1)*D: per-sample rows (zero-padded batches: resized per image)
  const float4 p = *reinterpret_cast<const float4*>(pos + (long)b * pos_bstride + (long)tok * D + c);
  const float4 t = *reinterpret_cast<const float4*>(vtype1 + c);
  float4 o = make_float4(a.x + p.x, a.y + p.y, a.z + p.z, a.w + p.w);
  if (dthresh) {                                              // pos_drop (vision_transformer.py:667), before the token type
    const uint32_t di = (uint32_t)(((long)b * (P + 1) + tok) * D + c);
    drop_scale4(dseed, di, dthresh, dinv, o.x, o.y, o.z, o.w);
  }
  *reinterpret_cast<float4*>(x + ((long)b * N + L + tok) * D + c) = make_float4(o.x + t.x, o.y + t.y, o.z + t.z, o.w + t.w);
}
int rmcl_image_assemble_fwd(const float* pe, const float* cls, const float* pos, const float* vtype1, float* x, int B, int P,
                            int L, int N, int D, uint32_t dseed, uint32_t dthresh, float dinv, int pos_per_sample, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0, "image_assemble: D%4");
  RMCL_LAUNCH(image_assemble_fwd_kernel, dim3(cdiv((long)B * (P + 1) * (D / 4), 256)), dim3(256), 0, s, pe, cls, pos, vtype1, x, B, P, L, N, D,
              dseed, dthresh, dinv, pos_per_sample ? (long)(P + 1) * D : 0L);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// backward: dpe[b*P+p] = dx[b*N+L+1+p] (as T);  optional param grads:
//   dpos[tok] += sum_b dx[b, L+tok];  dcls += sum_b dx[b, L];  dvtype1 += sum over all image tokens
template <typename T>
__global__ __launch_bounds__(256) void image_assemble_bwd_kernel(const float* __restrict__ dx, T* __restrict__ dpe,
                                                                 float* __restrict__ dpos, float* __restrict__ dcls,
                                                                 float* __restrict__ dvtype1, int B, int P, int L, int N, int D,
                                                                 uint32_t dseed, uint32_t dthresh, float dinv, float* __restrict__ dpos_tok) {
  const int tok = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
  if (c >= D) return;
  float acc = 0.f, acc_raw = 0.f;
  for (int b = 0; b < B; ++b) {
    const float raw = dx[((long)b * N + L + tok) * D + c];
    const float v = dthresh ? raw * drop_scale(dseed, (uint32_t)(((long)b * (P + 1) + tok) * D + c), dthresh, dinv) : raw;
    acc += v;
    acc_raw += raw;
    if (tok > 0) dpe[((long)b * P + tok - 1) * D + c] = from_f32<T>(v);
    if (dpos_tok) dpos_tok[((long)b * (P + 1) + tok) * D + c] = v;     // per-sample position rows: scattered by pos_resize_bwd
  }
  if (dpos) {
    if (!dpos_tok) atomicAdd(dpos + (long)tok * D + c, acc);
    atomicAdd(dvtype1 + c, acc_raw);                         // the token type is added after pos_drop
    if (tok == 0) atomicAdd(dcls + c, acc);
  }
}
int rmcl_image_assemble_bwd(const float* dx, void* dpe, int dt, float* dpos, float* dcls, float* dvtype1, int B, int P, int L,
                            int N, int D, uint32_t dseed, uint32_t dthresh, float dinv, float* dpos_tok, hipStream_t s) {
  dim3 grid(P + 1, cdiv(D, 256));
  if (dt == RMCL_F32) RMCL_LAUNCH(image_assemble_bwd_kernel<float>, grid, dim3(256), 0, s, dx, (float*)dpe, dpos, dcls, dvtype1, B, P, L, N, D, dseed, dthresh, dinv, dpos_tok);
  else RMCL_LAUNCH(image_assemble_bwd_kernel<bf16_t>, grid, dim3(256), 0, s, dx, (bf16_t*)dpe, dpos, dcls, dvtype1, B, P, L, N, D, dseed, dthresh, dinv, dpos_tok);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Patch layout: image [B,C,Hh,Ww] f32 <-> patches [B*gh*gw, C*ps*ps] f32, K ordered (c,ky,kx)
// (the GEMM view of Conv2d(3,D,ps,ps,stride ps), vision_transformer.py:397-409).  ps % 4 == 0.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2patch_kernel(const float* __restrict__ img, float* __restrict__ pat, int B, int C, int Hh,
                                                       int Ww, int ps, int to_image) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;  // one float4 of the image
  const long total = (long)B * C * Hh * Ww / 4;
  if (i >= total) return;
  const long e = i * 4;
  const int xw = (int)(e % Ww), y = (int)((e / Ww) % Hh), c = (int)((e / ((long)Ww * Hh)) % C), b = (int)(e / ((long)Ww * Hh * C));
  const int gw = Ww / ps, gh = Hh / ps;
  const int px = xw / ps, kx = xw % ps, py = y / ps, ky = y % ps;
  const long prow = ((long)b * gh + py) * gw + px;
  const long pidx = prow * ((long)C * ps * ps) + ((long)c * ps + ky) * ps + kx;
  if (to_image) *reinterpret_cast<float4*>(const_cast<float*>(img) + e) = *reinterpret_cast<const float4*>(pat + pidx);
  else *reinterpret_cast<float4*>(pat + pidx) = *reinterpret_cast<const float4*>(img + e);
}
int rmcl_im2patch(const float* img, float* pat, int B, int C, int Hh, int Ww, int ps, int to_image, hipStream_t s) {
  RMCL_REQUIRE(ps % 4 == 0 && Hh % ps == 0 && Ww % ps == 0, "im2patch: image sides must be multiples of the patch size");
  RMCL_LAUNCH(im2patch_kernel, dim3(cdiv((long)B * C * Hh * Ww / 4, 256)), dim3(256), 0, s, img, pat, B, C, Hh, Ww, ps, to_image);
  RMCL_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// Zero-padded batches of smaller images (VisionTransformer.visual_embed, vision_transformer.py:559-677; collate pads
// bottom/right with zeros, base_dataset.py:192-206).
//   patch_select : per sample, pixel mask (sum_c != 0) sampled at each patch's top-left pixel (:563-565), x_h / x_w (:566-567),
//                  and the selection list: valid patches in row-major order, then the first non-valid patch repeated (the
//                  reference draws the pads at random among the non-valid patches, which are all identical tokens).
//   im2patch_sel : image <-> compact patch rows [B*n, C*ps*ps] of the selected patches (the patch GEMM then runs on n, not
//                  Gh*Gw, patches per sample; pad rows are zero)
//   pos_resize   : per-sample bilinear, align_corners=True resize of the G0 x G0 position table to (h, w) (:570-583) evaluated
//                  at the selected patches, and its transpose (scatter-add) for the position-embedding gradient.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void patch_select_kernel(const float* __restrict__ img, int C, int Hh, int Ww, int ps,
                                                           int* __restrict__ sel, int* __restrict__ counts, int* __restrict__ hw) {
  __shared__ int flag[1024];
  const int b = blockIdx.x, gh = Hh / ps, gw = Ww / ps, G = gh * gw;
  for (int p = threadIdx.x; p < G; p += 256) {
    const int py = p / gw, px = p % gw;
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += img[(((long)b * C + c) * Hh + (long)py * ps) * Ww + (long)px * ps];
    flag[p] = s != 0.f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    int h = 0, w = 0, n = 0, first_pad = -1;
    for (int py = 0; py < gh; ++py) h += flag[py * gw];
    for (int px = 0; px < gw; ++px) w += flag[px];
    for (int p = 0; p < G; ++p) {
      if (flag[p]) sel[(long)b * G + n++] = p;
      else if (first_pad < 0) first_pad = p;
    }
    for (int k = n; k < G; ++k) sel[(long)b * G + k] = first_pad < 0 ? 0 : first_pad;
    counts[b] = n;
    hw[2 * b] = h;
    hw[2 * b + 1] = w;
  }
}
int rmcl_patch_select(const float* img, int B, int C, int Hh, int Ww, int ps, int* sel, int* counts, int* hw, hipStream_t s) {
  RMCL_REQUIRE(Hh % ps == 0 && Ww % ps == 0 && (Hh / ps) * (Ww / ps) <= 1024, "patch_select: sides must be multiples of the patch size, <= 1024 patches");
  RMCL_LAUNCH(patch_select_kernel, dim3(B), dim3(256), 0, s, img, C, Hh, Ww, ps, sel, counts, hw);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// one float4 of a selected patch row per thread; sel_ld = row pitch of `sel`; to_image: scatter rows k < counts[b] back
__global__ __launch_bounds__(256) void im2patch_sel_kernel(float* __restrict__ img, float* __restrict__ pat, const int* __restrict__ sel,
                                                           const int* __restrict__ counts, int sel_ld, int B, int n, int C, int Hh, int Ww, int ps,
                                                           int to_image) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int kq = C * ps * ps / 4;
  if (i >= (long)B * n * kq) return;
  const int e = (int)(i % kq) * 4;
  const int k = (int)((i / kq) % n), b = (int)(i / ((long)kq * n));
  const int gw = Ww / ps, p = sel[(long)b * sel_ld + k];
  const int py = p / gw, px = p % gw;
  const int c = e / (ps * ps), ky = (e / ps) % ps, kx = e % ps;
  float* ip = img + (((long)b * C + c) * Hh + (long)py * ps + ky) * Ww + (long)px * ps + kx;
  float* pp = pat + ((long)b * n + k) * ((long)C * ps * ps) + e;
  if (to_image) {
    if (k < counts[b]) *reinterpret_cast<float4*>(ip) = *reinterpret_cast<const float4*>(pp);
  } else {
    *reinterpret_cast<float4*>(pp) = k < counts[b] ? *reinterpret_cast<const float4*>(ip) : make_float4(0.f, 0.f, 0.f, 0.f);
  }
}
int rmcl_im2patch_sel(float* img, float* pat, const int* sel, const int* counts, int sel_ld, int B, int n, int C, int Hh, int Ww, int ps,
                      int to_image, hipStream_t s) {
  RMCL_REQUIRE(ps % 4 == 0 && Hh % ps == 0 && Ww % ps == 0, "im2patch_sel: image sides must be multiples of the patch size");
  if (to_image) {
    hipError_t e = hipMemsetAsync(img, 0, (size_t)B * C * Hh * Ww * sizeof(float), s);
    if (e != hipSuccess) { rmcl_set_error(hipGetErrorString(e)); return (int)e; }
  }
  RMCL_LAUNCH(im2patch_sel_kernel, dim3(cdiv((long)B * n * (C * ps * ps / 4), 256)), dim3(256), 0, s, img, pat, sel, counts, sel_ld, B, n, C, Hh, Ww, ps,
              to_image);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// uint8 HWC batch -> normalised fp32 patch rows in ONE pass (row f3: the feed path).  The host pipeline hands over the decoded,
// resized images as bytes - [B, Hmax, Wmax, 3] uint8, each sample in its top-left corner, `sizes[b]` = its (h, w) - a quarter of
// the float batch's bytes over PCIe; this kernel applies ToTensor + Normalize(0.5, 0.5) through a 256-entry table (built on the
// host with the reference's own arithmetic, pixelbert.py:9-17, so the values are bit-identical to the float pipeline's), writes the
// exact zeros of BaseDataset.collate's padding (base_dataset.py:192-206) outside a sample's extent, and cuts the result into the
// patch rows of the patch-embedding GEMM (K order (c, ky, kx)) - what im2patch / im2patch_sel produce from the float image.
// One workgroup per (sample, slot): thread t takes 4 pixels (12 contiguous bytes) of patch line ky = t / 8.
__global__ __launch_bounds__(256) void u8_to_patches_kernel(const unsigned char* __restrict__ img, const int* __restrict__ sizes,
                                                            const int* __restrict__ sel, const int* __restrict__ counts, int sel_ld, int n,
                                                            int Hmax, int Wmax, const float* __restrict__ lut, float* __restrict__ pat) {
  __shared__ float tab[256];
  tab[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const int b = blockIdx.x / n, k = blockIdx.x - b * n;
  const int gw = Wmax / 32;
  const bool live = !counts || k < counts[b];
  const int p = sel ? sel[(long)b * sel_ld + k] : k;
  const int py = p / gw, px = p - py * gw;
  const int ky = threadIdx.x >> 3, kx = (threadIdx.x & 7) * 4;
  const int y = py * 32 + ky, x = px * 32 + kx;
  float v[3][4];
#pragma unroll
  for (int c = 0; c < 3; ++c)
#pragma unroll
    for (int i = 0; i < 4; ++i) v[c][i] = 0.f;
  // extents clamped to the padded batch (the host checks them too: a hand-built batch must not send this read past its sample); a
  // selected patch index outside the grid reads nothing
  const int hb = min(sizes[2 * b], Hmax), wb = min(sizes[2 * b + 1], Wmax);
  if (live && p >= 0 && y < hb && x < wb) {                                // (w is a multiple of 32: the 4 pixels are in or out together)
    const uint3 w = *reinterpret_cast<const uint3*>(img + (((long)b * Hmax + y) * Wmax + x) * 3);
    const unsigned wd[3] = {w.x, w.y, w.z};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const int byte = i * 3 + c;
        v[c][i] = tab[(wd[byte >> 2] >> ((byte & 3) * 8)) & 0xff];
      }
  }
  float* row = pat + (long)blockIdx.x * 3072 + ky * 32 + kx;
#pragma unroll
  for (int c = 0; c < 3; ++c) *reinterpret_cast<float4*>(row + c * 1024) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
}
// ---------------------------------------------------------------------------------------------
// Stash prefetch (round 4): the backward reads every layer's stash (pre-activation u 72 MB, qkv 54 MB, attention output, residual
// stream) long after the forward wrote it - HBM-cold inside GEMM epilogues and attention prologues, where the latency is exposed.  This
// kernel only TOUCHES a buffer: a few workgroups on the CUs the persistent GEMMs leave free stream it through non-temporal loads one
// layer ahead of the backward chain, so the bytes sit in the 256 MB Infinity Cache when their consumer arrives.  Nothing is written.
// ---------------------------------------------------------------------------------------------
typedef unsigned int tch_u32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(1024) void touch_kernel(const tch_u32x4* __restrict__ p, long n16, unsigned* __restrict__ sink) {
  // 16 waves x 8 independent 16-byte loads per lane = 128 KiB in flight per CU: the rate of ONE workgroup with 4 loads per lane was
  // latency-bound at a few GB/s (72 MB took over a millisecond: the touches trailed the backward they were meant to lead)
  unsigned acc = 0;
  const long stride = (long)gridDim.x * 1024;
  long i = (long)blockIdx.x * 1024 + threadIdx.x;
  for (; i + 7 * stride < n16; i += 8 * stride) {
    tch_u32x4 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = __builtin_nontemporal_load(p + i + k * stride);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc ^= v[k].x;
  }
  for (; i < n16; i += stride) acc ^= __builtin_nontemporal_load(p + i).x;
  if (acc == 0x9e3779b9u && sink) *sink = acc;                // (keeps the loads alive; practically never taken)
}
int rmcl_touch(const void* p, size_t bytes, int wgs, hipStream_t s) {
  if (!p || bytes < 16 || wgs <= 0) return 0;
  static unsigned* sink = nullptr;
  if (!sink && hipMalloc(&sink, 64) != hipSuccess) sink = nullptr;
  // the kernel asks for (almost) a whole CU's LDS although it uses none: a touch workgroup then never shares a CU with a GEMM workgroup
  // (sharing made that CU the launch's straggler: 34.8 -> 37.6 ms per step with plain 8-workgroup touches), it takes one of the CUs the
  // 248-workgroup GEMM rounds leave free
  constexpr int HOG = 150 * 1024;
  static RmclLdsOnce once;
  RMCL_TRY(rmcl_set_max_lds(once, reinterpret_cast<const void*>(touch_kernel), HOG));
  RMCL_LAUNCH(touch_kernel, dim3(wgs), dim3(1024), HOG, s, reinterpret_cast<const tch_u32x4*>(p), (long)(bytes / 16), sink);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// L2 prefetch agent (round 4, EXPERIMENT behind tools/l2_prefetch_bench.py): the GEMM k-loops run at 0.71 us per k-tile on operands that sit
// in their XCD's L2 and at ~1.0 in the step, where the first reader of every activation line misses to the Infinity Cache / HBM and the
// loop's two k-tiles of LDS-DMA lookahead do not cover that latency; the compute waves cannot look further ahead themselves (LDS is full,
// and a touch load of their own sits in the same in-order vmcnt queue as the LDS-DMA).  One single-wave workgroup per XCD - its own vmcnt,
// no LDS, a handful of registers, so it fits beside a GEMM workgroup - walks the k-tiles `lead` ahead of a clock-paced schedule and pulls
// one dword of every 128-byte line the XCD's workgroups are about to stream: the A rows of that XCD's row panels and the B rows
// of all column tiles.  Which XCD a workgroup runs on is read from the hardware (XCC_ID); the first `per_xcd` arrivals on an XCD work.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void l2_prefetch_kernel(const char* __restrict__ A, long lda_b, int M, int rows_per_tile, int tiles_per_xcd,
                                                         int col_tiles, const char* __restrict__ B, long ldb_b, int nB, int nk, int tick,
                                                         int lead, int per_xcd, int* __restrict__ counter, long long* __restrict__ stamps) {
  const int lane = threadIdx.x;
  uint32_t xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID, 0, 4)" : "=s"(xcc));
  int slot = 0;
  if (lane == 0) slot = atomicAdd(counter + xcc, 1);
  slot = __builtin_amdgcn_readfirstlane(slot);
  if (slot >= per_xcd) return;
  const int id0 = (int)xcc * tiles_per_xcd, id1 = id0 + tiles_per_xcd - 1;
  const int r_lo = (id0 / col_tiles) * rows_per_tile, r_hi = min(M, (id1 / col_tiles + 1) * rows_per_tile);
  // the touches are LDS-DMA dwords into a 256-byte sink nobody reads: no register destination, so nothing has to stay reserved while one is in
  // flight (a first version loaded into a dead VGPR by inline assembly - the allocator is free to reuse such a register at once)
  __shared__ uint32_t sink[64];
  const uint32_t sink_addr = (uint32_t)(uintptr_t)sink;
  const long long t0 = wall_clock64();
  if (stamps && lane == 0) stamps[xcc * 2] = t0;
  for (int kt = 0; kt < nk; ++kt) {
    const long kb = (long)kt * 128;
    for (int r = r_lo + slot * 64 + lane; r < r_hi; r += per_xcd * 64) {
      const char* ptr = A + (long)r * lda_b + kb;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(ptr), "s"(sink_addr) : "memory");
    }
    for (int n = slot * 64 + lane; n < nB; n += per_xcd * 64) {
      const char* ptr = B + (long)n * ldb_b + kb;
      uint32_t keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(ptr), "s"(sink_addr) : "memory");
    }
    while (wall_clock64() - t0 < (long long)(kt + 1 - lead) * tick) __builtin_amdgcn_s_sleep(2);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (stamps && lane == 0) stamps[xcc * 2 + 1] = wall_clock64();
}
int rmcl_debug_l2_prefetch(const void* A, long lda_b, int M, int rows_per_tile, int tiles_per_xcd, int col_tiles, const void* B, long ldb_b, int nB,
                           int nk, int tick, int lead, int per_xcd, int wgs, int* counter, long long* stamps, hipStream_t s) {
  RMCL_LAUNCH(l2_prefetch_kernel, dim3(wgs), dim3(64), 0, s, (const char*)A, lda_b, M, rows_per_tile, tiles_per_xcd, col_tiles, (const char*)B, ldb_b,
              nB, nk, tick, lead, per_xcd, counter, stamps);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// MinMaxResize on the device (row f3, round 4): PIL's 8-bit bicubic resize (the reference's vilt/transforms/utils.py:5-26 calls
// Image.resize(size, BICUBIC)) as two passes over the decoded bytes of a zero-padded batch [B, Hs, Ws, 3] - horizontal into a uint8
// intermediate [B, Hs, Wd, 3], then vertical into [B, Hd, Wd, 3] - with PIL's own integer tables (vilt/transforms/resample.py builds
// them on the host: per output index a first input index, a tap count and 22-bit fixed-point weights): acc = 2^21 + sum pixel * weight,
// byte = clip(acc >> 22).  Integer arithmetic end to end, so the bytes are PIL's bytes.  One thread per output pixel (3 channels); the
// taps of neighbouring pixels overlap, so the byte gathers are served by the L1 / L2; 64 images of 640 x 480 -> 512 x 384: 59 MB read
// + 50 MB written in the two passes together - microseconds on the device against ~2.5 ms per image on a host core.
// ---------------------------------------------------------------------------------------------
#define RS_BITS 22
template <bool VERT>
__global__ __launch_bounds__(256) void resize_u8_kernel(const unsigned char* __restrict__ in, unsigned char* __restrict__ out,
                                                        const int* __restrict__ in_sizes, const int* __restrict__ out_sizes, int in_h, int in_w,
                                                        int out_h, int out_w, const int* __restrict__ bounds, const int* __restrict__ kk, int ks) {
  // horizontal: in [B, in_h, in_w, 3] -> out [B, in_h, out_w, 3] (rows = the SOURCE rows; tables over output columns [B, out_w, .])
  // vertical:   in [B, in_h, in_w, 3] -> out [B, out_h, in_w, 3] (columns = the already resized ones; tables over output rows [B, out_h, .])
  const int b = blockIdx.z;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int sh = in_sizes[2 * b], dh = out_sizes[2 * b], dw = out_sizes[2 * b + 1];
  const int rows = VERT ? dh : sh, cols = dw;                    // live extent of this pass's OUTPUT for sample b
  if (y >= rows || x >= cols) return;
  const int o = VERT ? y : x, n_out = VERT ? out_h : out_w;
  const int* bd = bounds + ((long)b * n_out + o) * 2;
  const int lo = bd[0], n = bd[1];
  const int* w = kk + ((long)b * n_out + o) * ks;
  int a0 = 1 << (RS_BITS - 1), a1 = a0, a2 = a0;
  const long istride = VERT ? (long)in_w * 3 : 3;
  const unsigned char* p = in + (((long)b * in_h + (VERT ? lo : y)) * in_w + (VERT ? x : lo)) * 3;
  for (int t = 0; t < n; ++t) {
    const int c = w[t];
    a0 += (int)p[0] * c; a1 += (int)p[1] * c; a2 += (int)p[2] * c;
    p += istride;
  }
  unsigned char* q = out + (((long)b * (VERT ? out_h : in_h) + y) * (VERT ? in_w : out_w) + x) * 3;
  q[0] = (unsigned char)min(max(a0 >> RS_BITS, 0), 255);
  q[1] = (unsigned char)min(max(a1 >> RS_BITS, 0), 255);
  q[2] = (unsigned char)min(max(a2 >> RS_BITS, 0), 255);
}
int rmcl_resize_u8(const unsigned char* src, const int* src_sizes, int B, int Hs, int Ws, const int* dst_sizes, int Hd, int Wd, const int* hb,
                   const int* hk, int ksh, const int* vb, const int* vk, int ksv, unsigned char* tmp, unsigned char* dst, hipStream_t s) {
  RMCL_REQUIRE(B > 0 && Hs > 0 && Ws > 0 && Hd > 0 && Wd > 0 && ksh > 0 && ksv > 0, "resize_u8: bad dims");
  hipError_t e = hipMemsetAsync(dst, 0, (size_t)B * Hd * Wd * 3, s);           // (the pad region of the resized batch reads as zeros)
  if (e != hipSuccess) { rmcl_set_error(hipGetErrorString(e)); return (int)e; }
  RMCL_LAUNCH(resize_u8_kernel<false>, dim3(cdiv(Wd, 64), cdiv(Hs, 4), B), dim3(256), 0, s, src, tmp, src_sizes, dst_sizes, Hs, Ws, Hd, Wd, hb, hk, ksh);
  RMCL_CHECK_LAUNCH();
  RMCL_LAUNCH(resize_u8_kernel<true>, dim3(cdiv(Wd, 64), cdiv(Hd, 4), B), dim3(256), 0, s, tmp, dst, src_sizes, dst_sizes, Hs, Wd, Hd, Wd, vb, vk, ksv);
  RMCL_CHECK_LAUNCH();
  return 0;
}
int rmcl_u8_to_patches(const unsigned char* img, const int* sizes, const int* sel, const int* counts, int sel_ld, int B, int n, int Hmax, int Wmax,
                       const float* lut, float* pat, hipStream_t s) {
  RMCL_REQUIRE(Hmax % 32 == 0 && Wmax % 32 == 0 && n > 0 && B > 0, "u8_to_patches: sides must be multiples of the 32-pixel patch");
  RMCL_REQUIRE(sel || n == (Hmax / 32) * (Wmax / 32), "u8_to_patches: without a selection n must be the whole grid");
  RMCL_LAUNCH(u8_to_patches_kernel, dim3(B * n), dim3(256), 0, s, img, sizes, sel, counts, sel_ld, n, Hmax, Wmax, lut, pat);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// source index and weight of torch's bilinear align_corners=True resize (upsample_bilinear2d): src = dst * (in-1)/(out-1)
__device__ __forceinline__ void bilin_src(int dst, int in, int out, int& i0, int& i1, float& l1) {
  const float scale = out > 1 ? (float)(in - 1) / (float)(out - 1) : 0.f;
  const float src = scale * (float)dst;
  i0 = (int)src;
  i1 = i0 + (i0 < in - 1 ? 1 : 0);
  l1 = src - (float)i0;
}

// pos_tok[b, 0] = table[0]; pos_tok[b, 1+k] = resized spatial table at the k-th selected patch (0 for pad slots)
__global__ __launch_bounds__(256) void pos_resize_fwd_kernel(const float* __restrict__ table, const int* __restrict__ sel, const int* __restrict__ counts,
                                                             const int* __restrict__ hw, int sel_ld, int gw, int G0, int B, int n, int D,
                                                             float* __restrict__ out) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int dv = D / 4;
  if (i >= (long)B * (n + 1) * dv) return;
  const int c = (int)(i % dv) * 4, tok = (int)((i / dv) % (n + 1)), b = (int)(i / ((long)dv * (n + 1)));
  float4 o = make_float4(0.f, 0.f, 0.f, 0.f);
  if (tok == 0) {
    o = *reinterpret_cast<const float4*>(table + c);
  } else if (tok - 1 < counts[b]) {
    const int p = sel[(long)b * sel_ld + tok - 1], py = p / gw, px = p % gw;
    int y0, y1, x0, x1;
    float ly, lx;
    bilin_src(py, G0, hw[2 * b], y0, y1, ly);
    bilin_src(px, G0, hw[2 * b + 1], x0, x1, lx);
    const float4 v00 = *reinterpret_cast<const float4*>(table + (long)(1 + y0 * G0 + x0) * D + c);
    const float4 v01 = *reinterpret_cast<const float4*>(table + (long)(1 + y0 * G0 + x1) * D + c);
    const float4 v10 = *reinterpret_cast<const float4*>(table + (long)(1 + y1 * G0 + x0) * D + c);
    const float4 v11 = *reinterpret_cast<const float4*>(table + (long)(1 + y1 * G0 + x1) * D + c);
    const float hy = 1.f - ly, hx = 1.f - lx;
    o.x = hy * (hx * v00.x + lx * v01.x) + ly * (hx * v10.x + lx * v11.x);
    o.y = hy * (hx * v00.y + lx * v01.y) + ly * (hx * v10.y + lx * v11.y);
    o.z = hy * (hx * v00.z + lx * v01.z) + ly * (hx * v10.z + lx * v11.z);
    o.w = hy * (hx * v00.w + lx * v01.w) + ly * (hx * v10.w + lx * v11.w);
  }
  *reinterpret_cast<float4*>(out + ((long)b * (n + 1) + tok) * D + c) = o;
}
int rmcl_pos_resize_fwd(const float* table, const int* sel, const int* counts, const int* hw, int sel_ld, int gw, int G0, int B, int n, int D,
                        float* out, hipStream_t s) {
  RMCL_REQUIRE(D % 4 == 0, "pos_resize: D%4");
  RMCL_LAUNCH(pos_resize_fwd_kernel, dim3(cdiv((long)B * (n + 1) * (D / 4), 256)), dim3(256), 0, s, table, sel, counts, hw, sel_ld, gw, G0, B, n, D, out);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// transpose of the above: dtable += scatter(dpos_tok) (float atomics: a table row collects from many (sample, patch) pairs)
__global__ __launch_bounds__(256) void pos_resize_bwd_kernel(const float* __restrict__ dtok, const int* __restrict__ sel, const int* __restrict__ counts,
                                                             const int* __restrict__ hw, int sel_ld, int gw, int G0, int B, int n, int D,
                                                             float* __restrict__ dtable) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= (long)B * (n + 1) * D) return;
  const int c = (int)(i % D), tok = (int)((i / D) % (n + 1)), b = (int)(i / ((long)D * (n + 1)));
  const float g = dtok[i];
  if (tok == 0) { atomicAdd(dtable + c, g); return; }
  if (tok - 1 >= counts[b]) return;
  const int p = sel[(long)b * sel_ld + tok - 1], py = p / gw, px = p % gw;
  int y0, y1, x0, x1;
  float ly, lx;
  bilin_src(py, G0, hw[2 * b], y0, y1, ly);
  bilin_src(px, G0, hw[2 * b + 1], x0, x1, lx);
  const float hy = 1.f - ly, hx = 1.f - lx;
  atomicAdd(dtable + (long)(1 + y0 * G0 + x0) * D + c, g * hy * hx);
  atomicAdd(dtable + (long)(1 + y0 * G0 + x1) * D + c, g * hy * lx);
  atomicAdd(dtable + (long)(1 + y1 * G0 + x0) * D + c, g * ly * hx);
  atomicAdd(dtable + (long)(1 + y1 * G0 + x1) * D + c, g * ly * lx);
}
int rmcl_pos_resize_bwd(const float* dtok, const int* sel, const int* counts, const int* hw, int sel_ld, int gw, int G0, int B, int n, int D,
                        float* dtable, hipStream_t s) {
  RMCL_LAUNCH(pos_resize_bwd_kernel, dim3(cdiv((long)B * (n + 1) * D, 256)), dim3(256), 0, s, dtok, sel, counts, hw, sel_ld, gw, G0, B, n, D, dtable);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// out_T = a + d1 + d2 (d1, d2 optional), n % 4 == 0: the "img_init + delta" of pgd_attack_vilt.py:144
// fused with the cast to the GEMM operand type.
template <typename T>
__global__ __launch_bounds__(256) void add_cast_kernel(const float* __restrict__ a, const float* __restrict__ d1,
                                                       const float* __restrict__ d2, T* __restrict__ out, long n4) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 v = reinterpret_cast<const float4*>(a)[i];
    if (d1) { const float4 w = reinterpret_cast<const float4*>(d1)[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    if (d2) { const float4 w = reinterpret_cast<const float4*>(d2)[i]; v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w; }
    if constexpr (sizeof(T) == 4) reinterpret_cast<float4*>(out)[i] = v;
    else {
      uint2 pk;
      pk.x = (uint32_t)f2bf(v.x) | ((uint32_t)f2bf(v.y) << 16);
      pk.y = (uint32_t)f2bf(v.z) | ((uint32_t)f2bf(v.w) << 16);
      reinterpret_cast<uint2*>(out)[i] = pk;
    }
  }
}
int rmcl_k_add_cast(const float* a, const float* d1, const float* d2, void* out, int dt, long n, hipStream_t s) {
  RMCL_REQUIRE(n % 4 == 0, "add_cast: n%4");
  const int grid = (int)std::min<long>(cdiv(n / 4, 256), 8192);
  if (dt == RMCL_F32) RMCL_LAUNCH(add_cast_kernel<float>, dim3(grid), dim3(256), 0, s, a, d1, d2, (float*)out, n / 4);
  else RMCL_LAUNCH(add_cast_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, a, d1, d2, (bf16_t*)out, n / 4);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// Owner-side sum of the direct reduce-scatter (dist_utils.DirectReduce): pieces [W][n] holds this rank's slice as every
// rank sent it (rank-major); out32[n] = sum over W in rank order, fp32 accumulation; out_wire (optional) = the same sum in
// the wire type, the operand of the all-gather that follows.
template <typename T>
__global__ __launch_bounds__(256) void shard_sum_kernel(const T* __restrict__ pieces, int W, long n4, float* __restrict__ out32,
                                                        T* __restrict__ out_wire) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int w = 0; w < W; ++w) {
      if constexpr (sizeof(T) == 4) {
        const float4 v = reinterpret_cast<const float4*>(pieces)[(long)w * n4 + i];
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      } else {
        const uint2 v = reinterpret_cast<const uint2*>(pieces)[(long)w * n4 + i];
        acc.x += bf2f((bf16_t)(v.x & 0xffffu)); acc.y += bf2f((bf16_t)(v.x >> 16));
        acc.z += bf2f((bf16_t)(v.y & 0xffffu)); acc.w += bf2f((bf16_t)(v.y >> 16));
      }
    }
    if (out32) reinterpret_cast<float4*>(out32)[i] = acc;
    if (out_wire) {
      if constexpr (sizeof(T) == 4) reinterpret_cast<float4*>(out_wire)[i] = acc;
      else {
        uint2 pk;
        pk.x = (uint32_t)f2bf(acc.x) | ((uint32_t)f2bf(acc.y) << 16);
        pk.y = (uint32_t)f2bf(acc.z) | ((uint32_t)f2bf(acc.w) << 16);
        reinterpret_cast<uint2*>(out_wire)[i] = pk;
      }
    }
  }
}
int rmcl_k_shard_sum(const void* pieces, int dt, int W, long n, float* out32, void* out_wire, hipStream_t s) {
  RMCL_REQUIRE(n % 4 == 0 && W >= 1, "shard_sum: n%4 / W");
  const int grid = (int)std::min<long>(cdiv(n / 4, 256), 8192);
  if (dt == RMCL_F32) RMCL_LAUNCH(shard_sum_kernel<float>, dim3(grid), dim3(256), 0, s, (const float*)pieces, W, n / 4, out32, (float*)out_wire);
  else RMCL_LAUNCH(shard_sum_kernel<bf16_t>, dim3(grid), dim3(256), 0, s, (const bf16_t*)pieces, W, n / 4, out32, (bf16_t*)out_wire);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// co_mask[b, 0:L] = text_mask[b,:] != 0; co_mask[b, L] = 1; co_mask[b, L+1+p] = (sum_c patch[b*P+p][c*ps*ps] != 0)
// (pixel mask sampled at every patch's top-left pixel: vision_transformer.py:564-565,672; vilt_module.py:324)
template <typename T>
__global__ __launch_bounds__(256) void co_mask_kernel(const long* __restrict__ text_mask, const T* __restrict__ pat, int* __restrict__ co,
                                                      int B, int L, int P, int C, int pp) {
  const int i = blockIdx.x * 256 + threadIdx.x, N = L + 1 + P;
  if (i >= B * N) return;
  const int b = i / N, t = i % N;
  int m;
  if (t < L) m = text_mask[(long)b * L + t] != 0;
  else if (t == L) m = 1;
  else {
    float s = 0.f;
    for (int c = 0; c < C; ++c) s += to_f32<T>(pat[((long)b * P + (t - L - 1)) * ((long)C * pp) + (long)c * pp]);
    m = s != 0.f;
  }
  co[i] = m;
}
int rmcl_co_mask(const long* text_mask, const void* pat, int dt, int* co, int B, int L, int P, int C, int pp, hipStream_t s) {
  dim3 grid(cdiv((long)B * (L + 1 + P), 256));
  if (dt == RMCL_F32) RMCL_LAUNCH(co_mask_kernel<float>, grid, dim3(256), 0, s, text_mask, (const float*)pat, co, B, L, P, C, pp);
  else RMCL_LAUNCH(co_mask_kernel<bf16_t>, grid, dim3(256), 0, s, text_mask, (const bf16_t*)pat, co, B, L, P, C, pp);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// PGD step (attack/pgd_attack_vilt.py:162-173), in patch layout (a per-sample permutation of the
// image layout, so the per-sample inf-norm is unchanged):
//   amax[b] = max |g_b|;  delta <- clamp(delta + lr * g / max(amax[b], 1e-8), +-eps)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void absmax_kernel(const T* __restrict__ g, unsigned* __restrict__ amax_bits, long per_sample) {
  __shared__ float red[4];
  const int b = blockIdx.y;
  const T* p = g + (long)b * per_sample;
  float m = 0.f;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per_sample; i += (long)gridDim.x * 1024) {
    if constexpr (sizeof(T) == 2) {                          // per_sample % 4 == 0 (checked by the launcher): 8-byte loads
      const uint2 u = *reinterpret_cast<const uint2*>(p + i);
      m = fmaxf(fmaxf(m, fabsf(__uint_as_float(u.x << 16))), fabsf(__uint_as_float(u.x & 0xffff0000u)));
      m = fmaxf(fmaxf(m, fabsf(__uint_as_float(u.y << 16))), fabsf(__uint_as_float(u.y & 0xffff0000u)));
    } else {
      const float4 u = *reinterpret_cast<const float4*>(p + i);
      m = fmaxf(fmaxf(m, fabsf(u.x)), fmaxf(fabsf(u.y), fmaxf(fabsf(u.z), fabsf(u.w))));
    }
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0)                                      // per-block partial (slot blockIdx.x of the sample's 64): no atomics, so
    amax_bits[b * 64 + blockIdx.x] = __float_as_uint(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3])));   // no memset launch ahead of it
}
// max over the sample's partials (gridDim.x <= 64 of them), floored like pgd_attack_vilt.py:166; every thread of the block calls it
__device__ __forceinline__ float pgd_den(const unsigned* __restrict__ part, int b, int nb) {
  __shared__ float sden;
  if (threadIdx.x < 64) {
    float m = (int)threadIdx.x < nb ? __uint_as_float(part[b * 64 + threadIdx.x]) : 0.f;
    m = wave_max(m);
    if (threadIdx.x == 0) sden = m;
  }
  __syncthreads();
  return fmaxf(sden, 1e-8f);
}
template <typename T>
__global__ __launch_bounds__(256) void pgd_update_kernel(const T* __restrict__ g, const unsigned* __restrict__ amax_bits,
                                                         float* __restrict__ delta, long per_sample, float lr, float eps) {
  const int b = blockIdx.y;
  const float den = pgd_den(amax_bits, b, gridDim.x);
  const T* p = g + (long)b * per_sample;
  float* d = delta + (long)b * per_sample;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per_sample; i += (long)gridDim.x * 1024) {
    float4 v = *reinterpret_cast<float4*>(d + i);
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = o[j] + lr * to_f32<T>(p[i + j]) / den;
      if (eps > 0.f) o[j] = fminf(fmaxf(o[j], -eps), eps);
    }
    *reinterpret_cast<float4*>(d + i) = make_float4(o[0], o[1], o[2], o[3]);
  }
}
int rmcl_pgd_update(const void* g, int dt, float* delta, unsigned* amax_bits, int B, long per_sample, float lr, float eps, hipStream_t s) {
  RMCL_REQUIRE(per_sample % 4 == 0, "pgd_update: per_sample%4");
  dim3 grid(std::min<long>(cdiv(per_sample, 1024), 64), B);
  if (dt == RMCL_F32) {
    RMCL_LAUNCH(absmax_kernel<float>, grid, dim3(256), 0, s, (const float*)g, amax_bits, per_sample);
    RMCL_LAUNCH(pgd_update_kernel<float>, grid, dim3(256), 0, s, (const float*)g, amax_bits, delta, per_sample, lr, eps);
  } else {
    RMCL_LAUNCH(absmax_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)g, amax_bits, per_sample);
    RMCL_LAUNCH(pgd_update_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)g, amax_bits, delta, per_sample, lr, eps);
  }
  RMCL_CHECK_LAUNCH();
  return 0;
}

// The same step fused with what the loop does next: delta_new = clamp(delta_old + lr * g / max|g|, +-eps) AND, in the same pass,
// the operand of the next encoder forward, cast(base + delta_new) (pgd_attack_vilt.py:144) - or, on the last step, the attacked view
// cast((base + delta_old) + delta_new) (the reference's img + delta_{K-1} + delta_K, objectives.py:176).  ZERO: delta_old is the
// all-zero delta_0 of step 0 and is not read (no zero fill of the 113 MB buffer).  Replaces pgd_update + add_cast (+ the
// delta_prev copy and the three-input add_cast of the attacked view): 623 -> 510 MB per step and one launch less.
template <typename T, typename TO, bool ZERO, bool SUMPREV>
__global__ __launch_bounds__(256) void pgd_update_fused_kernel(const T* __restrict__ g, const unsigned* __restrict__ amax_bits,
                                                               float* __restrict__ delta, const float* __restrict__ base,
                                                               TO* __restrict__ out, long per_sample, float lr, float eps) {
  const int b = blockIdx.y;
  const float den = pgd_den(amax_bits, b, gridDim.x);
  const long off = (long)b * per_sample;
  for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < per_sample; i += (long)gridDim.x * 1024) {
    float old[4] = {0.f, 0.f, 0.f, 0.f};
    if (!ZERO) { const float4 v = *reinterpret_cast<const float4*>(delta + off + i); old[0] = v.x; old[1] = v.y; old[2] = v.z; old[3] = v.w; }
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = old[j] + lr * to_f32<T>(g[off + i + j]) / den;
      if (eps > 0.f) o[j] = fminf(fmaxf(o[j], -eps), eps);
    }
    *reinterpret_cast<float4*>(delta + off + i) = make_float4(o[0], o[1], o[2], o[3]);
    if (out) {
      const float4 a = *reinterpret_cast<const float4*>(base + off + i);
      float v[4] = {a.x, a.y, a.z, a.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = SUMPREV ? (v[j] + old[j]) + o[j] : v[j] + o[j];
      if constexpr (sizeof(TO) == 4) *reinterpret_cast<float4*>(out + off + i) = make_float4(v[0], v[1], v[2], v[3]);
      else {
        uint2 pk;
        pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16);
        pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
        *reinterpret_cast<uint2*>(out + off + i) = pk;
      }
    }
  }
}
template <typename T, typename TO>
static void pgd_fused_launch(const void* g, const unsigned* amax, float* delta, const float* base, void* out, long per, float lr, float eps,
                             int flags, dim3 grid, hipStream_t s) {
  const T* gp = (const T*)g;
  TO* op = (TO*)out;
  const bool zero = flags & 1, sum = (flags & 2) && !zero;       // (delta_old = 0: the sum with it is the plain form)
  if (zero) RMCL_LAUNCH((pgd_update_fused_kernel<T, TO, true, false>), grid, dim3(256), 0, s, gp, amax, delta, base, op, per, lr, eps);
  else if (sum) RMCL_LAUNCH((pgd_update_fused_kernel<T, TO, false, true>), grid, dim3(256), 0, s, gp, amax, delta, base, op, per, lr, eps);
  else RMCL_LAUNCH((pgd_update_fused_kernel<T, TO, false, false>), grid, dim3(256), 0, s, gp, amax, delta, base, op, per, lr, eps);
}
int rmcl_pgd_update_fused(const void* g, int dt, float* delta, unsigned* amax_bits, int B, long per_sample, float lr, float eps,
                          const float* base, void* out, int dt_out, int flags, hipStream_t s) {
  RMCL_REQUIRE(per_sample % 4 == 0, "pgd_update: per_sample%4");
  RMCL_REQUIRE(!out || base, "pgd_update: an operand output needs the base image");
  dim3 grid(std::min<long>(cdiv(per_sample, 1024), 64), B);
  if (dt == RMCL_F32) RMCL_LAUNCH(absmax_kernel<float>, grid, dim3(256), 0, s, (const float*)g, amax_bits, per_sample);
  else RMCL_LAUNCH(absmax_kernel<bf16_t>, grid, dim3(256), 0, s, (const bf16_t*)g, amax_bits, per_sample);
  if (dt == RMCL_F32) {
    if (dt_out == RMCL_F32) pgd_fused_launch<float, float>(g, amax_bits, delta, base, out, per_sample, lr, eps, flags, grid, s);
    else pgd_fused_launch<float, bf16_t>(g, amax_bits, delta, base, out, per_sample, lr, eps, flags, grid, s);
  } else {
    if (dt_out == RMCL_F32) pgd_fused_launch<bf16_t, float>(g, amax_bits, delta, base, out, per_sample, lr, eps, flags, grid, s);
    else pgd_fused_launch<bf16_t, bf16_t>(g, amax_bits, delta, base, out, per_sample, lr, eps, flags, grid, s);
  }
  RMCL_CHECK_LAUNCH();
  return 0;
}

// mean over (b, y, x) of the channel-wise L2 norm of delta (objectives.py:184), delta in patch layout
// [B*P, C*pp]: out += sum_{row, i<pp} sqrt(sum_c delta[row][c*pp + i]^2)   (caller divides by B*P*pp)
__global__ __launch_bounds__(256) void delta_chan_norm_kernel(const float* __restrict__ d, float* __restrict__ out, long rows, int C, int pp) {
  __shared__ float red[4];
  float acc = 0.f;
  const int pv = pp / 4;                                     // pp % 4 == 0 (launcher): float4 per lane
  const long total = rows * pv;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / pv;
    const int k = (int)(i % pv) * 4;
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int c = 0; c < C; ++c) {
      const float4 v = *reinterpret_cast<const float4*>(d + r * ((long)C * pp) + (long)c * pp + k);
      s.x += v.x * v.x; s.y += v.y * v.y; s.z += v.z * v.z; s.w += v.w * v.w;
    }
    acc += sqrtf(s.x) + sqrtf(s.y) + sqrtf(s.z) + sqrtf(s.w);
  }
  acc = wave_sum(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(out, (red[0] + red[1]) + (red[2] + red[3]));   // one atomic per block
}
int rmcl_delta_chan_norm(const float* d, float* out, long rows, int C, int pp, hipStream_t s) {
  RMCL_REQUIRE(pp % 4 == 0, "delta_chan_norm: pixels per patch % 4");
  RMCL_LAUNCH(delta_chan_norm_kernel, dim3(std::min<long>(cdiv(rows * (pp / 4), 256), 2048)), dim3(256), 0, s, d, out, rows, C, pp);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// EMA of the momentum encoder over a flat parameter arena (objectives.py:219-224,257-260):
//   k = k*m + q*(1-m);  optionally refreshes the bf16 shadow of k in the same pass.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void ema_kernel(float* __restrict__ k, const float* __restrict__ q, bf16_t* __restrict__ k_lp,
                                                  float m, long n4) {
  const float om = 1.0f - m;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    float4 a = reinterpret_cast<float4*>(k)[i];
    const float4 b = reinterpret_cast<const float4*>(q)[i];
    a.x = a.x * m + b.x * om; a.y = a.y * m + b.y * om; a.z = a.z * m + b.z * om; a.w = a.w * m + b.w * om;
    reinterpret_cast<float4*>(k)[i] = a;
    if (k_lp) {
      uint2 pk;
      pk.x = (uint32_t)f2bf(a.x) | ((uint32_t)f2bf(a.y) << 16);
      pk.y = (uint32_t)f2bf(a.z) | ((uint32_t)f2bf(a.w) << 16);
      reinterpret_cast<uint2*>(k_lp)[i] = pk;
    }
  }
}
int rmcl_ema(float* k, const float* q, void* k_lp, float m, long n, hipStream_t s) {
  RMCL_REQUIRE(n % 4 == 0, "ema: n%4");
  RMCL_LAUNCH(ema_kernel, dim3(std::min<long>(cdiv(n / 4, 256), 4096)), dim3(256), 0, s, k, q, (bf16_t*)k_lp, m, n / 4);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// queue[:, ptr + i] = keys[i, :]  for i < n  (objectives.py:244-246); queue [Pd, Kq] row-major
__global__ __launch_bounds__(256) void enqueue_kernel(float* __restrict__ queue, const float* __restrict__ keys, int n, int Pd, long Kq, long ptr) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= n * Pd) return;
  const int c = i / n, r = i % n;  // consecutive threads -> consecutive queue columns
  queue[(long)c * Kq + ptr + r] = keys[(long)r * Pd + c];
}
int rmcl_enqueue(float* queue, const float* keys, int n, int Pd, long Kq, long ptr, hipStream_t s) {
  RMCL_REQUIRE(ptr >= 0 && ptr + n <= Kq, "enqueue: block would run past the end of the queue (needs Kq % batch == 0)");
  RMCL_LAUNCH(enqueue_kernel, dim3(cdiv((long)n * Pd, 256)), dim3(256), 0, s, queue, keys, n, Pd, Kq, ptr);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// ---------------------------------------------------------------------------------------------
// Row L2 normalise (F.normalize eps 1e-12) forward / backward, ReLU/tanh backward multiplies
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(64) void l2norm_fwd_kernel(const float* __restrict__ z, float* __restrict__ q, float* __restrict__ nrm, int D, float eps,
                                                         float* __restrict__ q2) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) { const float v = z[(long)r * D + c]; s += v * v; }
  const float n = fmaxf(sqrtf(wave_sum(s)), eps);
  if (lane == 0 && nrm) nrm[r] = n;
  for (int c = lane; c < D; c += 64) {
    const float v = z[(long)r * D + c] / n;
    q[(long)r * D + c] = v;
    if (q2) q2[(long)r * D + c] = v;                         // (second copy: the caller's output next to the stash, no memcpy launch)
  }
}
int rmcl_l2norm_fwd(const float* z, float* q, float* nrm, int R, int D, float eps, hipStream_t s, float* q2) {
  RMCL_LAUNCH(l2norm_fwd_kernel, dim3(R), dim3(64), 0, s, z, q, nrm, D, eps, q2);
  RMCL_CHECK_LAUNCH();
  return 0;
}
// dz = (dq - q * (q . dq)) / nrm
__global__ __launch_bounds__(64) void l2norm_bwd_kernel(const float* __restrict__ dq, const float* __restrict__ q,
                                                        const float* __restrict__ nrm, float* __restrict__ dz, int D) {
  const int r = blockIdx.x, lane = threadIdx.x;
  float s = 0.f;
  for (int c = lane; c < D; c += 64) s += q[(long)r * D + c] * dq[(long)r * D + c];
  s = wave_sum(s);
  const float inv = 1.0f / nrm[r];
  for (int c = lane; c < D; c += 64) dz[(long)r * D + c] = (dq[(long)r * D + c] - q[(long)r * D + c] * s) * inv;
}
int rmcl_l2norm_bwd(const float* dq, const float* q, const float* nrm, float* dz, int R, int D, hipStream_t s) {
  RMCL_LAUNCH(l2norm_bwd_kernel, dim3(R), dim3(64), 0, s, dq, q, nrm, dz, D);
  RMCL_CHECK_LAUNCH();
  return 0;
}
// g *= (1 - y*y)   (tanh backward, y = tanh output)
__global__ __launch_bounds__(256) void tanh_bwd_kernel(float* __restrict__ g, const float* __restrict__ y, long n) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) g[i] *= (1.0f - y[i] * y[i]);
}
int rmcl_tanh_bwd(float* g, const float* y, long n, hipStream_t s) {
  RMCL_LAUNCH(tanh_bwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, g, y, n);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// f32 -> T cast of a flat arena (bf16 weight shadow refresh)
int rmcl_cast(const float* in, void* out, int dt, long n, hipStream_t s) { return rmcl_k_add_cast(in, nullptr, nullptr, out, dt, n, s); }

// ---------------------------------------------------------------------------------------------
// Fused multi-tensor AdamW over flat arenas (HF AdamW as used by vilt_utils.set_schedule :395-398):
//   m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr*sqrt(1-b2^t)/(1-b1^t) * m/(sqrt(v)+eps); p -= lr*wd*p
// Per-element lr scale / weight-decay come from a per-segment table (segments are 64-element
// aligned parameter tensors): seg_end[i] (exclusive element offset), seg_lr_mult[i], seg_wd[i].
// Optionally refreshes the bf16 shadow of p.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, bf16_t* __restrict__ p_lp, const long* __restrict__ seg_end,
                                                    const float* __restrict__ seg_lr_mult, const float* __restrict__ seg_wd, int nseg,
                                                    float lr, float b1, float b2, float eps, float bc1, float bc2s, long n4,
                                                    float grad_scale) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long e = i * 4;
    int lo = 0, hi = nseg - 1;  // first segment with seg_end > e
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (seg_end[mid] > e) hi = mid; else lo = mid + 1; }
    const float slr = lr * seg_lr_mult[lo], wd = seg_wd[lo];
    const float step = slr * bc2s / bc1;
    float4 pp = reinterpret_cast<float4*>(p)[i], gg = reinterpret_cast<const float4*>(g)[i];
    float4 mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    float P[4] = {pp.x, pp.y, pp.z, pp.w}, G[4] = {gg.x, gg.y, gg.z, gg.w}, Mo[4] = {mm.x, mm.y, mm.z, mm.w}, V[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float gj = G[j] * grad_scale;
      Mo[j] = Mo[j] * b1 + gj * (1.0f - b1);
      V[j] = V[j] * b2 + gj * gj * (1.0f - b2);
      P[j] = P[j] - step * Mo[j] / (sqrtf(V[j]) + eps);
      if (wd > 0.f) P[j] = P[j] - slr * wd * P[j];
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(P[0], P[1], P[2], P[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(Mo[0], Mo[1], Mo[2], Mo[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(V[0], V[1], V[2], V[3]);
    if (p_lp) {
      uint2 pk;
      pk.x = (uint32_t)f2bf(P[0]) | ((uint32_t)f2bf(P[1]) << 16);
      pk.y = (uint32_t)f2bf(P[2]) | ((uint32_t)f2bf(P[3]) << 16);
      reinterpret_cast<uint2*>(p_lp)[i] = pk;
    }
  }
}
int rmcl_adamw(float* p, const float* g, float* m, float* v, void* p_lp, const long* seg_end, const float* seg_lr_mult,
               const float* seg_wd, int nseg, float lr, float b1, float b2, float eps, int step, float grad_scale, long n,
               hipStream_t s) {
  RMCL_REQUIRE(n % 4 == 0 && nseg > 0 && step >= 1, "adamw: bad args");
  const float bc1 = (float)(1.0 - pow((double)b1, (double)step)), bc2s = (float)sqrt(1.0 - pow((double)b2, (double)step));
  RMCL_LAUNCH(adamw_kernel, dim3(std::min<long>(cdiv(n / 4, 256), 4096)), dim3(256), 0, s, p, g, m, v, (bf16_t*)p_lp, seg_end,
                     seg_lr_mult, seg_wd, nseg, lr, b1, b2, eps, bc1, bc2s, n / 4, grad_scale);
  RMCL_CHECK_LAUNCH();
  return 0;
}

// x[i] *= mask(seed, i)  (in place; with x pre-filled with ones this materialises a site's mask for the tests)
__global__ __launch_bounds__(256) void dropout_apply_kernel(float* __restrict__ x, long n, uint32_t dseed, uint32_t dthresh, float dinv) {
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) x[i] *= drop_scale(dseed, (uint32_t)i, dthresh, dinv);
}
// out[r, c] = in[r, c] * mask(seed, (r * row_mul) * cols + c): the dropout mask of DENSE row r * row_mul applied to compact row r (the cls-only
// tail under dropout draws the masks of the dense rows it stands for, so tail and dense block give the same numbers)
__global__ __launch_bounds__(256) void dropout_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int rows, int cols, long row_mul,
                                                           uint32_t dseed, uint32_t dthresh, float dinv) {
  const long n = (long)rows * cols;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
    const long r = i / cols, c = i - r * cols;
    out[i] = in[i] * drop_scale(dseed, (uint32_t)(r * row_mul * cols + c), dthresh, dinv);
  }
}
int rmcl_dropout_rows(const float* in, float* out, int rows, int cols, long row_mul, uint32_t dseed, uint32_t dthresh, float dinv, hipStream_t s) {
  if (rows <= 0 || cols <= 0) return 0;
  RMCL_LAUNCH(dropout_rows_kernel, dim3(std::min<long>(cdiv((long)rows * cols, 256), 1024)), dim3(256), 0, s, in, out, rows, cols, row_mul, dseed,
              dthresh, dinv);
  RMCL_CHECK_LAUNCH();
  return 0;
}
int rmcl_dropout_apply(float* x, long n, uint32_t dseed, uint32_t dthresh, float dinv, hipStream_t s) {
  if (n <= 0 || dthresh == 0) return 0;
  RMCL_LAUNCH(dropout_apply_kernel, dim3(std::min<long>(cdiv(n, 256), 4096)), dim3(256), 0, s, x, n, dseed, dthresh, dinv);
  RMCL_CHECK_LAUNCH();
  return 0;
}


// ---------------------------------------------------------------------------------------------
// Transposed bf16 shadows of the four weight matrices of every layer (same arena offsets, [cols][rows] instead of
// [rows][cols]): the data-gradient GEMMs dX = dY W then read W^T rows with k contiguous ([rows][K] x [cols][K] form, plain
// ds_read_b128 fragments) instead of W through transposed LDS reads - measured 10-15 % per GEMM.  Refreshed once per
// optimizer step (170 MB read + 170 MB written).  64x64 tiles through LDS, 8-byte accesses on both sides.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void weight_transpose_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst, long layer0, long stride,
                                                               long o0, long o1, long o2, long o3, int r0, int c0, int r1, int c1, int r2, int c2,
                                                               int r3, int c3, int tiles_per_layer) {
  __shared__ bf16_t tile[64][68];
  const int l = blockIdx.x / tiles_per_layer;
  int t = blockIdx.x % tiles_per_layer;
  const long offs[4] = {o0, o1, o2, o3};
  const int rs[4] = {r0, r1, r2, r3}, cs[4] = {c0, c1, c2, c3};
  int mi = 0;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int nt = (rs[q] / 64) * (cs[q] / 64);
    if (mi == q && t >= nt) { t -= nt; mi = q + 1; }
  }
  const int R = rs[mi], Cn = cs[mi], tc = t % (Cn / 64), tr = t / (Cn / 64);
  const bf16_t* S = src + layer0 + (long)l * stride + offs[mi];
  bf16_t* Dp = dst + layer0 + (long)l * stride + offs[mi];
  const int x4 = (threadIdx.x & 15) * 4, y = threadIdx.x >> 4;          // 16 x 16 threads, 4 elements each, 4 row passes
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int r = y + 16 * p;
    const uint2 v = *reinterpret_cast<const uint2*>(S + (long)(tr * 64 + r) * Cn + tc * 64 + x4);
    tile[r][x4] = (bf16_t)(v.x & 0xffff); tile[r][x4 + 1] = (bf16_t)(v.x >> 16);
    tile[r][x4 + 2] = (bf16_t)(v.y & 0xffff); tile[r][x4 + 3] = (bf16_t)(v.y >> 16);
  }
  __syncthreads();
#pragma unroll
  for (int p = 0; p < 4; ++p) {
    const int cc = y + 16 * p;                                        // output row = source column
    uint2 v;
    v.x = (uint32_t)tile[x4][cc] | ((uint32_t)tile[x4 + 1][cc] << 16);
    v.y = (uint32_t)tile[x4 + 2][cc] | ((uint32_t)tile[x4 + 3][cc] << 16);
    *reinterpret_cast<uint2*>(Dp + (long)(tc * 64 + cc) * R + tr * 64 + x4) = v;
  }
}
int rmcl_weight_transpose(const bf16_t* src, bf16_t* dst, long layer0, long stride, int layers, const long* offs, const int* rows, const int* cols,
                          hipStream_t s) {
  int tiles = 0;
  for (int q = 0; q < 4; ++q) {
    RMCL_REQUIRE(rows[q] % 64 == 0 && cols[q] % 64 == 0, "weight_transpose: dims must be multiples of 64");
    tiles += (rows[q] / 64) * (cols[q] / 64);
  }
  RMCL_LAUNCH(weight_transpose_kernel, dim3(layers * tiles), dim3(256), 0, s, src, dst, layer0, stride, offs[0], offs[1], offs[2], offs[3], rows[0],
              cols[0], rows[1], cols[1], rows[2], cols[2], rows[3], cols[3], tiles);
  RMCL_CHECK_LAUNCH();
  return 0;
}
