// The fp32-accuracy compute mode ("fp32x3") of the UniMM-UL hot path on gfx950.
//
// The reference's dense-annotation fine-tune runs WITHOUT autocast (dense_annotation_finetuning.py:253: fp32 end to
// end) and BASELINE.json's north_star gates fp32 results at 1e-3.  CDNA4 has no fast fp32 matrix path
// (v_mfma_f32_16x16x4_f32 runs at 1/16 of the bf16 rate), so this mode keeps the bf16 MFMA GEMM kernels and feeds them
// SPLIT operands: x = hi + lo with hi = bf16(x), lo = bf16(x - hi), and
//     x . w  ~=  hi(x) hi(w) + lo(x) hi(w) + hi(x) lo(w)          (the dropped lo.lo term is 2^-16 relative)
// accumulated in fp32 by the same kernel -- as ONE product over a reduction axis that is three planes long:
//     activation operand  X3[M, 3 Kp] = [ hi(X) | lo(X) | hi(X) ]        ("x-type" planes)
//     weight operand      W3[N, 3 Kp] = [ hi(W) | hi(W) | lo(W) ]        ("w-type" planes)
//     X3 . W3^T = hi hi + lo hi + hi lo.         Kp = K rounded up to 64, padding columns are zero.
// Weight gradients dW += dY^T X become three problems of the grouped TN launch over column planes of the x-type
// buffers ((hi, hi), (lo, hi), (hi, lo)), accumulating into the same fp32 gradient.  Everything between two GEMMs is
// fp32: the GEMMs write fp32, and the kernels of this file turn fp32 results into the next split operand (with the
// elementwise op that the bf16 path fuses into its epilogues), run LayerNorm / embedding / loss backward on fp32
// gradients, and compute the attention cores in exact fp32 on v_mfma_f32_16x16x4_f32 (~1.3 % of the model's FLOPs; the
// vector-ALU kernels of the first version stay behind unimm_x3_attn_set_impl(0)).  Same dropout counters, mask words, row
// maps and device-side row counts as the bf16 kernels.
#include "common.h"
#include "rows.h"

namespace {

__device__ __forceinline__ void split2(float x, float& hi, float& lo) {
  hi = bf2f(f2bf(x));
  lo = x - hi;            // exact in fp32; rounded to bf16 when packed
}

// ------------------------------------------------------------------------------------------------
// fp32 -> split operand, with the elementwise op of the bf16 path's GEMM epilogues in front
// ------------------------------------------------------------------------------------------------
struct SplitParams {
  const float* a; const float* b; float* out32; bf16_t* out3;
  long rows; int cols, cp, lda, ldb, ld32, op, wtype;
};

__device__ __forceinline__ float ew_op(int op, float a, float b) {
  switch (op) {
    case UNIMM_X3_ADD: return a + b;
    case UNIMM_X3_GELU: return gelu_erf(a);
    case UNIMM_X3_MUL_DGELU: return a * gelu_erf_grad(b);
    default: return a;
  }
}

__global__ __launch_bounds__(256) void x3_split_kernel(SplitParams p) {
  const int nch = p.cp >> 3;                                  // 8-column chunks per plane (cp % 64 == 0)
  const long total = p.rows * nch;
  const bool veca = (p.lda & 3) == 0 && (((uintptr_t)p.a) & 15) == 0;
  const bool vecb = p.b == nullptr || ((p.ldb & 3) == 0 && (((uintptr_t)p.b) & 15) == 0);
  const bool veco = p.out32 == nullptr || ((p.ld32 & 3) == 0 && (((uintptr_t)p.out32) & 15) == 0);
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const long r = i / nch;
    const int c0 = (int)(i - r * nch) * 8;
    float y[8];
    const float* ar = p.a + (size_t)r * p.lda;
    const float* br = p.b != nullptr ? p.b + (size_t)r * p.ldb : nullptr;
    if (c0 + 8 <= p.cols && veca && vecb) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(ar + c0), a1 = *reinterpret_cast<const f32x4*>(ar + c0 + 4);
      f32x4 b0 = {0.f, 0.f, 0.f, 0.f}, b1 = {0.f, 0.f, 0.f, 0.f};
      if (br != nullptr) { b0 = *reinterpret_cast<const f32x4*>(br + c0); b1 = *reinterpret_cast<const f32x4*>(br + c0 + 4); }
#pragma unroll
      for (int e = 0; e < 4; ++e) { y[e] = ew_op(p.op, a0[e], b0[e]); y[4 + e] = ew_op(p.op, a1[e], b1[e]); }
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        y[e] = c < p.cols ? ew_op(p.op, ar[c], br != nullptr ? br[c] : 0.f) : 0.f;
      }
    }
    if (p.out32 != nullptr) {
      float* o = p.out32 + (size_t)r * p.ld32;
      if (c0 + 8 <= p.cols && veco) {
        *reinterpret_cast<f32x4*>(o + c0) = f32x4{y[0], y[1], y[2], y[3]};
        *reinterpret_cast<f32x4*>(o + c0 + 4) = f32x4{y[4], y[5], y[6], y[7]};
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e)
          if (c0 + e < p.cols) o[c0 + e] = y[e];
      }
    }
    if (p.out3 != nullptr) {
      float hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split2(y[e], hi[e], lo[e]);
      const u32x4 H = u32x4{pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3]), pack2bf(hi[4], hi[5]), pack2bf(hi[6], hi[7])};
      const u32x4 L = u32x4{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(lo[4], lo[5]), pack2bf(lo[6], lo[7])};
      bf16_t* o = p.out3 + (size_t)r * 3 * p.cp + c0;
      *reinterpret_cast<u32x4*>(o) = H;
      *reinterpret_cast<u32x4*>(o + p.cp) = p.wtype ? H : L;
      *reinterpret_cast<u32x4*>(o + 2 * p.cp) = p.wtype ? L : H;
    }
  }
}

// transposed w-type split of a weight matrix: src fp32 [R, C] (row stride lds) -> dst bf16 [C, 3 Rp]:
// dst[c][p * Rp + r] = plane_p(src[r][c]); columns r in [R, Rp) are left untouched (the caller zero-fills once).
__global__ __launch_bounds__(256) void x3_split_wt_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, int R, int C,
                                                           int lds, int Rp) {
  __shared__ float tile[32][33];
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
  for (int k = ty; k < 32; k += 8) {
    const int r = r0 + k, c = c0 + tx;
    tile[k][tx] = (r < R && c < C) ? src[(size_t)r * lds + c] : 0.f;
  }
  __syncthreads();
  for (int k = ty; k < 32; k += 8) {
    const int c = c0 + k, r = r0 + tx;
    if (c < C && r < R) {
      float hi, lo;
      split2(tile[tx][k], hi, lo);
      bf16_t* o = dst + (size_t)c * 3 * Rp + r;
      o[0] = f2bf(hi); o[Rp] = f2bf(hi); o[2 * Rp] = f2bf(lo);
    }
  }
}

__device__ __forceinline__ void store_row_split3(bf16_t* __restrict__ p, int cp, int H, int lane, const Row8& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c * 8 < H) {
      float hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) split2(r.v[i][e], hi[e], lo[e]);
      const u32x4 Hh = u32x4{pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3]), pack2bf(hi[4], hi[5]), pack2bf(hi[6], hi[7])};
      const u32x4 Ll = u32x4{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(lo[4], lo[5]), pack2bf(lo[6], lo[7])};
      *reinterpret_cast<u32x4*>(p + c * 8) = Hh;
      *reinterpret_cast<u32x4*>(p + cp + c * 8) = Ll;
      *reinterpret_cast<u32x4*>(p + 2 * cp + c * 8) = Hh;
    }
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm forward (layernorm_fwd_kernel of rowops.hip, same arithmetic) that also writes the normalised row as an x-type
// split operand: the next GEMM's input without a pass of its own over the fp32 row.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float* __restrict__ y32,
                                                               bf16_t* __restrict__ y3, float* __restrict__ mean_o,
                                                               float* __restrict__ rstd_o, int M, int H, float eps, DropoutArg drop) {
  drop_resolve(drop);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  Row8 g, b;
  load_vec_f32(gamma, H, lane, g);
  load_vec_f32(beta, H, lane, b);
  Row8 xv;
  if (wave < M) load_vec_f32(x + (size_t)wave * H, H, lane, xv);
  for (int row = wave; row < M; row += nwaves) {
    Row8 nx;
    const int nrow = row + nwaves;
    if (nrow < M) load_vec_f32(x + (size_t)nrow * H, H, lane, nx);      // next row in flight while this one is reduced
    float mean, rstd;
    row_stats(xv, H, mean, rstd, eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const uint32_t kb = drop.thr != 0u ? drop_bits8(drop, (uint32_t)row, (uint32_t)H, (uint32_t)((lane + 64 * i) * 8)) : 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = (xv.v[i][j] - mean) * rstd * g.v[i][j] + b.v[i][j];
        if (drop.thr != 0u) v = ((kb >> j) & 1u) ? v * drop.scale : 0.f;
        xv.v[i][j] = v;
      }
    }
    if (y32 != nullptr) store_row_f32(y32 + (size_t)row * H, H, lane, xv);
    store_row_split3(y3 + (size_t)row * 3 * H, H, H, lane, xv);
    if (lane == 0 && mean_o != nullptr) { mean_o[row] = mean; rstd_o[row] = rstd; }
    if (nrow < M) xv = nx;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward on fp32 gradients (same arithmetic and partials layout as layernorm_bwd_kernel of rowops.hip):
// dx32 = gradient w.r.t. the pre-LayerNorm sum (fp32, the residual branch), dxd3 = its dropout-masked copy as an
// x-type split operand (the dY of the dense branch: dgrad GEMM operand and weight-gradient operand).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void x3_layernorm_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ x,
                                                               const float* __restrict__ mean_i, const float* __restrict__ rstd_i,
                                                               const float* __restrict__ gamma, float* __restrict__ dx32,
                                                               bf16_t* __restrict__ dxd3, float* __restrict__ partials, int M, int H,
                                                               DropoutArg drop, DropoutArg out_drop,
                                                               const int32_t* __restrict__ m_dev) {
  __shared__ float red[4 * 1024];
  drop_resolve(drop);
  drop_resolve(out_drop);
  if (m_dev != nullptr) M = min(M, m_dev[0]);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  const int nwaves = gridDim.x * 4;
  Row8 g;
  load_vec_f32(gamma, H, lane, g);
  Row8 dg, db, dbias;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg.v[i][j] = 0.f; db.v[i][j] = 0.f; dbias.v[i][j] = 0.f; }
  const float invH = 1.0f / (float)H;
  for (int row = wave; row < M; row += nwaves) {
    Row8 dyv, xv;
    load_vec_f32(dy + (size_t)row * H, H, lane, dyv);
    load_vec_f32(x + (size_t)row * H, H, lane, xv);
    const float mean = mean_i[row], rstd = rstd_i[row];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const uint32_t kbo = out_drop.thr != 0u ? drop_bits8(out_drop, (uint32_t)row, (uint32_t)H, (uint32_t)((lane + 64 * i) * 8)) : 0u;
      const bool in = (lane + 64 * i) * 8 < H;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = dyv.v[i][j];
        if (out_drop.thr != 0u) d = ((kbo >> j) & 1u) ? d * out_drop.scale : 0.f;
        const float xh = in ? (xv.v[i][j] - mean) * rstd : 0.f;
        const float gg = d * g.v[i][j];
        dg.v[i][j] += d * xh;
        db.v[i][j] += d;
        s1 += gg;
        s2 += gg * xh;
        dyv.v[i][j] = gg;
        xv.v[i][j] = xh;
      }
    }
    s1 = wave_sum(s1) * invH;
    s2 = wave_sum(s2) * invH;
    Row8 dd;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const uint32_t kb = drop.thr != 0u ? drop_bits8(drop, (uint32_t)row, (uint32_t)H, (uint32_t)((lane + 64 * i) * 8)) : 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rstd * (dyv.v[i][j] - s1 - xv.v[i][j] * s2);
        dyv.v[i][j] = v;
        float vd = v;
        if (drop.thr != 0u) vd = ((kb >> j) & 1u) ? v * drop.scale : 0.f;
        dd.v[i][j] = vd;
        dbias.v[i][j] += vd;
      }
    }
    if (dx32 != nullptr) store_row_f32(dx32 + (size_t)row * H, H, lane, dyv);
    if (dxd3 != nullptr) store_row_split3(dxd3 + (size_t)row * 3 * H, H, H, lane, dd);
  }
  for (int qn = 0; qn < 3; ++qn) {
    const Row8& src = qn == 0 ? dg : (qn == 1 ? db : dbias);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = (lane + 64 * i) * 8 + j;
        if (col < H) red[wv * 1024 + col] = src.v[i][j];
      }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256)
      partials[((size_t)blockIdx.x * 3 + qn) * H + col] = red[col] + red[1024 + col] + red[2048 + col] + red[3072 + col];
  }
}

// ------------------------------------------------------------------------------------------------
// loss backward straight into x-type split operands (dlogits is both the dgrad GEMM's operand and the dY of the
// decoder's weight gradient)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void store8_split3(bf16_t* o, int cp, const float* v) {
  float hi[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split2(v[e], hi[e], lo[e]);
  const u32x4 H = u32x4{pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3]), pack2bf(hi[4], hi[5]), pack2bf(hi[6], hi[7])};
  const u32x4 L = u32x4{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(lo[4], lo[5]), pack2bf(lo[6], lo[7])};
  *reinterpret_cast<u32x4*>(o) = H;
  *reinterpret_cast<u32x4*>(o + cp) = L;
  *reinterpret_cast<u32x4*>(o + 2 * cp) = H;
}

__global__ __launch_bounds__(256) void x3_lm_loss_bwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                             const int32_t* __restrict__ weights, const float* __restrict__ lse_i,
                                                             const float* __restrict__ g, float inv_denom, bf16_t* __restrict__ out3,
                                                             int V, int ld, int cp, float clamp_min,
                                                             const int32_t* __restrict__ n_dev, const float* __restrict__ inv_dev) {
  const int row = blockIdx.x;
  bf16_t* dz = out3 + (size_t)row * 3 * cp;
  const bool live = n_dev == nullptr || row < n_dev[0];
  if (inv_dev != nullptr) inv_denom = inv_dev[0];
  const float* z = logits + (size_t)row * ld;
  float coef = 0.f, lse = 0.f;
  int y = -1;
  if (live) {
    y = labels[row];
    const int w = weights[row];
    lse = lse_i[row];
    if (y >= 0) {
      const float gs = g[0] * inv_denom;
      if (w > 0) coef = gs * (float)w;
      else if (w == -1) {
        const float py = expf(z[y] - lse);
        const float om = 1.0f - py;
        coef = om >= clamp_min ? -gs * py / om : 0.f;
      }
    }
  }
  for (int i = threadIdx.x; i < (cp >> 3); i += 256) {
    const int c0 = 8 * i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      v[e] = (live && coef != 0.f && c < V) ? coef * (expf(z[c] - lse) - (c == y ? 1.0f : 0.0f)) : 0.f;
    }
    store8_split3(dz + c0, cp, v);
  }
}

__global__ __launch_bounds__(256) void x3_kl_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                             const int32_t* __restrict__ label, const float* __restrict__ lse_i,
                                                             const float* __restrict__ g, float inv_denom, bf16_t* __restrict__ out3,
                                                             int C, int ld, int cp, const float* __restrict__ inv_dev) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  if (inv_dev != nullptr) inv_denom = inv_dev[0];
  const float* z = pred + (size_t)row * ld;
  const float* t = target + (size_t)row * C;
  bf16_t* dz = out3 + (size_t)row * 3 * cp;
  const bool on = label[row] == 1;
  float ts = 0.f;
  if (on)
    for (int i = threadIdx.x; i < C; i += 256) ts += t[i];
  ts = wave_sum(ts);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = ts;
  __syncthreads();
  ts = red[0] + red[1] + red[2] + red[3];
  const float gs = on ? g[0] * inv_denom : 0.f;
  const float lse = lse_i[row];
  for (int i = threadIdx.x; i < (cp >> 3); i += 256) {
    const int c0 = 8 * i;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = c0 + e;
      v[e] = (on && c < C) ? gs * (expf(z[c] - lse) * ts - t[c]) : 0.f;
    }
    store8_split3(dz + c0, cp, v);
  }
}

// dst[idx[r], :] += src[r, :], fp32 both (idx unique)
__global__ void x3_rows_add_kernel(float* __restrict__ dst, const int32_t* __restrict__ idx, const float* __restrict__ src, int n,
                                   int H, int ldd) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)n * H) return;
  const int r = (int)(i / H), c = (int)(i - (long)r * H);
  dst[(size_t)idx[r] * ldd + c] += src[i];
}

// ------------------------------------------------------------------------------------------------
// Attention cores in fp32 (models/vilbert_dialog.py:390-410, :519-539, :681-721 and their autograd), VECTOR-ALU version
// (the first one; kept for A/B runs, the product path is the matrix-instruction version below).
// One lane owns 64 of a row's D dimensions (D = 64: a lane per row; D = 128: two neighbouring lanes per row whose
// partial dot products meet in one cross-lane add); the other side's rows are staged in LDS in chunks of 32 and read
// as broadcasts.  Online softmax over the keys in the forward; the backward recomputes P from the saved log-sum-exp,
// dQ with a lane per query, dK / dV with a lane per key.  Masks, additive -10000, dropout counters, variable-length
// offsets: exactly the bf16 kernels' conventions (csrc/attention.hip).
// ------------------------------------------------------------------------------------------------
struct AttnF32 {
  const float* q; const float* k; const float* v; const float* o; const float* dout;
  float* out; float* lse; float* delta; float* dq; float* dk; float* dv;
  const uint32_t* mask;
  const int* q_off; const int* q_len; const int* k_off; const int* k_len;
  const int* order;      // or NULL: the sequences in the order the workgroups take them (unimm_attn_args.order; the matrix kernels)
  int B, H, Tq, Tk;
  int ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int mqs, mbs;
  float scale;
  DropoutArg drop;
  // optional x-type split outputs (planes of `cp3` columns, row stride ld3): the forward's context, the backward's dQ / dK / dV
  bf16_t* o3; bf16_t* dq3; bf16_t* dk3; bf16_t* dv3;
  int ld3, cp3;
};

#ifndef UNIMM_X3_ATTN_BWD_DH
#define UNIMM_X3_ATTN_BWD_DH 32
#endif
#ifndef UNIMM_X3_ATTN_FWD_DH
#define UNIMM_X3_ATTN_FWD_DH 64
#endif
constexpr int XA_T = 128;     // threads per workgroup
constexpr int XA_C = 32;      // staged rows per chunk (= one mask word)
// A row of D dimensions is split over LPR = D / DH neighbouring lanes (DH = 64 or 32 dimensions per lane: 32 halves the
// per-lane operand / accumulator registers, which is what lets the backward kernels keep two waves per SIMD); the partial
// dot products meet in log2(LPR) cross-lane adds.  Staged rows are stored as segments of DH + 4 floats, so that the lanes of
// one row read different banks.

template <int DH>
__device__ __forceinline__ void xa_load(const float* __restrict__ g, float* r) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(g + c);
    r[c] = t[0]; r[c + 1] = t[1]; r[c + 2] = t[2]; r[c + 3] = t[3];
  }
}
template <int DH>
__device__ __forceinline__ void xa_store(float* __restrict__ g, const float* r, float s) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) *reinterpret_cast<f32x4*>(g + c) = f32x4{r[c] * s, r[c + 1] * s, r[c + 2] * s, r[c + 3] * s};
}
template <int DH>
__device__ __forceinline__ float xa_dot(const float* r, const float* __restrict__ lds) {
  float d0 = 0.f, d1 = 0.f;
#pragma unroll
  for (int c = 0; c < DH; c += 8) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(lds + c), b = *reinterpret_cast<const f32x4*>(lds + c + 4);
    d0 = fmaf(r[c], a[0], d0); d0 = fmaf(r[c + 1], a[1], d0); d0 = fmaf(r[c + 2], a[2], d0); d0 = fmaf(r[c + 3], a[3], d0);
    d1 = fmaf(r[c + 4], b[0], d1); d1 = fmaf(r[c + 5], b[1], d1); d1 = fmaf(r[c + 6], b[2], d1); d1 = fmaf(r[c + 7], b[3], d1);
  }
  return d0 + d1;
}
template <int DH>
__device__ __forceinline__ void xa_axpy(float* acc, float a, const float* __restrict__ lds) {
#pragma unroll
  for (int c = 0; c < DH; c += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(lds + c);
    acc[c] = fmaf(a, t[0], acc[c]); acc[c + 1] = fmaf(a, t[1], acc[c + 1]);
    acc[c + 2] = fmaf(a, t[2], acc[c + 2]); acc[c + 3] = fmaf(a, t[3], acc[c + 3]);
  }
}
// stage rows [r0, r0 + XA_C) (zeros past `len`) of a [.., ld] fp32 matrix, D columns from column `col0`, into LDS
template <int LPR> __device__ __forceinline__ float xa_rowsum(float v) {       // sum over the LPR lanes of a row
#pragma unroll
  for (int o = 1; o < LPR; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}
template <int D, int DH>
__device__ __forceinline__ void xa_stage(float* __restrict__ lds, const float* __restrict__ g, int row_base, int r0, int len, int ld,
                                         int col0) {
  constexpr int RS = (D / DH) * (DH + 4);
  for (int i = threadIdx.x; i < XA_C * (D / 4); i += XA_T) {
    const int r = i / (D / 4), c = (i - r * (D / 4)) * 4;
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    if (r0 + r < len) t = *reinterpret_cast<const f32x4*>(g + (size_t)(row_base + r0 + r) * ld + col0 + c);
    *reinterpret_cast<f32x4*>(lds + r * RS + (c / DH) * (DH + 4) + (c % DH)) = t;
  }
}
__device__ __forceinline__ bool xa_keep(const DropoutArg& d, uint32_t hrow, uint32_t Tk, uint32_t key) {
  if (d.thr == 0u) return true;
  const uint32_t w = drop_word(d, drop_wbase(hrow, Tk, key));
  return ((key & 1u) ? (w >> 16) : (w & 0xffffu)) >= (d.thr >> 16);
}

template <int D, int DH>
__global__ __launch_bounds__(XA_T) void x3_attn_fwd_kernel(AttnF32 p) {
  constexpr int LPR = D / DH, RPW = XA_T / LPR, HS = DH + 4, RS = LPR * HS;
  __shared__ __attribute__((aligned(16))) float Ks[XA_C * RS];
  __shared__ __attribute__((aligned(16))) float Vs[XA_C * RS];
  drop_resolve(p.drop);
  const int b = blockIdx.z, head = blockIdx.y;
  const int qlen = p.q_len != nullptr ? p.q_len[b] : p.Tq, qoff = p.q_off != nullptr ? p.q_off[b] : b * p.Tq;
  const int klen = p.k_len != nullptr ? p.k_len[b] : p.Tk, koff = p.k_off != nullptr ? p.k_off[b] : b * p.Tk;
  const int r0 = blockIdx.x * RPW;
  if (r0 >= qlen) return;                                  // workgroup-uniform
  const int rl = threadIdx.x / LPR, half = threadIdx.x % LPR;
  const bool valid = r0 + rl < qlen;
  const int qr = valid ? r0 + rl : qlen - 1;
  float qv[DH], acc[DH];
  xa_load<DH>(p.q + (size_t)(qoff + qr) * p.ldq + head * D + half * DH, qv);
#pragma unroll
  for (int c = 0; c < DH; ++c) acc[c] = 0.f;
  float m = -INFINITY, l = 0.f;
  const uint32_t* mrow = p.mask + (size_t)b * p.mbs + (size_t)qr * p.mqs;
  const uint32_t hrow = ((uint32_t)b * p.H + head) * p.Tq + (uint32_t)qr;
  for (int k0 = 0; k0 < klen; k0 += XA_C) {
    __syncthreads();
    xa_stage<D, DH>(Ks, p.k, koff, k0, klen, p.ldk, head * D);
    xa_stage<D, DH>(Vs, p.v, koff, k0, klen, p.ldv, head * D);
    __syncthreads();
    const uint32_t mw = mrow[k0 >> 5];
    const int nk = min(XA_C, klen - k0);
    for (int j = 0; j < nk; ++j) {
      float d = xa_dot<DH>(qv, Ks + j * RS + half * HS);
      d = xa_rowsum<LPR>(d);
      const float sv = d * p.scale + (((mw >> j) & 1u) ? 0.f : -10000.0f);
      if (sv > m) {
        const float corr = __expf(m - sv);                 // first key: exp(-inf) = 0
        l *= corr;
#pragma unroll
        for (int c = 0; c < DH; ++c) acc[c] *= corr;
        m = sv;
      }
      const float e = __expf(sv - m);
      l += e;
      const float pe = xa_keep(p.drop, hrow, (uint32_t)p.Tk, (uint32_t)(k0 + j)) ? e : 0.f;
      xa_axpy<DH>(acc, pe, Vs + j * RS + half * HS);
    }
  }
  if (!valid) return;
  const float inv = l > 0.f ? (p.drop.thr != 0u ? p.drop.scale : 1.0f) / l : 0.f;
  xa_store<DH>(p.out + (size_t)(qoff + qr) * p.ldo + head * D + half * DH, acc, inv);
  if (p.lse != nullptr && half == 0) p.lse[((size_t)b * p.H + head) * p.Tq + qr] = m + __logf(l);
}

// dQ (+ delta = rowsum(dO o O)): a lane per query, keys staged
template <int D, int DH>
__global__ __launch_bounds__(XA_T) void x3_attn_bwd_dq_kernel(AttnF32 p) {
  constexpr int LPR = D / DH, RPW = XA_T / LPR, HS = DH + 4, RS = LPR * HS;
  __shared__ __attribute__((aligned(16))) float Ks[XA_C * RS];
  __shared__ __attribute__((aligned(16))) float Vs[XA_C * RS];
  drop_resolve(p.drop);
  const int b = blockIdx.z, head = blockIdx.y;
  const int qlen = p.q_len != nullptr ? p.q_len[b] : p.Tq, qoff = p.q_off != nullptr ? p.q_off[b] : b * p.Tq;
  const int klen = p.k_len != nullptr ? p.k_len[b] : p.Tk, koff = p.k_off != nullptr ? p.k_off[b] : b * p.Tk;
  const int r0 = blockIdx.x * RPW;
  if (r0 >= qlen) return;
  const int rl = threadIdx.x / LPR, half = threadIdx.x % LPR;
  const bool valid = r0 + rl < qlen;
  const int qr = valid ? r0 + rl : qlen - 1;
  float qv[DH], dov[DH], dq[DH];
  xa_load<DH>(p.q + (size_t)(qoff + qr) * p.ldq + head * D + half * DH, qv);
  xa_load<DH>(p.dout + (size_t)(qoff + qr) * p.lddo + head * D + half * DH, dov);
  float delta = 0.f;
  {
    const float* og = p.o + (size_t)(qoff + qr) * p.ldo + head * D + half * DH;
#pragma unroll
    for (int c = 0; c < DH; c += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(og + c);
      delta = fmaf(dov[c], t[0], delta); delta = fmaf(dov[c + 1], t[1], delta);
      delta = fmaf(dov[c + 2], t[2], delta); delta = fmaf(dov[c + 3], t[3], delta);
    }
    delta = xa_rowsum<LPR>(delta);
  }
  const size_t li = ((size_t)b * p.H + head) * p.Tq + qr;
  const float lse = p.lse[li];
  if (valid && half == 0) p.delta[li] = delta;
#pragma unroll
  for (int c = 0; c < DH; ++c) dq[c] = 0.f;
  const uint32_t* mrow = p.mask + (size_t)b * p.mbs + (size_t)qr * p.mqs;
  const uint32_t hrow = ((uint32_t)b * p.H + head) * p.Tq + (uint32_t)qr;
  const float dsc = p.drop.thr != 0u ? p.drop.scale : 1.0f;
  for (int k0 = 0; k0 < klen; k0 += XA_C) {
    __syncthreads();
    xa_stage<D, DH>(Ks, p.k, koff, k0, klen, p.ldk, head * D);
    xa_stage<D, DH>(Vs, p.v, koff, k0, klen, p.ldv, head * D);
    __syncthreads();
    const uint32_t mw = mrow[k0 >> 5];
    const int nk = min(XA_C, klen - k0);
    for (int j = 0; j < nk; ++j) {
      const float* kr = Ks + j * RS + half * HS;
      float d = xa_dot<DH>(qv, kr);
      float dpv = xa_dot<DH>(dov, Vs + j * RS + half * HS);
      d = xa_rowsum<LPR>(d); dpv = xa_rowsum<LPR>(dpv);
      const float pr = __expf(d * p.scale + (((mw >> j) & 1u) ? 0.f : -10000.0f) - lse);
      const float dP = xa_keep(p.drop, hrow, (uint32_t)p.Tk, (uint32_t)(k0 + j)) ? dpv * dsc : 0.f;
      const float dS = pr * (dP - delta) * p.scale;
      xa_axpy<DH>(dq, dS, kr);
    }
  }
  if (valid) xa_store<DH>(p.dq + (size_t)(qoff + qr) * p.lddq + head * D + half * DH, dq, 1.0f);
}

// dK, dV: a lane per key, queries staged (Q, dO rows + their lse / delta / mask words)
template <int D, int DH>
__global__ __launch_bounds__(XA_T) void x3_attn_bwd_dkv_kernel(AttnF32 p) {
  constexpr int LPR = D / DH, RPW = XA_T / LPR, HS = DH + 4, RS = LPR * HS;
  __shared__ __attribute__((aligned(16))) float Qs[XA_C * RS];
  __shared__ __attribute__((aligned(16))) float Os[XA_C * RS];
  __shared__ float Ls[XA_C], Ds[XA_C];
  __shared__ uint32_t Ms[XA_C * 8];
  drop_resolve(p.drop);
  const int b = blockIdx.z, head = blockIdx.y;
  const int qlen = p.q_len != nullptr ? p.q_len[b] : p.Tq, qoff = p.q_off != nullptr ? p.q_off[b] : b * p.Tq;
  const int klen = p.k_len != nullptr ? p.k_len[b] : p.Tk, koff = p.k_off != nullptr ? p.k_off[b] : b * p.Tk;
  const int r0 = blockIdx.x * RPW;
  if (r0 >= klen) return;
  const int rl = threadIdx.x / LPR, half = threadIdx.x % LPR;
  const bool valid = r0 + rl < klen;
  const int key = valid ? r0 + rl : klen - 1;
  float kv[DH], vv[DH], dk[DH], dv[DH];
  xa_load<DH>(p.k + (size_t)(koff + key) * p.ldk + head * D + half * DH, kv);
  xa_load<DH>(p.v + (size_t)(koff + key) * p.ldv + head * D + half * DH, vv);
#pragma unroll
  for (int c = 0; c < DH; ++c) { dk[c] = 0.f; dv[c] = 0.f; }
  const int nw = (p.Tk + 31) >> 5;
  const uint32_t hbase = ((uint32_t)b * p.H + head) * p.Tq;
  const float dsc = p.drop.thr != 0u ? p.drop.scale : 1.0f;
  const size_t lbase = ((size_t)b * p.H + head) * p.Tq;
  for (int q0 = 0; q0 < qlen; q0 += XA_C) {
    __syncthreads();
    xa_stage<D, DH>(Qs, p.q, qoff, q0, qlen, p.ldq, head * D);
    xa_stage<D, DH>(Os, p.dout, qoff, q0, qlen, p.lddo, head * D);
    const int nq = min(XA_C, qlen - q0);
    if (threadIdx.x < nq) { Ls[threadIdx.x] = p.lse[lbase + q0 + threadIdx.x]; Ds[threadIdx.x] = p.delta[lbase + q0 + threadIdx.x]; }
    for (int i = threadIdx.x; i < nq * nw; i += XA_T) {
      const int r = i / nw, w = i - r * nw;
      Ms[r * 8 + w] = p.mask[(size_t)b * p.mbs + (size_t)(q0 + r) * p.mqs + w];
    }
    __syncthreads();
    for (int j = 0; j < nq; ++j) {
      const float* qr_ = Qs + j * RS + half * HS;
      const float* or_ = Os + j * RS + half * HS;
      float d = xa_dot<DH>(kv, qr_);
      float dpv = xa_dot<DH>(vv, or_);
      d = xa_rowsum<LPR>(d); dpv = xa_rowsum<LPR>(dpv);
      const uint32_t mw = Ms[j * 8 + (key >> 5)];
      const float pr = __expf(d * p.scale + (((mw >> (key & 31)) & 1u) ? 0.f : -10000.0f) - Ls[j]);
      const bool keep = xa_keep(p.drop, hbase + (uint32_t)(q0 + j), (uint32_t)p.Tk, (uint32_t)key);
      const float pd = keep ? pr * dsc : 0.f;
      const float dP = keep ? dpv * dsc : 0.f;
      const float dS = pr * (dP - Ds[j]) * p.scale;
      xa_axpy<DH>(dv, pd, or_);
      xa_axpy<DH>(dk, dS, qr_);
    }
  }
  if (!valid) return;
  xa_store<DH>(p.dk + (size_t)(koff + key) * p.lddk + head * D + half * DH, dk, 1.0f);
  xa_store<DH>(p.dv + (size_t)(koff + key) * p.lddv + head * D + half * DH, dv, 1.0f);
}

// ------------------------------------------------------------------------------------------------
// The same attention cores on the fp32 matrix instruction (v_mfma_f32_16x16x4_f32: A[m = lane % 16][k = lane / 16],
// B[k = lane / 16][n = lane % 16], D[m = 4 (lane / 16) + i][n = lane % 16], i = 0..3).  Its peak equals the packed fp32
// vector rate, but the vector kernels above are nowhere near that: every FMA takes its second operand from a BROADCAST
// ds_read_b128, which still moves 64 x 16 bytes through the LDS crossbar (8 cycles per instruction, one per 4 FMAs per
// wave: the LDS saturates at ~1/4 of the vector rate; 334 us for the text self-attention forward of the dense step =
// 17 TFLOP/s).  In the matrix form a wave owns 16 rows of its own side as the B operand IN REGISTERS (reduction index
// permuted so that four k-steps are one 16-byte load: k-step 4 J + jj of lane group g = lane / 16 is dimension
// 16 J + 4 g + jj), the other side's rows are staged in LDS once per workgroup and read as the A operand (one
// ds_read_b128 per 4 MFMAs, every lane its own bytes), and a product's result -- lane holds [other-side row 4 g + i]
// [own row lane % 16] -- IS the B operand of the next product over the other side's rows (k-step i <-> row 4 g + i),
// whose A operand (the other side transposed: m = dimension, k = row 4 g + i) is a 4-byte LDS read per MFMA.
// Exact fp32 operands, fp32 accumulation; mask, -10000, dropout counters and variable lengths as above.
// ------------------------------------------------------------------------------------------------
constexpr int XM_T = 256;     // 4 waves x 16 own rows
constexpr int XM_R = 64;      // own rows per workgroup
#define XM_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
constexpr float XM_LOG2E = 1.4426950408889634f, XM_NEG2 = -10000.0f * 1.4426950408889634f;
#ifndef UNIMM_X3M_MIN_WAVES
#define UNIMM_X3M_MIN_WAVES 2
#endif

// Workgroup -> (row block, head, sequence).  Workgroups go to the 8 XCDs round-robin by their linear id, and every XCD has
// its own L2: with the plain (row block, head, sequence) grid the row blocks of one (sequence, head) -- which stage the SAME
// other-side rows -- land on different XCDs and each fetches them from HBM.  Here the 1-D id d is decoded as xcd = d % 8,
// j = d / 8, row block = j % nblk, pair = (j / nblk) * 8 + xcd: all row blocks of a pair run on one XCD, back to back.
__device__ __forceinline__ bool xm_decode(int nblk, int H, int B, const int* order, int& blk, int& head, int& b) {
  const int d = blockIdx.x, xcd = d & 7, j = d >> 3;
  blk = j % nblk;
  const int pair = (j / nblk) * 8 + xcd;
  if (pair >= H * B) return false;
  const int seq = pair / H;
  head = pair - seq * H;
  b = order != nullptr ? order[seq] : seq;       // longest sequences first: the launch's tail is its shortest items
  return true;
}
__host__ inline unsigned xm_grid(int nblk, int H, int B) { return (unsigned)(((H * B + 7) / 8) * nblk * 8); }

__device__ __forceinline__ f32x4 xm_mfma(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// own row `row` (clamped by the caller), D dimensions from column col0, as the B operand: reg[4 J + jj] = x[16 J + 4 g + jj]
template <int D>
__device__ __forceinline__ void xm_load_b(const float* __restrict__ g, int grp, float* reg) {
#pragma unroll
  for (int J = 0; J < D / 16; ++J) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(g + 16 * J + 4 * grp);
    reg[4 * J] = t[0]; reg[4 * J + 1] = t[1]; reg[4 * J + 2] = t[2]; reg[4 * J + 3] = t[3];
  }
}
// Staging: rows [r0, r0 + NR) of the other side (zeros past `len`) into LDS rows of D + 4 floats, in two halves, so that a
// chunk's global loads are in flight while the previous chunk is being multiplied:
// load() requests rows [r0, r0 + NR) into registers, store() puts them into LDS (between two barriers).
template <int D, int NR> struct XmChunk {
  static constexpr int N = NR * (D / 4) / XM_T;
  static_assert(NR * (D / 4) % XM_T == 0, "chunk must split evenly over the workgroup");
  f32x4 t[N];
  __device__ __forceinline__ void load(const float* __restrict__ g, int row_base, int r0, int len, int ld, int col0) {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int i = n * XM_T + threadIdx.x, r = i / (D / 4), c = (i - r * (D / 4)) * 4;
      t[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (r0 + r < len) t[n] = *reinterpret_cast<const f32x4*>(g + (size_t)(row_base + r0 + r) * ld + col0 + c);
    }
  }
  __device__ __forceinline__ void store(float* __restrict__ lds) const {
#pragma unroll
    for (int n = 0; n < N; ++n) {
      const int i = n * XM_T + threadIdx.x, r = i / (D / 4), c = (i - r * (D / 4)) * 4;
      *reinterpret_cast<f32x4*>(lds + r * (D + 4) + c) = t[n];
    }
  }
};
// [16 staged rows of `tile`] x [the wave's 16 own rows]: the lane gets [staged row 4 g + i][own row lane % 16]
template <int D>
__device__ __forceinline__ f32x4 xm_rows_x_own(const float* __restrict__ tile, int r, int grp, const float* reg) {
  constexpr int LD = D + 4;
  f32x4 a[D / 16];
#pragma unroll
  for (int J = 0; J < D / 16; ++J) a[J] = *reinterpret_cast<const f32x4*>(tile + r * LD + 16 * J + 4 * grp);
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int J = 0; J < D / 16; ++J)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) acc = xm_mfma(a[J][jj], reg[4 * J + jj], acc);
  return acc;
}
// Two products at once -- [16 staged rows of tA] x [own rows as regA] and [16 staged rows of tB] x [own rows as regB]: the
// lane gets [staged row 4 g + i][own row lane % 16] of each.  All fragments are requested before the first MFMA and the
// two accumulation chains alternate (left to the compiler, every ds_read_b128 was followed by a full lgkmcnt(0) wait and
// four dependent MFMAs: the LDS round trip was exposed once per 128 matrix cycles).
template <int D>
__device__ __forceinline__ void xm_rows_x_own2(const float* __restrict__ tA, const float* __restrict__ tB, int r, int grp,
                                               const float* regA, const float* regB, f32x4& oA, f32x4& oB) {
  constexpr int LD = D + 4;
  f32x4 a[D / 16], b[D / 16];
#pragma unroll
  for (int J = 0; J < D / 16; ++J) {
    a[J] = *reinterpret_cast<const f32x4*>(tA + r * LD + 16 * J + 4 * grp);
    b[J] = *reinterpret_cast<const f32x4*>(tB + r * LD + 16 * J + 4 * grp);
  }
  f32x4 ca = {0.f, 0.f, 0.f, 0.f}, cb = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int J = 0; J < D / 16; ++J)
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      ca = xm_mfma(a[J][jj], regA[4 * J + jj], ca);
      cb = xm_mfma(b[J][jj], regB[4 * J + jj], cb);
    }
  oA = ca; oB = cb;
}
// acc^T[dimension][own row] += staged^T[dimension][staged row 4 g + i] . w[staged row 4 g + i][own row]   (D / 16 independent chains)
template <int D>
__device__ __forceinline__ void xm_accum_t(f32x4 (&acc)[D / 16], const float* __restrict__ tile, int r, int grp, const f32x4 w) {
  constexpr int LD = D + 4;
  float a[D / 16][4];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt)
#pragma unroll
    for (int i = 0; i < 4; ++i) a[dt][i] = tile[(4 * grp + i) * LD + 16 * dt + r];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int dt = 0; dt < D / 16; ++dt) acc[dt] = xm_mfma(a[dt][i], w[i], acc[dt]);
}
// Result tiles leave the workgroup row-contiguously.  The accumulators are transposed -- a lane holds dimensions 16 dt + 4 g + i
// of ONE own row -- so storing them directly is a 16-byte (fp32) or 8-byte (split planes) piece per lane on 16 different rows
// per instruction; with the split planes written that way the forward kernel of the 37-region sides ran 40 % longer (58.8 ->
// 82.7 us).  Instead every wave puts its [16 rows][D] tile into its quarter of the (by then dead) staging buffer and walks it
// with consecutive lanes on consecutive 16 bytes: fp32 rows (out32 or NULL) four floats per lane, the x-type split planes
// [hi | lo | hi] (out3 or NULL) eight values per lane.  nrows = how many of the wave's 16 rows exist.
template <int D>
__device__ __forceinline__ void xm_store_rows(float* __restrict__ tile, const f32x4 (&acc)[D / 16], float s, int r, int grp, int lane,
                                              int nrows, float* __restrict__ out32, int ld32, bf16_t* __restrict__ out3, int ld3, int cp) {
  constexpr int LD = D + 4;
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt)
    *reinterpret_cast<f32x4*>(tile + r * LD + 16 * dt + 4 * grp) = f32x4{acc[dt][0] * s, acc[dt][1] * s, acc[dt][2] * s, acc[dt][3] * s};
  // (wave-private rows: the wave's own LDS writes are ordered before its reads by the waitcnt the compiler places)
  if (out32 != nullptr) {
#pragma unroll
    for (int it = 0; it < D / 16; ++it) {
      const int idx = it * 64 + lane, row = idx / (D / 4), c = (idx % (D / 4)) * 4;
      const f32x4 t = *reinterpret_cast<const f32x4*>(tile + row * LD + c);
      if (row < nrows) *reinterpret_cast<f32x4*>(out32 + (size_t)row * ld32 + c) = t;
    }
  }
  if (out3 != nullptr) {
#pragma unroll
    for (int it = 0; it < D / 32; ++it) {
      const int idx = it * 64 + lane, row = idx / (D / 8), c = (idx % (D / 8)) * 8;
      const f32x4 t0 = *reinterpret_cast<const f32x4*>(tile + row * LD + c), t1 = *reinterpret_cast<const f32x4*>(tile + row * LD + c + 4);
      float hi[8], lo[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { split2(t0[e], hi[e], lo[e]); split2(t1[e], hi[4 + e], lo[4 + e]); }
      const u32x4 Hh = u32x4{pack2bf(hi[0], hi[1]), pack2bf(hi[2], hi[3]), pack2bf(hi[4], hi[5]), pack2bf(hi[6], hi[7])};
      const u32x4 Ll = u32x4{pack2bf(lo[0], lo[1]), pack2bf(lo[2], lo[3]), pack2bf(lo[4], lo[5]), pack2bf(lo[6], lo[7])};
      if (row < nrows) {
        bf16_t* o = out3 + (size_t)row * ld3 + c;
        *reinterpret_cast<u32x4*>(o) = Hh;
        *reinterpret_cast<u32x4*>(o + cp) = Ll;
        *reinterpret_cast<u32x4*>(o + 2 * cp) = Hh;
      }
    }
  }
}
__device__ __forceinline__ float xm_groups_max(float v) { v = fmaxf(v, __shfl_xor(v, 16, 64)); return fmaxf(v, __shfl_xor(v, 32, 64)); }
__device__ __forceinline__ float xm_groups_sum(float v) { v += __shfl_xor(v, 16, 64); return v + __shfl_xor(v, 32, 64); }

// Forward.  NT = 16-key tiles a query row can have (4 / 8 / 16): the whole score row of a query lives in registers
// (exact two-pass softmax), K streams through LDS 64 keys at a time, then V.
template <int D, int NT>
__global__ __launch_bounds__(XM_T, UNIMM_X3M_MIN_WAVES) void x3m_attn_fwd_kernel(AttnF32 p) {
  constexpr bool OWN_Q = true;
  constexpr int LD = D + 4, KC = 64;
  __shared__ __attribute__((aligned(16))) float Xs[KC * LD];
  drop_resolve(p.drop);
  int blk, head, b;
  if (!xm_decode(OWN_Q ? (p.Tq + XM_R - 1) / XM_R : (p.Tk + XM_R - 1) / XM_R, p.H, p.B, p.order, blk, head, b)) return;
  const int qlen = p.q_len != nullptr ? p.q_len[b] : p.Tq, qoff = p.q_off != nullptr ? p.q_off[b] : b * p.Tq;
  const int klen = p.k_len != nullptr ? p.k_len[b] : p.Tk, koff = p.k_off != nullptr ? p.k_off[b] : b * p.Tk;
  const int r0 = blk * XM_R;
  if (r0 >= qlen) return;                                  // workgroup-uniform
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, grp = lane >> 4;
  const bool live = r0 + wave * 16 < qlen;                 // wave-uniform: some of the wave's queries exist
  const bool valid = r0 + wave * 16 + r < qlen;
  const int qr = valid ? r0 + wave * 16 + r : qlen - 1;
  float qreg[D / 4];
  xm_load_b<D>(p.q + (size_t)(qoff + qr) * p.ldq + head * D, grp, qreg);
  // the query's mask words, requested with Q (left inside the softmax loops each was a global load with its use right
  // behind it: one exposed round trip per word)
  const uint32_t* mrow = p.mask + (size_t)b * p.mbs + (size_t)qr * p.mqs;
  const int nmw = (p.Tk + 31) >> 5;
  uint32_t mws[NT / 2];
#pragma unroll
  for (int w = 0; w < NT / 2; ++w) mws[w] = mrow[w < nmw ? w : nmw - 1];
  f32x4 s[NT];
  XmChunk<D, KC> pre;
  pre.load(p.k, koff, 0, klen, p.ldk, head * D);
#pragma unroll
  for (int ch = 0; ch < NT / 4; ++ch) {
    if (ch * KC < klen) {
      __syncthreads();
      pre.store(Xs);
      __syncthreads();
      if ((ch + 1) * KC < klen) pre.load(p.k, koff, (ch + 1) * KC, klen, p.ldk, head * D);   // next K chunk, or V's first
      else pre.load(p.v, koff, 0, klen, p.ldv, head * D);
      if (live) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          if (ch * KC + tt * 16 < klen) s[4 * ch + tt] = xm_rows_x_own<D>(Xs + tt * 16 * LD, r, grp, qreg);
      }
    }
  }
  // mask + softmax: the lane holds keys 16 t + 4 g + i of query qr; the row's other keys are in lanes ^16, ^32, ^48
  const uint32_t hrow = ((uint32_t)b * p.H + head) * p.Tq + (uint32_t)qr;
  const float sc2 = p.scale * XM_LOG2E;
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (16 * t < klen) {
      const uint32_t mw = mws[t >> 1] >> (16 * (t & 1) + 4 * grp);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float sv = (16 * t + 4 * grp + i < klen) ? fmaf(s[t][i], sc2, ((mw >> i) & 1u) ? 0.f : XM_NEG2) : -INFINITY;   // log2 domain
        s[t][i] = sv;
        m = fmaxf(m, sv);
      }
    }
  }
  m = xm_groups_max(m);
  float l = 0.f;
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    if (16 * t < klen) {
      const uint32_t kb = p.drop.thr != 0u ? drop_bits4(p.drop, hrow, (uint32_t)p.Tk, (uint32_t)(16 * t + 4 * grp)) : 0xfu;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float e = __builtin_amdgcn_exp2f(s[t][i] - m);
        l += e;
        s[t][i] = ((kb >> i) & 1u) ? e : 0.f;
      }
    }
  }
  l = xm_groups_sum(l);
  f32x4 acc[D / 16];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) acc[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int ch = 0; ch < NT / 4; ++ch) {
    if (ch * KC < klen) {
      __syncthreads();
      pre.store(Xs);
      __syncthreads();
      if ((ch + 1) * KC < klen) pre.load(p.v, koff, (ch + 1) * KC, klen, p.ldv, head * D);
      if (live) {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
          if (ch * KC + tt * 16 < klen) xm_accum_t<D>(acc, Xs + tt * 16 * LD, r, grp, s[4 * ch + tt]);
      }
    }
  }
  __syncthreads();                                         // every wave is done with the last V chunk: the buffer becomes the store tiles
  if (!live) return;
  // acc is out^T: lane holds dimensions 16 dt + 4 g + i of query qr
  // (a sequence with no keys at all -- k_len[b] == 0 -- has l = 0: its rows are written as zeros, not 0 * inf)
  const float inv = l > 0.f ? (p.drop.thr != 0u ? p.drop.scale : 1.0f) / l : 0.f;
  const int row0 = r0 + wave * 16, nrows = qlen - row0 < 16 ? qlen - row0 : 16;
  xm_store_rows<D>(Xs + wave * 16 * LD, acc, inv, r, grp, lane, nrows, p.out + (size_t)(qoff + row0) * p.ldo + head * D, p.ldo,
                   p.o3 != nullptr ? p.o3 + (size_t)(qoff + row0) * p.ld3 + head * D : nullptr, p.ld3, p.cp3);
  if (valid && p.lse != nullptr && grp == 0) p.lse[((size_t)b * p.H + head) * p.Tq + qr] = m * 0.6931471805599453f + __logf(l);
}

// dQ (+ delta = rowsum(dO o O)): the wave's 16 queries in registers (Q and dO as B operands), K and V staged
template <int D>
__global__ __launch_bounds__(XM_T, UNIMM_X3M_MIN_WAVES) void x3m_attn_bwd_dq_kernel(AttnF32 p) {
  constexpr bool OWN_Q = true;
  constexpr int LD = D + 4, KC = D == 64 ? 64 : 32;
  __shared__ __attribute__((aligned(16))) float KV[2 * KC * LD];   // K chunk | V chunk; at the end the four waves' store tiles
  float* const Ks = KV;
  float* const Vs = KV + KC * LD;
  static_assert(2 * KC >= XM_R, "store tiles: 16 rows per wave must fit the staging buffer");
  __shared__ uint32_t Mq[XM_R * 9];                        // mask words of the 64 queries (rows of 8 + 1 pad)
  drop_resolve(p.drop);
  int blk, head, b;
  if (!xm_decode(OWN_Q ? (p.Tq + XM_R - 1) / XM_R : (p.Tk + XM_R - 1) / XM_R, p.H, p.B, p.order, blk, head, b)) return;
  const int qlen = p.q_len != nullptr ? p.q_len[b] : p.Tq, qoff = p.q_off != nullptr ? p.q_off[b] : b * p.Tq;
  const int klen = p.k_len != nullptr ? p.k_len[b] : p.Tk, koff = p.k_off != nullptr ? p.k_off[b] : b * p.Tk;
  const int r0 = blk * XM_R;
  if (r0 >= qlen) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, grp = lane >> 4;
  const bool live = r0 + wave * 16 < qlen;
  const bool valid = r0 + wave * 16 + r < qlen;
  const int qr = valid ? r0 + wave * 16 + r : qlen - 1;
  XmChunk<D, KC> prek, prev;                               // first K / V chunk: requested before anything that waits
  prek.load(p.k, koff, 0, klen, p.ldk, head * D);
  prev.load(p.v, koff, 0, klen, p.ldv, head * D);
  {
    const int nmw = (p.Tk + 31) >> 5;
    for (int i = threadIdx.x; i < XM_R * 8; i += XM_T) {
      const int ql = i >> 3, w = i & 7, qq = r0 + ql < qlen ? r0 + ql : qlen - 1;
      Mq[ql * 9 + w] = w < nmw ? p.mask[(size_t)b * p.mbs + (size_t)qq * p.mqs + w] : 0u;   // (visible after the loop's first barrier)
    }
  }
  float qreg[D / 4], doreg[D / 4];
  xm_load_b<D>(p.q + (size_t)(qoff + qr) * p.ldq + head * D, grp, qreg);
  xm_load_b<D>(p.dout + (size_t)(qoff + qr) * p.lddo + head * D, grp, doreg);
  float delta = 0.f;
  {
    float oreg[D / 4];
    xm_load_b<D>(p.o + (size_t)(qoff + qr) * p.ldo + head * D, grp, oreg);
#pragma unroll
    for (int c = 0; c < D / 4; ++c) delta = fmaf(doreg[c], oreg[c], delta);
    delta = xm_groups_sum(delta);
  }
  const size_t li = ((size_t)b * p.H + head) * p.Tq + qr;
  const float lse2 = p.lse[li] * XM_LOG2E, sc2 = p.scale * XM_LOG2E;         // exp(x) = exp2(x log2 e): one v_exp_f32 per element
  if (valid && grp == 0) p.delta[li] = delta;
  f32x4 dq[D / 16];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) dq[dt] = f32x4{0.f, 0.f, 0.f, 0.f};
  const uint32_t* mrow = Mq + (wave * 16 + r) * 9;
  const uint32_t hrow = ((uint32_t)b * p.H + head) * p.Tq + (uint32_t)qr;
  const float dsc = p.drop.thr != 0u ? p.drop.scale : 1.0f;
  const uint32_t nodrop = p.drop.thr != 0u ? 0u : 0xfu;
  for (int k0 = 0; k0 < klen; k0 += KC) {
    __syncthreads();
    prek.store(Ks);
    prev.store(Vs);
    __syncthreads();
    if (k0 + KC < klen) {
      prek.load(p.k, koff, k0 + KC, klen, p.ldk, head * D);
      prev.load(p.v, koff, k0 + KC, klen, p.ldv, head * D);
    }
    if (!live) continue;
    // Tiles of 16 keys, software-pipelined: the two products of tile t + 1 (32 MFMAs, nothing but LDS reads in front of them) are
    // issued BEFORE the elementwise work of tile t, whose ~60 vector instructions then run in the shadow of those MFMAs (a
    // v_mfma_f32_16x16x4_f32 holds the issue port for 8 of its 32 cycles); dS of tile t feeds the third product.
    const int nt = (klen - k0 + 15) >> 4 < KC / 16 ? (klen - k0 + 15) >> 4 : KC / 16;
    auto elementwise = [&](const f32x4& st, const f32x4& dpt, int kt) {
      const uint32_t mw = mrow[kt >> 5] >> ((kt & 16) + 4 * grp);
      const uint32_t kb = drop_bits4(p.drop, hrow, (uint32_t)p.Tk, (uint32_t)(kt + 4 * grp)) | nodrop;   // (no branch: one basic block)
      f32x4 ds;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        // keys past klen: their staged K rows are zero, so whatever dS they get adds nothing to dQ
        const float pr = __builtin_amdgcn_exp2f(fmaf(st[i], sc2, (((mw >> i) & 1u) ? 0.f : XM_NEG2) - lse2));
        ds[i] = (pr * p.scale) * fmaf(dpt[i], ((kb >> i) & 1u) ? dsc : 0.f, -delta);
      }
      return ds;
    };
    f32x4 st, dpt;
    xm_rows_x_own2<D>(Ks, Vs, r, grp, qreg, doreg, st, dpt);
    for (int tt = 0; tt + 1 < nt; ++tt) {
      f32x4 stn, dptn;
      xm_rows_x_own2<D>(Ks + (tt + 1) * 16 * LD, Vs + (tt + 1) * 16 * LD, r, grp, qreg, doreg, stn, dptn);
      const f32x4 ds = elementwise(st, dpt, k0 + tt * 16);
      xm_accum_t<D>(dq, Ks + tt * 16 * LD, r, grp, ds);
      st = stn; dpt = dptn;
      // issue order asked of the scheduler: the products' fragment reads, then one MFMA with three vector instructions behind
      // it until the elementwise work is spent, the transposed fragment reads, the third product
      XM_SGB(0x100, D / 8 + 1);
#pragma unroll
      for (int i = 0; i < D / 2; ++i) { XM_SGB(0x008, 1); XM_SGB(0x002, D == 64 ? 3 : 2); }
      XM_SGB(0x100, D / 8);
#pragma unroll
      for (int i = 0; i < D / 4; ++i) { XM_SGB(0x008, 1); XM_SGB(0x002, 1); }
    }
    {
      const f32x4 ds = elementwise(st, dpt, k0 + (nt - 1) * 16);
      xm_accum_t<D>(dq, Ks + (nt - 1) * 16 * LD, r, grp, ds);
    }
  }
  __syncthreads();                                         // the staging buffer becomes the store tiles
  if (!live) return;
  const int row0 = r0 + wave * 16, nrows = qlen - row0 < 16 ? qlen - row0 : 16;
  xm_store_rows<D>(KV + wave * 16 * LD, dq, 1.0f, r, grp, lane, nrows,
                   p.dq3 == nullptr ? p.dq + (size_t)(qoff + row0) * p.lddq + head * D : nullptr, p.lddq,
                   p.dq3 != nullptr ? p.dq3 + (size_t)(qoff + row0) * p.ld3 + head * D : nullptr, p.ld3, p.cp3);
}

// dK, dV: the wave's 16 keys in registers (K and V as B operands), Q and dO rows staged with their lse / delta / mask words
template <int D>
__global__ __launch_bounds__(XM_T, UNIMM_X3M_MIN_WAVES) void x3m_attn_bwd_dkv_kernel(AttnF32 p) {
  constexpr bool OWN_Q = false;
  constexpr int LD = D + 4, QC = D == 64 ? 64 : 32;
  __shared__ __attribute__((aligned(16))) float QO[2 * QC * LD];   // Q chunk | dO chunk; at the end the four waves' store tiles
  float* const Qs = QO;
  float* const Os = QO + QC * LD;
  static_assert(2 * QC >= XM_R, "store tiles: 16 rows per wave must fit the staging buffer");
  __shared__ float Ls[QC], Ds[QC];
  __shared__ uint32_t Ms[QC * 2];                          // per staged query: the two mask words this workgroup's 64 keys lie in
  drop_resolve(p.drop);
  int blk, head, b;
  if (!xm_decode(OWN_Q ? (p.Tq + XM_R - 1) / XM_R : (p.Tk + XM_R - 1) / XM_R, p.H, p.B, p.order, blk, head, b)) return;
  const int qlen = p.q_len != nullptr ? p.q_len[b] : p.Tq, qoff = p.q_off != nullptr ? p.q_off[b] : b * p.Tq;
  const int klen = p.k_len != nullptr ? p.k_len[b] : p.Tk, koff = p.k_off != nullptr ? p.k_off[b] : b * p.Tk;
  const int r0 = blk * XM_R;
  if (r0 >= klen) return;
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r = lane & 15, grp = lane >> 4;
  const bool live = r0 + wave * 16 < klen;
  const bool valid = r0 + wave * 16 + r < klen;
  const int key = valid ? r0 + wave * 16 + r : klen - 1;
  float kreg[D / 4], vreg[D / 4];
  xm_load_b<D>(p.k + (size_t)(koff + key) * p.ldk + head * D, grp, kreg);
  xm_load_b<D>(p.v + (size_t)(koff + key) * p.ldv + head * D, grp, vreg);
  f32x4 dk[D / 16], dv[D / 16];
#pragma unroll
  for (int dt = 0; dt < D / 16; ++dt) { dk[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; dv[dt] = f32x4{0.f, 0.f, 0.f, 0.f}; }
  const uint32_t hbase = ((uint32_t)b * p.H + head) * p.Tq;
  const float dsc = p.drop.thr != 0u ? p.drop.scale : 1.0f, sc2 = p.scale * XM_LOG2E;
  const size_t lbase = ((size_t)b * p.H + head) * p.Tq;
  // every key of this workgroup lies in mask word r0 / 32 or the next one (64 keys from a multiple of 64): the word
  // of the wave's 16 keys is wave-uniform
  const int mword = (wave * 16) >> 5;                       // (r0 is a multiple of 64: word r0 / 32 or the next one)
  XmChunk<D, QC> preq, preo;
  float prel = 0.f, pred = 0.f;
  uint32_t prem = 0u;
  const int nmw = (p.Tk + 31) >> 5;
  auto request = [&](int q0) {
    preq.load(p.q, qoff, q0, qlen, p.ldq, head * D);
    preo.load(p.dout, qoff, q0, qlen, p.lddo, head * D);
    if (threadIdx.x < QC) {
      const bool ok = q0 + (int)threadIdx.x < qlen;
      prel = ok ? p.lse[lbase + q0 + threadIdx.x] * XM_LOG2E : 0.f;        // log2 domain
      pred = ok ? p.delta[lbase + q0 + threadIdx.x] : 0.f;
    } else if (threadIdx.x < 3 * QC) {
      const int ql = (threadIdx.x - QC) >> 1, w = (r0 >> 5) + (threadIdx.x & 1);
      prem = (q0 + ql < qlen && w < nmw) ? p.mask[(size_t)b * p.mbs + (size_t)(q0 + ql) * p.mqs + w] : 0u;
    }
  };
  request(0);
  for (int q0 = 0; q0 < qlen; q0 += QC) {
    __syncthreads();
    preq.store(Qs);
    preo.store(Os);
    if (threadIdx.x < QC) { Ls[threadIdx.x] = prel; Ds[threadIdx.x] = pred; }
    else if (threadIdx.x < 3 * QC) Ms[threadIdx.x - QC] = prem;
    __syncthreads();
    if (q0 + QC < qlen) request(q0 + QC);
    if (!live) continue;
#pragma unroll
    for (int tt = 0; tt < QC / 16; ++tt) {
      const int qt = q0 + tt * 16;
      if (qt >= qlen) break;
      f32x4 sq, dpq;                                                          // [query 4 g + i][key r]
      xm_rows_x_own2<D>(Qs + tt * 16 * LD, Os + tt * 16 * LD, r, grp, kreg, vreg, sq, dpq);
      f32x4 pd, ds;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ql = tt * 16 + 4 * grp + i, qq = q0 + ql;
        const bool qok = qq < qlen;
        const uint32_t mw = Ms[ql * 2 + mword];
        const float pr = (qok && valid) ? __builtin_amdgcn_exp2f(fmaf(sq[i], sc2, (((mw >> (key & 31)) & 1u) ? 0.f : XM_NEG2) - Ls[ql])) : 0.f;
        const bool keep = xa_keep(p.drop, hbase + (uint32_t)qq, (uint32_t)p.Tk, (uint32_t)key);
        pd[i] = keep ? pr * dsc : 0.f;
        const float dP = keep ? dpq[i] * dsc : 0.f;
        ds[i] = pr * (dP - Ds[ql]) * p.scale;
      }
      xm_accum_t<D>(dv, Os + tt * 16 * LD, r, grp, pd);
      xm_accum_t<D>(dk, Qs + tt * 16 * LD, r, grp, ds);
    }
  }
  __syncthreads();                                         // the staging buffer becomes the store tiles
  if (!live) return;
  const int row0 = r0 + wave * 16, nrows = klen - row0 < 16 ? klen - row0 : 16;
  float* tile = QO + wave * 16 * LD;
  const bool pl = p.dk3 != nullptr;
  xm_store_rows<D>(tile, dk, 1.0f, r, grp, lane, nrows, pl ? nullptr : p.dk + (size_t)(koff + row0) * p.lddk + head * D, p.lddk,
                   pl ? p.dk3 + (size_t)(koff + row0) * p.ld3 + head * D : nullptr, p.ld3, p.cp3);
  xm_store_rows<D>(tile, dv, 1.0f, r, grp, lane, nrows, pl ? nullptr : p.dv + (size_t)(koff + row0) * p.lddv + head * D, p.lddv,
                   pl ? p.dv3 + (size_t)(koff + row0) * p.ld3 + head * D : nullptr, p.ld3, p.cp3);
}

int g_x3_attn_impl = 1;       // 1 = matrix-instruction kernels, 0 = the vector kernels (A/B: unimm_x3_attn_set_impl)

int fill_attn(const unimm_attn_args* a, AttnF32& p) {
  if (a == nullptr || !a->q || !a->k || !a->v || !a->out || !a->mask) return UNIMM_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->Tq <= 0 || a->Tk <= 0 || a->Tq > 256 || a->Tk > 256) return UNIMM_E_SHAPE;
  if (a->D != 64 && a->D != 128) return UNIMM_E_SHAPE;
  if ((a->ldq % 4) || (a->ldk % 4) || (a->ldv % 4) || (a->ldo % 4)) return UNIMM_E_ALIGN;
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) & 15) return UNIMM_E_ALIGN;
  if ((a->q_off == nullptr) != (a->q_len == nullptr) || (a->k_off == nullptr) != (a->k_len == nullptr)) return UNIMM_E_ARG;
  if (a->ks_off != nullptr || a->ks_len != nullptr) return UNIMM_E_ARG;      // the spliced shared key segment is a bf16-kernel feature
  p = AttnF32{};
  p.q = (const float*)a->q; p.k = (const float*)a->k; p.v = (const float*)a->v; p.out = (float*)a->out; p.lse = a->lse;
  p.mask = a->mask; p.q_off = a->q_off; p.q_len = a->q_len; p.k_off = a->k_off; p.k_len = a->k_len;
  p.order = a->order;
  p.B = a->B; p.H = a->H; p.Tq = a->Tq; p.Tk = a->Tk;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
  p.mqs = a->mask_q_stride; p.mbs = a->mask_b_stride; p.scale = a->scale;
  p.drop.key = a->drop_key; p.drop.thr = a->drop_thr; p.drop.scale = a->drop_scale; p.drop.salt = a->drop_salt; p.drop.key2 = 0u;
  return UNIMM_OK;
}

}  // namespace

extern "C" int unimm_x3_split(const unimm_x3_split_args* a, void* stream) {
  if (a == nullptr || a->a == nullptr || (a->out32 == nullptr && a->out3 == nullptr)) return UNIMM_E_ARG;
  if (a->rows <= 0 || a->cols <= 0 || a->cp < a->cols || (a->cp % 64) != 0 || a->lda < a->cols) return UNIMM_E_SHAPE;
  if (a->op < 0 || a->op > UNIMM_X3_MUL_DGELU) return UNIMM_E_ARG;
  if ((a->op == UNIMM_X3_ADD || a->op == UNIMM_X3_MUL_DGELU) && (a->b == nullptr || a->ldb < a->cols)) return UNIMM_E_ARG;
  if (a->out32 != nullptr && a->ld32 < a->cols) return UNIMM_E_SHAPE;
  if (a->out3 != nullptr && (((uintptr_t)a->out3) & 15)) return UNIMM_E_ALIGN;
  SplitParams p;
  p.a = a->a; p.b = (a->op == UNIMM_X3_ADD || a->op == UNIMM_X3_MUL_DGELU) ? a->b : nullptr;
  p.out32 = a->out32; p.out3 = (bf16_t*)a->out3;
  p.rows = a->rows; p.cols = a->cols; p.cp = a->cp; p.lda = a->lda; p.ldb = a->ldb; p.ld32 = a->ld32; p.op = a->op;
  p.wtype = a->wtype != 0;
  const long total = p.rows * (p.cp >> 3);
  const unsigned blocks = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
  hipLaunchKernelGGL(x3_split_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_split_wt(const float* src, void* dst, int32_t R, int32_t C, int32_t lds, int32_t Rp, void* stream) {
  if (src == nullptr || dst == nullptr) return UNIMM_E_ARG;
  if (R <= 0 || C <= 0 || lds < C || Rp < R || (Rp % 64) != 0) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(x3_split_wt_kernel, dim3((C + 31) / 32, (R + 31) / 32), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, R, C, lds, Rp);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_layernorm_bwd_partials(const float* dy, const float* x, const float* mean, const float* rstd,
                                               const float* gamma, float* dx32, void* dxd3, float* partials, int32_t M,
                                               int32_t H, uint32_t drop_key, uint32_t drop_thr, float drop_scale,
                                               uint32_t odrop_key, uint32_t odrop_thr, float odrop_scale, int32_t* blocks_out,
                                               const int32_t* m_dev, const uint32_t* drop_salt, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !partials || !blocks_out || (!dx32 && !dxd3)) return UNIMM_E_ARG;
  if (M <= 0 || H <= 0 || H > 1024 || (H % 64) != 0) return UNIMM_E_SHAPE;
  DropoutArg d{drop_key, drop_thr, drop_scale, drop_salt, 0u}, od{odrop_key, odrop_thr, odrop_scale, drop_salt, 0u};
  int blocks = (M + 3) / 4;
  blocks = blocks < RED_BLOCKS ? blocks : RED_BLOCKS;
  *blocks_out = blocks;
  hipLaunchKernelGGL(x3_layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, x, mean, rstd, gamma, dx32,
                     (bf16_t*)dxd3, partials, M, H, d, od, m_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y32, void* y3, float* mean,
                                      float* rstd, int32_t M, int32_t H, float eps, uint32_t drop_key, uint32_t drop_thr,
                                      float drop_scale, const uint32_t* drop_salt, void* stream) {
  if (!x || !gamma || !beta || !y3 || ((mean == nullptr) != (rstd == nullptr))) return UNIMM_E_ARG;
  if (M <= 0 || H <= 0 || H > MAXC * 512 || (H % 64) != 0) return UNIMM_E_SHAPE;
  if (((uintptr_t)y3) & 15) return UNIMM_E_ALIGN;
  int blocks = (M + 3) / 4;
  blocks = blocks > 2048 ? 2048 : blocks;
  DropoutArg d{drop_key, drop_thr, drop_scale, drop_salt, 0u};
  hipLaunchKernelGGL(x3_layernorm_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y32, (bf16_t*)y3,
                     mean, rstd, M, H, eps, d);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_lm_loss_bwd(const float* logits, const int32_t* labels, const int32_t* weights, const float* lse,
                                    const float* g, float inv_denom, void* out3, int32_t n, int32_t V, int32_t ld, int32_t cp,
                                    const int32_t* n_dev, const float* inv_dev, void* stream) {
  if (!logits || !labels || !weights || !lse || !g || !out3) return UNIMM_E_ARG;
  if (n <= 0 || V <= 0 || ld < V || cp < V || (cp % 64)) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(x3_lm_loss_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, labels, weights, lse, g,
                     inv_denom, (bf16_t*)out3, V, ld, cp, 1e-6f, n_dev, inv_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_kl_loss_bwd(const float* pred, const float* target, const int32_t* label, const float* lse,
                                    const float* g, float inv_denom, void* out3, int32_t rows, int32_t C, int32_t ld, int32_t cp,
                                    const float* inv_dev, void* stream) {
  if (!pred || !target || !label || !lse || !g || !out3) return UNIMM_E_ARG;
  if (rows <= 0 || C <= 0 || ld < C || cp < C || (cp % 64)) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(x3_kl_loss_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, target, label, lse, g,
                     inv_denom, (bf16_t*)out3, C, ld, cp, inv_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_rows_add(float* dst, const int32_t* idx, const float* src, int32_t n, int32_t H, int32_t ldd, void* stream) {
  if (!dst || !idx || !src || n <= 0 || H <= 0 || ldd < H) return UNIMM_E_ARG;
  const long total = (long)n * H;
  hipLaunchKernelGGL(x3_rows_add_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dst, idx, src,
                     n, H, ldd);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_attn_fwd(const unimm_attn_args* a, const unimm_x3_attn_planes* pl, void* stream) {
  AttnF32 p;
  const int rc = fill_attn(a, p);
  if (rc != UNIMM_OK) return rc;
  if (pl != nullptr && pl->out3 != nullptr) {
    if (g_x3_attn_impl == 0) return UNIMM_E_ARG;                       // split outputs: the matrix kernels only
    if ((((uintptr_t)pl->out3) & 15) || (pl->ld3 % 8) || (pl->cp3 % 8) || pl->cp3 < a->H * a->D || pl->ld3 < 3 * pl->cp3) return UNIMM_E_ALIGN;
    p.o3 = (bf16_t*)pl->out3; p.ld3 = pl->ld3; p.cp3 = pl->cp3;
  }
  hipStream_t s = (hipStream_t)stream;
  if (g_x3_attn_impl != 0) {
    const dim3 grid(xm_grid((a->Tq + XM_R - 1) / XM_R, a->H, a->B));
    const int nt = (a->Tk + 15) / 16;
#define UNIMM_X3M_FWD(D_, NT_) hipLaunchKernelGGL((x3m_attn_fwd_kernel<D_, NT_>), grid, dim3(XM_T), 0, s, p)
    if (a->D == 64) { if (nt <= 4) UNIMM_X3M_FWD(64, 4); else if (nt <= 8) UNIMM_X3M_FWD(64, 8); else UNIMM_X3M_FWD(64, 16); }
    else { if (nt <= 4) UNIMM_X3M_FWD(128, 4); else if (nt <= 8) UNIMM_X3M_FWD(128, 8); else UNIMM_X3M_FWD(128, 16); }
#undef UNIMM_X3M_FWD
    UNIMM_CHECK_LAUNCH();
    return UNIMM_OK;
  }
  // the vector kernels: 64 dimensions per lane (176-184 registers: two waves per SIMD either way, half the cross-lane adds)
  constexpr int DH = UNIMM_X3_ATTN_FWD_DH;
  constexpr int R64 = XA_T / (64 / DH), R128 = XA_T / (128 / DH);
  if (a->D == 64) hipLaunchKernelGGL((x3_attn_fwd_kernel<64, DH>), dim3((a->Tq + R64 - 1) / R64, a->H, a->B), dim3(XA_T), 0, s, p);
  else hipLaunchKernelGGL((x3_attn_fwd_kernel<128, DH>), dim3((a->Tq + R128 - 1) / R128, a->H, a->B), dim3(XA_T), 0, s, p);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_x3_attn_set_impl(int32_t impl) {
  if (impl != 0 && impl != 1) return UNIMM_E_ARG;
  g_x3_attn_impl = impl;
  return UNIMM_OK;
}

extern "C" int unimm_x3_attn_bwd(const unimm_attn_bwd_args* a, const unimm_x3_attn_planes* pl, void* stream) {
  const bool planes = pl != nullptr && (pl->dq3 != nullptr || pl->dk3 != nullptr || pl->dv3 != nullptr);
  if (a == nullptr || !a->out || !a->dout || !a->lse || !a->delta) return UNIMM_E_ARG;
  if (planes ? (!pl->dq3 || !pl->dk3 || !pl->dv3 || g_x3_attn_impl == 0) : (!a->dq || !a->dk || !a->dv)) return UNIMM_E_ARG;
  unimm_attn_args f{};
  f.q = a->q; f.k = a->k; f.v = a->v; f.out = (void*)a->out; f.lse = (float*)a->lse; f.mask = a->mask;
  f.q_off = a->q_off; f.q_len = a->q_len; f.k_off = a->k_off; f.k_len = a->k_len;
  f.order = a->order;
  f.B = a->B; f.H = a->H; f.Tq = a->Tq; f.Tk = a->Tk; f.D = a->D;
  f.ldq = a->ldq; f.ldk = a->ldk; f.ldv = a->ldv; f.ldo = a->ldo;
  f.mask_q_stride = a->mask_q_stride; f.mask_b_stride = a->mask_b_stride; f.scale = a->scale;
  f.drop_key = a->drop_key; f.drop_thr = a->drop_thr; f.drop_scale = a->drop_scale; f.drop_salt = a->drop_salt;
  AttnF32 p;
  const int rc = fill_attn(&f, p);
  if (rc != UNIMM_OK) return rc;
  if ((a->lddo % 4) || (((uintptr_t)a->dout) & 15)) return UNIMM_E_ALIGN;
  p.o = (const float*)a->out; p.out = nullptr; p.dout = (const float*)a->dout; p.delta = a->delta; p.lddo = a->lddo;
  if (planes) {
    if ((((uintptr_t)pl->dq3 | (uintptr_t)pl->dk3 | (uintptr_t)pl->dv3) & 15) || (pl->ld3 % 8) || (pl->cp3 % 8) || pl->cp3 < a->H * a->D ||
        pl->ld3 < 3 * pl->cp3)                                            // a plane narrower than H * D would overlap its neighbour
      return UNIMM_E_ALIGN;
    p.dq3 = (bf16_t*)pl->dq3; p.dk3 = (bf16_t*)pl->dk3; p.dv3 = (bf16_t*)pl->dv3; p.ld3 = pl->ld3; p.cp3 = pl->cp3;
  } else {
    if ((a->lddq % 4) || (a->lddk % 4) || (a->lddv % 4)) return UNIMM_E_ALIGN;
    if (((uintptr_t)a->dq | (uintptr_t)a->dk | (uintptr_t)a->dv) & 15) return UNIMM_E_ALIGN;
    p.dq = (float*)a->dq; p.dk = (float*)a->dk; p.dv = (float*)a->dv;
    p.lddq = a->lddq; p.lddk = a->lddk; p.lddv = a->lddv;
  }
  hipStream_t s = (hipStream_t)stream;
  if (g_x3_attn_impl != 0) {
    const dim3 gq(xm_grid((a->Tq + XM_R - 1) / XM_R, a->H, a->B)), gk(xm_grid((a->Tk + XM_R - 1) / XM_R, a->H, a->B));
    if (a->D == 64) {
      hipLaunchKernelGGL((x3m_attn_bwd_dq_kernel<64>), gq, dim3(XM_T), 0, s, p);
      hipLaunchKernelGGL((x3m_attn_bwd_dkv_kernel<64>), gk, dim3(XM_T), 0, s, p);
    } else {
      hipLaunchKernelGGL((x3m_attn_bwd_dq_kernel<128>), gq, dim3(XM_T), 0, s, p);
      hipLaunchKernelGGL((x3m_attn_bwd_dkv_kernel<128>), gk, dim3(XM_T), 0, s, p);
    }
    UNIMM_CHECK_LAUNCH();
    return UNIMM_OK;
  }
  // the vector kernels: 32 dimensions per lane (the dK / dV kernel holds k, v, dk, dv: 4 x 64 registers at 64 per lane = one wave per SIMD
  // with nothing to hide the LDS round trips behind)
  constexpr int DH = UNIMM_X3_ATTN_BWD_DH;
  if (a->D == 64) {
    constexpr int RPW = XA_T / (64 / DH);
    hipLaunchKernelGGL((x3_attn_bwd_dq_kernel<64, DH>), dim3((a->Tq + RPW - 1) / RPW, a->H, a->B), dim3(XA_T), 0, s, p);
    hipLaunchKernelGGL((x3_attn_bwd_dkv_kernel<64, DH>), dim3((a->Tk + RPW - 1) / RPW, a->H, a->B), dim3(XA_T), 0, s, p);
  } else {
    constexpr int RPW = XA_T / (128 / DH);
    hipLaunchKernelGGL((x3_attn_bwd_dq_kernel<128, DH>), dim3((a->Tq + RPW - 1) / RPW, a->H, a->B), dim3(XA_T), 0, s, p);
    hipLaunchKernelGGL((x3_attn_bwd_dkv_kernel<128, DH>), dim3((a->Tk + RPW - 1) / RPW, a->H, a->B), dim3(XA_T), 0, s, p);
  }
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}
