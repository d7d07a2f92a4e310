// Loss kernels of the UniMM-UL hot path (gfx950): token-level likelihood / unlikelihood over the
// 30522-way vocabulary, masked-region KL, weighted 2-way NSP cross-entropy -- forward and backward.
// One 256-thread workgroup per row; wavefront-reduced online log-sum-exp in fp32 (logits, lse and
// 1-p are never rounded to bf16: models/vilbert_dialog.py:1587 needs log(clamp(1-p, 1e-6))).
#include "common.h"

namespace {

__device__ __forceinline__ float block_sum(float v, float* red) {
  v = wave_sum(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return red[0] + red[1] + red[2] + red[3];
}
__device__ __forceinline__ float block_max(float v, float* red) {
  v = wave_max(v);
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
  __syncthreads();
  return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// log-sum-exp of one fp32 row (16-byte loads when `vec`), block-wide
__device__ __forceinline__ float row_lse(const float* __restrict__ z, int V, bool vec, float* red) {
  float m = -INFINITY, s = 0.f;
  if (vec) {
    const int nv = V >> 2;
    int i = threadIdx.x;
    for (; i + 768 < nv; i += 1024) {        // four 16-byte loads in flight per lane (the row is read once, from HBM)
      f32x4 a[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) a[u] = *reinterpret_cast<const f32x4*>(z + 4 * (i + 256 * u));
      float mm = -INFINITY;
#pragma unroll
      for (int u = 0; u < 4; ++u) mm = fmaxf(mm, fmaxf(fmaxf(a[u][0], a[u][1]), fmaxf(a[u][2], a[u][3])));
      if (mm > m) { s *= __expf(m - mm); m = mm; }
#pragma unroll
      for (int u = 0; u < 4; ++u) s += __expf(a[u][0] - m) + __expf(a[u][1] - m) + __expf(a[u][2] - m) + __expf(a[u][3] - m);
    }
    for (; i < nv; i += 256) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(z + 4 * i);
      const float mm = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
      if (mm > m) { s *= __expf(m - mm); m = mm; }
      s += __expf(a[0] - m) + __expf(a[1] - m) + __expf(a[2] - m) + __expf(a[3] - m);
    }
    for (int i = (nv << 2) + threadIdx.x; i < V; i += 256) {
      const float a = z[i];
      if (a > m) { s *= __expf(m - a); m = a; }
      s += __expf(a - m);
    }
  } else {
    for (int i = threadIdx.x; i < V; i += 256) {
      const float a = z[i];
      if (a > m) { s *= __expf(m - a); m = a; }
      s += __expf(a - m);
    }
  }
  const float gm = block_max(m, red);
  const float gs = block_sum(m == -INFINITY ? 0.f : s * __expf(m - gm), red);
  return gm + logf(gs);
}

// ---- MLM likelihood / unlikelihood (models/vilbert_dialog.py:1577-1604) ---------------------------
__global__ __launch_bounds__(256) void lm_loss_fwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ weights, float* __restrict__ rowloss,
                                                          float* __restrict__ rownll, float* __restrict__ lse_o, int V,
                                                          int ld, float clamp_min, const int32_t* __restrict__ n_dev) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  if (n_dev != nullptr && row >= n_dev[0]) return;      // rows of the launch's capacity beyond the step's real count
  const float* z = logits + (size_t)row * ld;
  const float lse = row_lse(z, V, (ld & 3) == 0, red);
  if (threadIdx.x == 0) {
    const int y = labels[row], w = weights[row];
    float loss = 0.f, nll = 0.f;
    if (y >= 0) {
      const float logp = z[y] - lse;
      nll = -logp;
      if (w > 0) loss = -logp * (float)w;
      else if (w == -1) loss = -logf(fmaxf(1.0f - expf(logp), clamp_min));
    }
    rowloss[row] = loss;
    rownll[row] = nll;
    lse_o[row] = lse;
  }
}

__global__ __launch_bounds__(256) void lm_loss_bwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                          const int32_t* __restrict__ weights, const float* __restrict__ lse_i,
                                                          const float* __restrict__ g, float inv_denom,
                                                          bf16_t* __restrict__ dlogits, int V, int ld, int ldd,
                                                          float clamp_min, const int32_t* __restrict__ n_dev,
                                                          const float* __restrict__ inv_dev) {
  const int row = blockIdx.x;
  if (n_dev != nullptr && row >= n_dev[0]) return;
  if (inv_dev != nullptr) inv_denom = inv_dev[0];
  const float* z = logits + (size_t)row * ld;
  bf16_t* dz = dlogits + (size_t)row * ldd;
  const int y = labels[row], w = weights[row];
  const float lse = lse_i[row];
  float coef = 0.f;
  if (y >= 0) {
    const float gs = g[0] * inv_denom;
    if (w > 0) coef = gs * (float)w;
    else if (w == -1) {
      const float py = expf(z[y] - lse);
      const float om = 1.0f - py;
      coef = om >= clamp_min ? -gs * py / om : 0.f;   // d/dz of -log(clamp(1-p_y)): zero once clamped
    }
  }
  // 8 columns per lane and iteration: two 16-byte loads, one 16-byte store (ldd % 8 == 0; columns >= V are written as
  // zeros: K padding of the dgrad GEMM).  Rows of the logits are 16-byte aligned when ld % 4 == 0.
  const int n8 = ldd >> 3;
  const bool vec = (ld & 3) == 0;
  for (int i = threadIdx.x; i < n8; i += 256) {
    const int c0 = 8 * i;
    float v[8];
    if (vec && c0 + 8 <= V) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(z + c0), b2 = *reinterpret_cast<const f32x4*>(z + c0 + 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = a[e]; v[4 + e] = b2[e]; }
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = coef * (__expf(v[e] - lse) - (c0 + e == y ? 1.0f : 0.0f));
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        v[e] = (c < V) ? coef * (__expf(z[c] - lse) - (c == y ? 1.0f : 0.0f)) : 0.f;
      }
    }
    *reinterpret_cast<u32x4*>(dz + c0) = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  }
}

// ---- masked-region KL (models/vilbert_dialog.py:1569-1574) ----------------------------------------
__global__ __launch_bounds__(256) void kl_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                          const int32_t* __restrict__ label, float* __restrict__ rowloss,
                                                          float* __restrict__ lse_o, int C, int ld) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  const float* z = pred + (size_t)row * ld;
  const float lse = row_lse(z, C, false, red);
  float acc = 0.f;
  if (label[row] == 1) {
    const float* t = target + (size_t)row * C;
    for (int i = threadIdx.x; i < C; i += 256) {
      const float ti = t[i];
      if (ti > 0.f) acc += ti * (__logf(ti) - (z[i] - lse));   // KLDivLoss: t*(log t - input), 0 where t == 0
    }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) { rowloss[row] = acc; lse_o[row] = lse; }
}

__global__ __launch_bounds__(256) void kl_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                          const int32_t* __restrict__ label, const float* __restrict__ lse_i,
                                                          const float* __restrict__ g, float inv_denom,
                                                          bf16_t* __restrict__ dpred, int C, int ld, int ldd,
                                                          const float* __restrict__ inv_dev) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  if (inv_dev != nullptr) inv_denom = inv_dev[0];
  const float* z = pred + (size_t)row * ld;
  const float* t = target + (size_t)row * C;
  bf16_t* dz = dpred + (size_t)row * ldd;
  const bool on = label[row] == 1;
  float ts = 0.f;
  if (on)
    for (int i = threadIdx.x; i < C; i += 256) ts += t[i];
  ts = block_sum(ts, red);
  const float gs = on ? g[0] * inv_denom : 0.f;
  const float lse = lse_i[row];
  for (int i = threadIdx.x; i < ldd; i += 256) {
    float v = 0.f;
    if (on && i < C) v = gs * (__expf(z[i] - lse) * ts - t[i]);
    dz[i] = f2bf(v);
  }
}

// ---- masked-region MSE, the predict_feature branch (models/vilbert_dialog.py:1562-1566) ---------------
// rowloss = [label == 1] * sum_j (pred_j - target_j)^2 / C  (the caller divides by max(#selected rows, 1): the reference
// divides by the number of selected ELEMENTS)
__global__ __launch_bounds__(256) void mse_loss_fwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           const int32_t* __restrict__ label, float* __restrict__ rowloss, int C,
                                                           int ld) {
  __shared__ float red[4];
  const int row = blockIdx.x;
  float acc = 0.f;
  if (label[row] == 1) {
    const float* z = pred + (size_t)row * ld;
    const float* t = target + (size_t)row * C;
    for (int i = threadIdx.x; i < C; i += 256) { const float d = z[i] - t[i]; acc += d * d; }
  }
  acc = block_sum(acc, red);
  if (threadIdx.x == 0) rowloss[row] = acc / (float)C;
}

// dpred = [label == 1] * g * inv_denom * 2 (pred - target) / C; out_split == 0: bf16 [rows, ldd] (columns >= C zero),
// else an x-type split operand [rows, 3 ldd] (the fp32-accuracy mode, csrc/x3ops.hip)
__global__ __launch_bounds__(256) void mse_loss_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                           const int32_t* __restrict__ label, const float* __restrict__ g,
                                                           float inv_denom, bf16_t* __restrict__ dpred, int C, int ld, int ldd,
                                                           int out_split) {
  const int row = blockIdx.x;
  const float* z = pred + (size_t)row * ld;
  const float* t = target + (size_t)row * C;
  const float gs = label[row] == 1 ? g[0] * inv_denom * 2.0f / (float)C : 0.f;
  for (int i = threadIdx.x; i < ldd; i += 256) {
    const float v = (gs != 0.f && i < C) ? gs * (z[i] - t[i]) : 0.f;
    if (out_split == 0) {
      dpred[(size_t)row * ldd + i] = f2bf(v);
    } else {
      const bf16_t hi = f2bf(v), lo = f2bf(v - bf2f(hi));
      bf16_t* o = dpred + (size_t)row * 3 * ldd + i;
      o[0] = hi; o[ldd] = lo; o[2 * ldd] = hi;
    }
  }
}

// ---- weighted NSP cross-entropy (models/vilbert_dialog.py:1605-1621), single workgroup -----------
__global__ __launch_bounds__(256) void nsp_loss_fwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                           float w0, float w1, float* __restrict__ loss, int B, int ld) {
  __shared__ float red[4];
  float num = 0.f, den = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float a = logits[(size_t)i * ld], b = logits[(size_t)i * ld + 1];
    const float m = fmaxf(a, b);
    const float lse = m + __logf(__expf(a - m) + __expf(b - m));
    const int y = labels[i];
    const float w = y == 0 ? w0 : w1;
    num += w * (lse - (y == 0 ? a : b));
    den += w;
  }
  num = block_sum(num, red);
  den = block_sum(den, red);
  if (threadIdx.x == 0) loss[0] = num / den;
}

__global__ __launch_bounds__(256) void nsp_loss_bwd_kernel(const float* __restrict__ logits, const int32_t* __restrict__ labels,
                                                           float w0, float w1, const float* __restrict__ g,
                                                           const float* __restrict__ extra, float* __restrict__ dlogits,
                                                           int B, int ld, int ldd) {
  __shared__ float red[4];
  float den = 0.f;
  for (int i = threadIdx.x; i < B; i += 256) den += labels[i] == 0 ? w0 : w1;
  den = block_sum(den, red);
  const float gs = g[0] / den;
  for (int i = threadIdx.x; i < B; i += 256) {
    const float a = logits[(size_t)i * ld], b = logits[(size_t)i * ld + 1];
    const float m = fmaxf(a, b);
    const float ea = __expf(a - m), eb = __expf(b - m);
    const float inv = 1.0f / (ea + eb);
    const int y = labels[i];
    const float w = (y == 0 ? w0 : w1) * gs;
    float* d = dlogits + (size_t)i * ldd;
    d[0] = w * (ea * inv - (y == 0 ? 1.f : 0.f)) + (extra != nullptr ? extra[2 * i] : 0.f);
    d[1] = w * (eb * inv - (y == 1 ? 1.f : 0.f)) + (extra != nullptr ? extra[2 * i + 1] : 0.f);
    for (int c = 2; c < ldd; ++c) d[c] = 0.f;
  }
}

// dst[0] = scale * sum(src[0..n))  -- single workgroup, fixed order (deterministic)
__global__ __launch_bounds__(256) void reduce_sum_kernel(const float* __restrict__ src, int64_t n, float* __restrict__ dst,
                                                         float scale, const int32_t* __restrict__ n_dev,
                                                         const float* __restrict__ scale_dev) {
  __shared__ float red[4];
  if (n_dev != nullptr) n = n_dev[0] < n ? n_dev[0] : n;
  if (scale_dev != nullptr) scale = scale_dev[0];
  float s = 0.f;
  for (int64_t i = threadIdx.x; i < n; i += 256) s += src[i];
  s = block_sum(s, red);
  if (threadIdx.x == 0) dst[0] = s * scale;
}

// dst[seg[i]] += sign * src[i]   (per-sequence log-likelihood: val_lm.py:133-136)
__global__ void segment_sum_kernel(const float* __restrict__ src, const int32_t* __restrict__ seg, float* __restrict__ dst,
                                   int64_t n, float sign) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) atomicAdd(dst + seg[i], sign * src[i]);
}

}  // namespace

extern "C" int unimm_lm_loss_fwd(const float* logits, const int32_t* labels, const int32_t* weights, float* rowloss,
                                 float* rownll, float* lse, int32_t n, int32_t V, int32_t ld, const int32_t* n_dev, void* stream) {
  if (!logits || !labels || !weights || !rowloss || !rownll || !lse) return UNIMM_E_ARG;
  if (n <= 0 || V <= 0 || ld < V) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(lm_loss_fwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, labels, weights, rowloss,
                     rownll, lse, V, ld, 1e-6f, n_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_lm_loss_bwd(const float* logits, const int32_t* labels, const int32_t* weights, const float* lse,
                                 const float* g, float inv_denom, void* dlogits, int32_t n, int32_t V, int32_t ld,
                                 int32_t ldd, const int32_t* n_dev, const float* inv_dev, void* stream) {
  if (!logits || !labels || !weights || !lse || !g || !dlogits) return UNIMM_E_ARG;
  if (n <= 0 || V <= 0 || ld < V || ldd < V || (ldd % 8)) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(lm_loss_bwd_kernel, dim3(n), dim3(256), 0, (hipStream_t)stream, logits, labels, weights, lse, g,
                     inv_denom, (bf16_t*)dlogits, V, ld, ldd, 1e-6f, n_dev, inv_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_kl_loss_fwd(const float* pred, const float* target, const int32_t* label, float* rowloss, float* lse,
                                 int32_t rows, int32_t C, int32_t ld, void* stream) {
  if (!pred || !target || !label || !rowloss || !lse) return UNIMM_E_ARG;
  if (rows <= 0 || C <= 0 || ld < C) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(kl_loss_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, target, label, rowloss, lse,
                     C, ld);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_kl_loss_bwd(const float* pred, const float* target, const int32_t* label, const float* lse,
                                 const float* g, float inv_denom, void* dpred, int32_t rows, int32_t C, int32_t ld,
                                 int32_t ldd, const float* inv_dev, void* stream) {
  if (!pred || !target || !label || !lse || !g || !dpred) return UNIMM_E_ARG;
  if (rows <= 0 || C <= 0 || ld < C || ldd < C) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(kl_loss_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, target, label, lse, g,
                     inv_denom, (bf16_t*)dpred, C, ld, ldd, inv_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_mse_loss_fwd(const float* pred, const float* target, const int32_t* label, float* rowloss, int32_t rows,
                                  int32_t C, int32_t ld, void* stream) {
  if (!pred || !target || !label || !rowloss) return UNIMM_E_ARG;
  if (rows <= 0 || C <= 0 || ld < C) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(mse_loss_fwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, target, label, rowloss, C, ld);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_mse_loss_bwd(const float* pred, const float* target, const int32_t* label, const float* g, float inv_denom,
                                  void* dpred, int32_t rows, int32_t C, int32_t ld, int32_t ldd, int32_t out_split, void* stream) {
  if (!pred || !target || !label || !g || !dpred) return UNIMM_E_ARG;
  if (rows <= 0 || C <= 0 || ld < C || ldd < C) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(mse_loss_bwd_kernel, dim3(rows), dim3(256), 0, (hipStream_t)stream, pred, target, label, g, inv_denom,
                     (bf16_t*)dpred, C, ld, ldd, out_split);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_nsp_loss_fwd(const float* logits, const int32_t* labels, float w0, float w1, float* loss, int32_t B,
                                  int32_t ld, void* stream) {
  if (!logits || !labels || !loss || B <= 0 || ld < 2) return UNIMM_E_ARG;
  hipLaunchKernelGGL(nsp_loss_fwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, w0, w1, loss, B, ld);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_nsp_loss_bwd(const float* logits, const int32_t* labels, float w0, float w1, const float* g,
                                  const float* extra, float* dlogits, int32_t B, int32_t ld, int32_t ldd, void* stream) {
  if (!logits || !labels || !g || !dlogits || B <= 0 || ld < 2 || ldd < 2) return UNIMM_E_ARG;
  hipLaunchKernelGGL(nsp_loss_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, logits, labels, w0, w1, g,
                     extra, dlogits, B, ld, ldd);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_reduce_sum(const float* src, int64_t n, float* dst, float scale, const int32_t* n_dev,
                                const float* scale_dev, void* stream) {
  if (!src || !dst || n <= 0) return UNIMM_E_ARG;
  hipLaunchKernelGGL(reduce_sum_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, src, n, dst, scale, n_dev, scale_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_segment_sum(const float* src, const int32_t* seg, float* dst, int64_t n, float sign, void* stream) {
  if (!src || !seg || !dst || n <= 0) return UNIMM_E_ARG;
  hipLaunchKernelGGL(segment_sum_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, seg,
                     dst, n, sign);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}
