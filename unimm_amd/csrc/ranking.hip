// NeuralNDCG-transposed of one slate of answer options, value and gradient in one launch
// (SURVEY.md 8 row F4).
//
// Replaces, for the dense-annotation fine-tuning objective (dense_annotation_finetuning.py:286-288),
// the chain `deterministic_neural_sort` -> `sinkhorn_scaling` -> expected discounts -> NDCG of
// utils/rank_loss.py:518-581 (:79-112, :55-78, :18-54) and its autograd backward.  On the device that
// chain is ~25 tiny kernels per Sinkhorn sweep plus a host sync for the stop test, for up to 50 sweeps,
// and the same again backwards: ~5 ms of launch latency per step next to a 29 ms encoder step.  Here
// one workgroup owns a slate (n <= 128 options, the reference uses 100) and keeps everything in LDS.
//
// Formulation.  Sinkhorn only rescales rows and columns, so the relaxed permutation P (row-softmax of
// ((m+1-2i) s_j - sum_k |s_j - s_k|) / tau, utils/rank_loss.py:79-112) is never rewritten: sweep t
// keeps two vectors with M_t = diag(u_t) P diag(v_t),
//     c = v * (P^T u),  v <- v / max(c, 1e-8);      r = u * (P v),  u <- u / max(r, 1e-8),
// and stops when max_j |v_j (P^T u)_j - 1| < tol (the row marginal is 1 up to rounding right after the
// row step), the same test the reference evaluates every sweep.  Each sweep is two 128x128 mat-vecs out
// of LDS.  u_t, v_t of every sweep are kept (sign bit = "the 1e-8 clamp was active"), which is all the
// reverse sweep needs: with the clamp inactive v_t = 1 / (P^T u_{t-1}) and u_t = 1 / (P v_t), so
//     dP += -(ubar u_t^2) v_t^T - u_{t-1} (vbar v_t^2)^T        (two rank-1 updates, in registers)
// and the adjoints move through the same two mat-vecs.  The softmax / |s_j - s_k| backward follows.
#include "common.h"

namespace {

constexpr int NMAX = UNIMM_NDCG_MAX_OPTIONS;   // 128
constexpr int LD = NMAX + 1;                   // odd stride: row- and column-wise walks are both conflict-free
constexpr int TMAX = UNIMM_NDCG_MAX_ITER;      // 64
constexpr int NTHREADS = 256;
constexpr float SK_EPS = 1e-8f;                // utils/rank_loss.py:6

struct NdcgParams {
  const float* pred; const float* truth; float* ndcg; float* alive; float* dpred; int32_t* iters;
  int n, k, powered, max_iter;
  float pad, inv_tau, tol;
};

struct Lds {
  float M[NMAX * LD];
  float uh[TMAX * NMAX], vh[TMAX * NMAX];
  float s[NMAX], a[NMAX], slope[NMAX], dsc[NMAX], gp[NMAX], ok[NMAX];
  float u[NMAX], v[NMAX], ub[NMAX], vb[NMAX], x[NMAX], y[NMAX], w[NMAX], z[NMAX];
  float part[2 * NMAX];
  float red[8];
};

// out_j = sum_i in_i M[i][j]      (in must be 0 on rows >= n)
__device__ __forceinline__ void matvec_t(Lds& L, float* out, const float* in, int n) {
  const int j = threadIdx.x & (NMAX - 1), h = threadIdx.x >> 7;
  const int half = (n + 1) >> 1, lo = h * half, hi = min(n, lo + half);
  float acc = 0.f;
#pragma unroll 4
  for (int i = lo; i < hi; ++i) acc = fmaf(in[i], L.M[i * LD + j], acc);
  L.part[h * NMAX + j] = acc;
  __syncthreads();
  if (threadIdx.x < NMAX) out[threadIdx.x] = L.part[threadIdx.x] + L.part[NMAX + threadIdx.x];
  __syncthreads();
}

// out_i = sum_j M[i][j] in_j
__device__ __forceinline__ void matvec(Lds& L, float* out, const float* in, int n) {
  const int i = threadIdx.x & (NMAX - 1), h = threadIdx.x >> 7;
  const int half = (n + 1) >> 1, lo = h * half, hi = min(n, lo + half);
  float acc = 0.f;
#pragma unroll 4
  for (int j = lo; j < hi; ++j) acc = fmaf(L.M[i * LD + j], in[j], acc);
  L.part[h * NMAX + i] = acc;
  __syncthreads();
  if (threadIdx.x < NMAX) out[threadIdx.x] = L.part[threadIdx.x] + L.part[NMAX + threadIdx.x];
  __syncthreads();
}

__device__ __forceinline__ float block_sum(Lds& L, float v) {
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) L.red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = L.red[0] + L.red[1] + L.red[2] + L.red[3];
  __syncthreads();
  return r;
}

__device__ __forceinline__ float block_max(Lds& L, float v) {
  v = wave_max(v);
  if ((threadIdx.x & 63) == 0) L.red[threadIdx.x >> 6] = v;
  __syncthreads();
  const float r = fmaxf(fmaxf(L.red[0], L.red[1]), fmaxf(L.red[2], L.red[3]));
  __syncthreads();
  return r;
}

__device__ __forceinline__ float sgn(float x) { return (x > 0.f) ? 1.f : ((x < 0.f) ? -1.f : 0.f); }

__global__ __launch_bounds__(NTHREADS) void neural_ndcg_kernel(NdcgParams P) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  Lds& L = *reinterpret_cast<Lds*>(smem);
  const int tid = threadIdx.x, n = P.n, slate = blockIdx.x;
  const float* pred = P.pred + (size_t)slate * n;
  const float* truth = P.truth + (size_t)slate * n;

  // ---- per-option quantities ----------------------------------------------------------------------
  float my_y = 0.f;
  if (tid < NMAX) {
    const bool in = tid < n;
    my_y = in ? truth[tid] : P.pad;
    const bool ok = in && my_y != P.pad;
    L.ok[tid] = ok ? 1.f : 0.f;
    L.s[tid] = ok ? pred[tid] : 0.f;
    L.y[tid] = ok ? my_y : 0.f;
  }
  __syncthreads();
  const float m_valid = block_sum(L, tid < NMAX ? L.ok[tid] : 0.f);
  const int m = (int)(m_valid + 0.5f);
  const int kk = min(P.k, n);
  float idcg_part = 0.f;
  if (tid < NMAX) {
    const bool ok = L.ok[tid] != 0.f;
    float a = 0.f;
    int pos = 0;
    if (ok) {
      const float sj = L.s[tid], yj = L.y[tid];
      for (int q = 0; q < n; ++q) {
        if (L.ok[q] == 0.f) continue;
        a += fabsf(sj - L.s[q]);
        pos += (L.y[q] > yj) || (L.y[q] == yj && q < tid);
      }
      // ideal DCG always uses the 2^y - 1 gain (utils/rank_loss.py:565-570 call dcg() with its default)
      if (pos < kk) idcg_part = (exp2f(yj) - 1.f) / log2f((float)pos + 2.f);
    }
    L.a[tid] = a;
    const int posr = tid + 1;                                    // row = rank position, raw index
    L.slope[tid] = (ok && posr <= m) ? (float)(m + 1 - 2 * posr) : 0.f;
    L.dsc[tid] = (ok && tid < kk) ? 1.f / log2f((float)tid + 2.f) : 0.f;
    L.u[tid] = ok ? 1.f : 0.f;
    L.v[tid] = ok ? 1.f : 0.f;
  }
  __syncthreads();
  const float idcg = block_sum(L, idcg_part);
  if (tid < NMAX) {
    const bool ok = L.ok[tid] != 0.f;
    const float gain = P.powered ? (exp2f(L.y[tid]) - 1.f) : L.y[tid];
    L.gp[tid] = ok ? gain / (idcg + SK_EPS) : 0.f;
  }

  // ---- relaxed permutation: row softmax over the unpadded columns ------------------------------------
  {
    const int wave = tid >> 6, lane = tid & 63;
    for (int i = wave; i < NMAX; i += NTHREADS / 64) {
      const bool row_ok = i < n && L.ok[i] != 0.f;
      const float sl = L.slope[i];
      float l0 = -INFINITY, l1 = -INFINITY;
      if (row_ok && L.ok[lane] != 0.f) l0 = (sl * L.s[lane] - L.a[lane]) * P.inv_tau;
      if (row_ok && L.ok[lane + 64] != 0.f) l1 = (sl * L.s[lane + 64] - L.a[lane + 64]) * P.inv_tau;
      const float mx = wave_max(fmaxf(l0, l1));
      const float e0 = (l0 == -INFINITY) ? 0.f : __expf(l0 - mx), e1 = (l1 == -INFINITY) ? 0.f : __expf(l1 - mx);
      const float sum = wave_sum(e0 + e1);
      const float inv = row_ok ? 1.f / sum : 0.f;
      L.M[i * LD + lane] = e0 * inv;
      L.M[i * LD + lane + 64] = e1 * inv;
    }
  }
  __syncthreads();

  // ---- Sinkhorn sweeps on the scaling vectors ---------------------------------------------------------
  matvec_t(L, L.x, L.u, n);                        // x = P^T u
  int T = 0;
  for (int t = 0; t < P.max_iter; ++t) {
    if (tid < NMAX) {
      float vj = L.v[tid], keep = 0.f;
      if (L.ok[tid] != 0.f) {
        const float c = vj * L.x[tid];
        const bool cl = c < SK_EPS;
        vj = cl ? vj / SK_EPS : vj / c;
        keep = cl ? -vj : vj;
      }
      L.v[tid] = vj;
      L.vh[t * NMAX + tid] = keep;
    }
    __syncthreads();
    matvec(L, L.x, L.v, n);                        // x = P v
    if (tid < NMAX) {
      float ui = L.u[tid], keep = 0.f;
      if (L.ok[tid] != 0.f) {
        const float r = ui * L.x[tid];
        const bool cl = r < SK_EPS;
        ui = cl ? ui / SK_EPS : ui / r;
        keep = cl ? -ui : ui;
      }
      L.u[tid] = ui;
      L.uh[t * NMAX + tid] = keep;
    }
    __syncthreads();
    matvec_t(L, L.x, L.u, n);                      // column marginals of the new iterate / v
    float e = 0.f;
    if (tid < NMAX && L.ok[tid] != 0.f) e = fabsf(L.v[tid] * L.x[tid] - 1.f);
    const float err = block_max(L, e);
    T = t + 1;
    if (err < P.tol) break;                        // block-uniform
  }

  // ---- NDCG of the expected discounts ------------------------------------------------------------------
  if (tid < NMAX) {
    L.w[tid] = L.dsc[tid] * L.u[tid];
    L.z[tid] = L.v[tid] * L.gp[tid];
  }
  __syncthreads();
  matvec_t(L, L.x, L.w, n);                        // x_j = sum_i d_i u_i P_ij   (expected discount / v_j)
  const float ndcg = block_sum(L, tid < NMAX ? L.z[tid] * L.x[tid] : 0.f);
  float* dp = P.dpred + (size_t)slate * n;
  if (idcg == 0.f) {                               // no relevant option: the slate is left out of the mean
    if (tid == 0) { P.ndcg[slate] = 0.f; P.alive[slate] = 0.f; P.iters[slate] = T; }
    if (tid < n) dp[tid] = 0.f;
    return;
  }
  if (tid == 0) { P.ndcg[slate] = ndcg; P.alive[slate] = 1.f; P.iters[slate] = T; }

  // ---- reverse sweep -------------------------------------------------------------------------------------
  if (tid < NMAX) L.vb[tid] = L.gp[tid] * L.x[tid];
  __syncthreads();
  matvec(L, L.x, L.z, n);
  if (tid < NMAX) L.ub[tid] = L.dsc[tid] * L.x[tid];
  const int ti = tid >> 4, tj = tid & 15;          // this thread's 8x8 entries of dP: rows ti+16a, columns tj+16b
  float acc[8][8];
#pragma unroll
  for (int a = 0; a < 8; ++a)
#pragma unroll
    for (int b = 0; b < 8; ++b) acc[a][b] = L.w[ti + 16 * a] * L.z[tj + 16 * b];
  __syncthreads();

  for (int t = T - 1; t >= 0; --t) {
    // u_t = u_{t-1} / max(u_{t-1} (P v_t), eps)
    if (tid < NMAX) {
      const float ut = L.uh[t * NMAX + tid], ub = L.ub[tid];
      L.x[tid] = (ut > 0.f) ? -ub * ut * ut : 0.f;                    // adjoint of q = P v_t
      L.ub[tid] = (ut < 0.f) ? ub / SK_EPS : 0.f;                     // what reaches u_{t-1} directly
      L.z[tid] = fabsf(L.vh[t * NMAX + tid]);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = fmaf(L.x[ti + 16 * a], L.z[tj + 16 * b], acc[a][b]);
    matvec_t(L, L.y, L.x, n);
    // v_t = v_{t-1} / max(v_{t-1} (P^T u_{t-1}), eps)
    if (tid < NMAX) {
      const float vt = L.vh[t * NMAX + tid], vb = L.vb[tid] + L.y[tid];
      L.x[tid] = (vt > 0.f) ? -vb * vt * vt : 0.f;                    // adjoint of p = P^T u_{t-1}
      L.vb[tid] = (vt < 0.f) ? vb / SK_EPS : 0.f;
      L.w[tid] = (t > 0) ? fabsf(L.uh[(t - 1) * NMAX + tid]) : L.ok[tid];
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 8; ++a)
#pragma unroll
      for (int b = 0; b < 8; ++b) acc[a][b] = fmaf(L.w[ti + 16 * a], L.x[tj + 16 * b], acc[a][b]);
    matvec(L, L.y, L.x, n);
    if (tid < NMAX) L.ub[tid] += L.y[tid];
    __syncthreads();
  }

  // ---- softmax backward, in place: M <- dLogits ------------------------------------------------------------
#pragma unroll
  for (int a = 0; a < 8; ++a) {
    const int i = ti + 16 * a;
    float pr[8], dot = 0.f;
#pragma unroll
    for (int b = 0; b < 8; ++b) { pr[b] = L.M[i * LD + tj + 16 * b]; dot = fmaf(acc[a][b], pr[b], dot); }
    dot += __shfl_xor(dot, 1, 64); dot += __shfl_xor(dot, 2, 64);
    dot += __shfl_xor(dot, 4, 64); dot += __shfl_xor(dot, 8, 64);   // the 16 lanes that share row i
#pragma unroll
    for (int b = 0; b < 8; ++b) L.M[i * LD + tj + 16 * b] = pr[b] * (acc[a][b] - dot) * P.inv_tau;
  }
  __syncthreads();
  matvec_t(L, L.x, L.slope, n);                    // d/ds_j through the slope term
  matvec_t(L, L.y, L.ok, n);                       // -(adjoint of a_j)
  if (tid < n) {
    float g = 0.f;
    if (L.ok[tid] != 0.f) {
      const float sj = L.s[tid], aj = -L.y[tid];
      g = L.x[tid];
      for (int q = 0; q < n; ++q)
        if (L.ok[q] != 0.f) g = fmaf(sgn(sj - L.s[q]), aj - L.y[q], g);
    }
    dp[tid] = g;
  }
}

}  // namespace

extern "C" int unimm_neural_ndcg(const unimm_ndcg_args* a, void* stream) {
  if (a == nullptr || a->pred == nullptr || a->truth == nullptr || a->ndcg == nullptr || a->alive == nullptr ||
      a->dpred == nullptr || a->iters == nullptr)
    return UNIMM_E_ARG;
  if (a->slates < 1 || a->n < 1 || a->n > NMAX) return UNIMM_E_SHAPE;
  if (a->max_iter < 1 || a->max_iter > TMAX || !(a->temperature > 0.f)) return UNIMM_E_ARG;
  NdcgParams p;
  p.pred = a->pred; p.truth = a->truth; p.ndcg = a->ndcg; p.alive = a->alive; p.dpred = a->dpred; p.iters = a->iters;
  p.n = a->n; p.k = (a->k <= 0) ? a->n : a->k; p.powered = a->powered_relevancies; p.max_iter = a->max_iter;
  p.pad = a->pad_label; p.inv_tau = 1.0f / a->temperature; p.tol = a->tol;
  static bool attr_set = false;
  if (!attr_set) {
    if (hipFuncSetAttribute((const void*)neural_ndcg_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sizeof(Lds)) !=
        hipSuccess)
      return UNIMM_E_HIP;
    attr_set = true;
  }
  hipLaunchKernelGGL(neural_ndcg_kernel, dim3((unsigned)a->slates), dim3(NTHREADS), sizeof(Lds), (hipStream_t)stream, p);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}
