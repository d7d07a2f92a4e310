// fp32 head arithmetic of the UniMM-UL hot path (gfx950): the two poolers, the fused pooled vector and the NSP head.
//
// The reference computes these in fp32 from the fp32 encoder output (models/vilbert_dialog.py:946-952, :961-967,
// :1064-1070; CPU path = everything fp32).  They are [B, 768..1024] x [1024, 768..1024] products - 0.4 GFLOP per
// step - but sit on top of the whole network: with bf16 operands a ReLU unit of one of the B pooled rows switches
// with the last bit of the forward and the two pooler gradients differed from the reference's by 8-16 % (round 2).
// So they run on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32 (the fp32 VECTOR rate, which is plenty),
// straight from the fp32 master weights, forward and backward.
//
//   unimm_linear_f32 : OUT[M,N] (+)= act( A[M,K] . B[K,N] + bias ), every operand fp32 with free element strides,
//                      so one kernel serves y = x W^T + b, dx = dy W and dW += dy^T x (+ db += colsum(dy))
//   unimm_rows_add_f32 : bf16 rows dst[idx[r], :] += fp32 src[r, :]   (pooler input gradient -> first-token rows)
#include "common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) float v4f;

struct LinF32 {
  const float* a; const float* b; const float* bias; float* out; float* rowsum;
  int M, N, K;
  long sa_m, sa_k, sb_k, sb_n, ldo;
  int relu, accumulate;
};

// One workgroup per 16x16 output tile; its four waves split the reduction (K) four ways and meet in LDS.  With one wave
// per tile and one dependent load -> MFMA chain of K / 16 steps, the 9 launches of a step were pure latency (48 us each for
// 0.4 GFLOP in all: 960-3,072 waves on 1,024 SIMDs, nothing to hide a load behind); a quarter of the chain per wave and four
// chunks of loads in flight per wave bring them to a few microseconds.
// v_mfma_f32_16x16x4_f32: lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15]; D[row = 4 (l >> 4) + r]
// [col = l & 15] in register r.  Bit for bit a k-ordered fmaf chain per wave; the four partial sums are added in wave order.
__global__ __launch_bounds__(256) void linear_f32_kernel(LinF32 p) {
  __shared__ float red[3][64][4];
  __shared__ float reds[3][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int tiles_n = (p.N + 15) >> 4;
  const int tile = blockIdx.x;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int i = lane & 15, q = lane >> 4;
  const int am = tm * 16 + i, bn = tn * 16 + i;
  const bool a_ok = am < p.M, b_ok = bn < p.N;
  const float* ap = p.a + (long)(a_ok ? am : 0) * p.sa_m;
  const float* bp = p.b + (long)(b_ok ? bn : 0) * p.sb_n;
  v4f acc = {0.f, 0.f, 0.f, 0.f};
  float asum = 0.f;                                           // sum over k of this lane's A elements (row sums of A)
  // this wave's share of the reduction: whole 16-deep chunks
  const int chunks = (p.K + 15) >> 4, per = (chunks + 3) >> 2;
  const int kbeg = wave * per * 16;
  int kend = kbeg + per * 16;
  kend = kend < p.K ? kend : p.K;
  if (p.sa_k == 1 && p.sb_k == 1 && (p.K & 15) == 0 && ((p.sa_m | p.sb_n) & 3) == 0 &&
      ((((uintptr_t)p.a) | ((uintptr_t)p.b)) & 15) == 0) {
    // both operands reduction-contiguous: 16-byte loads; MFMA e of a 16-deep chunk sums k = k0 + 4 q' + e over q' = 0..3
    // (the same k permutation on both operands, so every product of the chunk is taken exactly once)
    int k0 = kbeg;
    for (; k0 + 64 <= kend; k0 += 64) {                       // four chunks of loads in flight
      v4f av[4], bv[4];
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        av[c] = *reinterpret_cast<const v4f*>(ap + k0 + 16 * c + 4 * q);
        bv[c] = *reinterpret_cast<const v4f*>(bp + k0 + 16 * c + 4 * q);
      }
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        if (!a_ok) av[c] = v4f{0.f, 0.f, 0.f, 0.f};
        if (!b_ok) bv[c] = v4f{0.f, 0.f, 0.f, 0.f};
        asum += (av[c][0] + av[c][1]) + (av[c][2] + av[c][3]);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c][e], bv[c][e], acc, 0, 0, 0);
      }
    }
    for (; k0 < kend; k0 += 16) {
      v4f av = *reinterpret_cast<const v4f*>(ap + k0 + 4 * q);
      v4f bv = *reinterpret_cast<const v4f*>(bp + k0 + 4 * q);
      if (!a_ok) av = v4f{0.f, 0.f, 0.f, 0.f};
      if (!b_ok) bv = v4f{0.f, 0.f, 0.f, 0.f};
      asum += (av[0] + av[1]) + (av[2] + av[3]);
#pragma unroll
      for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[e], bv[e], acc, 0, 0, 0);
    }
  } else {
    for (int k0 = kbeg; k0 < kend; k0 += 32) {                // eight loads per operand in flight
      float av[8], bv[8];
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const int k = k0 + 4 * c + q;
        const bool k_ok = k < kend;
        av[c] = (a_ok && k_ok) ? ap[(long)k * p.sa_k] : 0.f;
        bv[c] = (b_ok && k_ok) ? bp[(long)k * p.sb_k] : 0.f;
      }
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        asum += av[c];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[c], bv[c], acc, 0, 0, 0);
      }
    }
  }
  asum += __shfl_xor(asum, 16, 64);
  asum += __shfl_xor(asum, 32, 64);
  if (wave > 0) {
    *reinterpret_cast<v4f*>(&red[wave - 1][lane][0]) = acc;
    if (lane < 16) reds[wave - 1][lane] = asum;
  }
  __syncthreads();
  if (wave > 0) return;
#pragma unroll
  for (int w = 0; w < 3; ++w) {
    const v4f o = *reinterpret_cast<const v4f*>(&red[w][lane][0]);
    acc += o;
    if (lane < 16) asum += reds[w][lane];
  }
  if (p.rowsum != nullptr && tn == 0 && q == 0 && a_ok) atomicAdd(p.rowsum + am, asum);   // rowsum[m] += sum_k A[m, k]
  const int n = tn * 16 + i;
  if (n >= p.N) return;
  const float bias = p.bias != nullptr ? p.bias[n] : 0.f;
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int m = tm * 16 + 4 * q + r;
    if (m >= p.M) continue;
    float v = acc[r] + bias;
    if (p.relu) v = fmaxf(v, 0.f);
    float* o = p.out + (long)m * p.ldo + n;
    if (p.accumulate) atomicAdd(o, v); else *o = v;
  }
}

__global__ __launch_bounds__(256) void rows_add_f32_kernel(bf16_t* __restrict__ dst, const int32_t* __restrict__ idx,
                                                           const float* __restrict__ src, int n, int H) {
  const int total = n * H;
  for (int e = blockIdx.x * 256 + threadIdx.x; e < total; e += gridDim.x * 256) {
    const int r = e / H, c = e - r * H;
    bf16_t* d = dst + (size_t)idx[r] * H + c;
    *d = f2bf(bf2f(*d) + src[e]);
  }
}

}  // namespace

extern "C" int unimm_linear_f32(const unimm_linear_f32_args* a, void* stream) {
  if (a == nullptr || a->a == nullptr || a->b == nullptr || a->out == nullptr) return UNIMM_E_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return UNIMM_E_SHAPE;
  if (a->ldo < a->N) return UNIMM_E_ALIGN;
  LinF32 p;
  p.a = a->a; p.b = a->b; p.bias = a->bias; p.out = a->out; p.rowsum = a->rowsum;
  p.M = a->M; p.N = a->N; p.K = a->K;
  p.sa_m = a->sa_m; p.sa_k = a->sa_k; p.sb_k = a->sb_k; p.sb_n = a->sb_n; p.ldo = a->ldo;
  p.relu = a->relu; p.accumulate = a->accumulate;
  const long tiles = (long)((p.M + 15) / 16) * ((p.N + 15) / 16);
  hipLaunchKernelGGL(linear_f32_kernel, dim3((unsigned)tiles), dim3(256), 0, (hipStream_t)stream, p);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_rows_add_f32(void* dst, const int32_t* idx, const float* src, int32_t n, int32_t H, void* stream) {
  if (!dst || !idx || !src || n <= 0 || H <= 0) return UNIMM_E_ARG;
  int blocks = (int)(((long)n * H + 255) / 256);
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(rows_add_f32_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (bf16_t*)dst, idx, src, n, H);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}
