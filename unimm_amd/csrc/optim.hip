// Fused AdamW over the flat fp32 parameter / gradient arenas.
//
// Replaces `pytorch_transformers.AdamW(optimizer_grouped_parameters, lr=...).step()` of the reference
// (train.py:322-347, :458): one group per parameter there, with lr in {lr, image_lr} (config/
// language_weights.json) and weight_decay in {0.01, 0} (bias / LayerNorm).  Published algorithm of that
// class (the library itself is not in this image):
//     m = b1 m + (1-b1) g ;  v = b2 v + (1-b2) g^2 ;  p -= lr sqrt(1-b2^t)/(1-b1^t) * m / (sqrt(v) + eps) ;
//     p -= lr wd p            (decoupled decay, applied AFTER the Adam update, on the updated value)
// Here the arena is walked once: 16 B per lane, the (lr, wd) group of a 64-element chunk from a byte
// table (arena views are 64-element aligned), the bf16 GEMM-operand copy of the weights written in the
// same pass.  HBM-bound: 4 x 4 B read + 3 x 4 B + 2 B written per parameter.
#include "common.h"

namespace {

struct AdamWParams {
  float* p; const float* g; float* m; float* v; bf16_t* w16; const uint8_t* group;
  size_t n;                    // elements (multiple of 4)
  float step_size[UNIMM_ADAMW_MAX_GROUPS];   // lr_g * bias correction
  float decay[UNIMM_ADAMW_MAX_GROUPS];       // lr_g * wd_g
  float beta1, beta2, c1, c2, eps, grad_scale;   // c = 1 - beta, formed in double on the host
  int zero_grad;
};

__device__ __forceinline__ float rnd(float x) { asm volatile("" : "+v"(x)); return x; }

__global__ __launch_bounds__(256) void adamw_kernel(AdamWParams a) {
  __shared__ float s_step[UNIMM_ADAMW_MAX_GROUPS], s_decay[UNIMM_ADAMW_MAX_GROUPS];
#pragma unroll
  for (int i = 0; i < UNIMM_ADAMW_MAX_GROUPS; ++i)
    if (threadIdx.x == i) { s_step[i] = a.step_size[i]; s_decay[i] = a.decay[i]; }
  __syncthreads();
  const size_t nvec = a.n >> 2;
  const float b1 = a.beta1, b2 = a.beta2, c1 = a.c1, c2 = a.c2;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += (size_t)gridDim.x * blockDim.x) {
    const uint32_t gid = a.group[i >> 4];                    // 16 lanes x 4 elements = one 64-element chunk
    if (gid >= UNIMM_ADAMW_MAX_GROUPS) continue;             // parameters that never receive a gradient
    const f32x4 g4 = reinterpret_cast<const f32x4*>(a.g)[i];
    f32x4 p4 = reinterpret_cast<f32x4*>(a.p)[i];
    f32x4 m4 = reinterpret_cast<f32x4*>(a.m)[i];
    f32x4 v4 = reinterpret_cast<f32x4*>(a.v)[i];
    const float ss = s_step[gid], dc = s_decay[gid];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      // one rounding per operation, as the reference's separate mul_ / add_ / addcmul_ / addcdiv_ calls have:
      // `rnd` is an empty asm that pins each product in a register so it cannot be contracted into an FMA
      const float g = rnd(g4[e] * a.grad_scale);
      const float m = rnd(m4[e] * b1) + rnd(c1 * g);
      const float v = rnd(v4[e] * b2) + rnd(rnd(c2 * g) * g);
      const float denom = __fsqrt_rn(v) + a.eps;
      float p = p4[e] + rnd(-ss * rnd(__fdiv_rn(m, denom)));
      if (dc != 0.0f) p = p + rnd(-dc * p);
      m4[e] = m; v4[e] = v; p4[e] = p;
    }
    reinterpret_cast<f32x4*>(a.p)[i] = p4;
    reinterpret_cast<f32x4*>(a.m)[i] = m4;
    reinterpret_cast<f32x4*>(a.v)[i] = v4;
    if (a.w16 != nullptr) reinterpret_cast<u32x2*>(a.w16)[i] = u32x2{pack2bf(p4[0], p4[1]), pack2bf(p4[2], p4[3])};
    if (a.zero_grad) reinterpret_cast<f32x4*>(const_cast<float*>(a.g))[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}

}  // namespace

extern "C" int unimm_adamw_step(const unimm_adamw_args* a, void* stream) {
  if (a == nullptr || a->p == nullptr || a->g == nullptr || a->m == nullptr || a->v == nullptr || a->group == nullptr)
    return UNIMM_E_ARG;
  if (a->n <= 0 || (a->n % 64) != 0) return UNIMM_E_SHAPE;
  if (((uintptr_t)a->p | (uintptr_t)a->g | (uintptr_t)a->m | (uintptr_t)a->v) & 15) return UNIMM_E_ALIGN;
  if (a->w16 != nullptr && ((uintptr_t)a->w16 & 7)) return UNIMM_E_ALIGN;
  if (a->step < 1 || a->n_groups < 1 || a->n_groups > UNIMM_ADAMW_MAX_GROUPS) return UNIMM_E_ARG;
  AdamWParams p;
  p.p = a->p; p.g = a->g; p.m = a->m; p.v = a->v; p.w16 = (bf16_t*)a->w16; p.group = a->group;
  p.n = (size_t)a->n;
  // bias correction in double on the host, as the reference computes it in Python floats
  double corr = 1.0;
  if (a->correct_bias) corr = sqrt(1.0 - pow(a->beta2, (double)a->step)) / (1.0 - pow(a->beta1, (double)a->step));
  for (int i = 0; i < UNIMM_ADAMW_MAX_GROUPS; ++i) {
    const bool on = i < a->n_groups;
    p.step_size[i] = on ? (float)(a->lr[i] * corr) : 0.f;
    p.decay[i] = on ? (float)(a->lr[i] * a->weight_decay[i]) : 0.f;
  }
  p.beta1 = (float)a->beta1; p.beta2 = (float)a->beta2; p.c1 = (float)(1.0 - a->beta1); p.c2 = (float)(1.0 - a->beta2);
  p.eps = (float)a->eps; p.grad_scale = a->grad_scale;
  p.zero_grad = a->zero_grad;
  const size_t nvec = p.n >> 2;
  size_t blocks = (nvec + 255) / 256;
  if (blocks > 256 * 16) blocks = 256 * 16;       // grid-stride: 16 workgroups of 4 waves per CU
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, p);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}
