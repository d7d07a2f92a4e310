// Prototype (not part of the product): 256x256x64 NT GEMM main loop with FOUR waves of 128x128 wave tiles whose 256
// accumulator registers live in AGPRs (inline-asm MFMA with "+a" operands), i.e. the wave tiling hipBLASLt's
// MT256x256x64 kernels use: 16 fragment reads per 64 MFMAs instead of the product's 12 per 32.  Answers one question:
// what does that main loop reach on gfx950 when the compiler cannot shuffle the accumulators?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/exp/agpr_gemm.hip -o gpurun_out/agpr_gemm && gpurun_out/agpr_gemm [M N K]
// Prints the time of the loop-only kernel (one guarded store keeps the accumulators live) and checks a build of the same
// loop that stores its tile against a host reference on sampled elements.
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <cstring>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef uint16_t bf16_t;

#define GLB_PTR(p) ((__attribute__((address_space(1))) void*)(p))
#define LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ int kswz(int row) { return (row >> 1) & 7; }

template <int OFF> __device__ __forceinline__ bf16x8 lds_read_b128(uint32_t addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}

struct P { const bf16_t* x; const bf16_t* w; float* out; int M, N, K; int store; };

// one K-step of the [W 256 rows | X 256 rows] x 64 bf16 tile: 64 LDS-DMA wave-instructions, 16 per wave
__device__ __forceinline__ void stage(const P& p, int n0, int m0, int k0, char* slot, int wave, int lane) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int inst = r * 4 + wave;
    const int rr = inst * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ kswz(rr);
    const bf16_t* src;
    if (rr < 256) { int g = n0 + rr; g = g < p.N ? g : p.N - 1; src = p.w + (size_t)g * p.K + k0 + chunk * 8; }
    else { int g = m0 + rr - 256; g = g < p.M ? g : p.M - 1; src = p.x + (size_t)g * p.K + k0 + chunk * 8; }
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(slot + inst * 1024), 16, 0, 0);
  }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void agpr_gemm(P p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + 255) / 256;
  const int nk = p.K / 64;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)LDS_PTR(smem);
  uint32_t aw0[2], ax0[2];
  {
    const int rw = wn * 128 + (lane & 15), rx = wm * 128 + (lane & 15), cq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      aw0[ks] = lds0 + rw * 128 + (((ks * 4 + cq) ^ kswz(rw)) << 4);
      ax0[ks] = lds0 + (256 + rx) * 128 + (((ks * 4 + cq) ^ kswz(rx)) << 4);
    }
  }
  for (int lt = blockIdx.x; lt < ntiles; lt += gridDim.x) {
    const int tm = lt / nbn, tn = lt - tm * nbn;
    const int m0 = tm * 256, n0 = tn * 256;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // this wave's 16 LDS-DMA pieces of a K-step: per-lane byte offsets (constant over K) on scalar bases
    uint32_t so[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int inst = r * 4 + wave;
      const int rr = inst * 8 + (lane >> 3);
      const int chunk = (lane & 7) ^ kswz(rr);
      int g = rr < 256 ? n0 + rr : m0 + rr - 256;
      const int lim = rr < 256 ? p.N : p.M;
      g = g < lim ? g : lim - 1;
      so[r] = (uint32_t)g * (uint32_t)(p.K * 2) + chunk * 16;
    }
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
#define DMA(r, kt)                                                                                                   \
    {                                                                                                                \
      const char* base_ = (const char*)((r) < 8 ? p.w : p.x) + (size_t)(kt) * 128;   /* pieces 0-7 of a wave are W rows */                \
      const uint32_t dst_ = ldsw + ((kt) & 1) * 65536 + (r) * 4096;                                                  \
      uint32_t keep_;                                                                                                \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep_) : "v"(so[r]), "s"(dst_), "s"(base_) : "memory");                                   \
    }
    __builtin_amdgcn_s_barrier();                    // the previous tile's fragment reads are over
#pragma unroll
    for (int r = 0; r < 16; ++r) DMA(r, 0)
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t so_l = (uint32_t)((t & 1) * 65536);
      const bool more = t + 1 < nk;
      bf16x8 fw[2][8], fx[2][8];
#define RDW(buf, ks, i) fw[buf][i] = lds_read_b128<(i) * 2048>(aw0[ks] + so_l);
#define RDX(buf, ks, j) fx[buf][j] = lds_read_b128<(j) * 2048>(ax0[ks] + so_l);
      RDW(0, 0, 0) RDW(0, 0, 1) RDW(0, 0, 2) RDW(0, 0, 3) RDW(0, 0, 4) RDW(0, 0, 5) RDW(0, 0, 6) RDW(0, 0, 7)
      RDX(0, 0, 0) RDX(0, 0, 1) RDX(0, 0, 2) RDX(0, 0, 3) RDX(0, 0, 4) RDX(0, 0, 5) RDX(0, 0, 6) RDX(0, 0, 7)
      // sub-step 0: unit j = X fragment j against the eight W fragments; behind it two fragment reads of sub-step 1 and
      // two LDS-DMA pieces of the next K-step
#define UNIT0(j)                                                                                                     \
      RDW(1, 1, j) RDX(1, 1, j)                                                                                      \
      if (more) { DMA(2 * (j), t + 1) DMA(2 * (j) + 1, t + 1) }                                                      \
      if ((j) == 0) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");                                               \
      __builtin_amdgcn_s_setprio(1);                                                                                 \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                  \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[0][i]), "v"(fx[0][j]));    \
      __builtin_amdgcn_s_setprio(0);                                                                                 \
      __builtin_amdgcn_sched_barrier(0);
      UNIT0(0) UNIT0(1) UNIT0(2) UNIT0(3) UNIT0(4) UNIT0(5) UNIT0(6) UNIT0(7)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
      for (int j = 0; j < 8; ++j) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[1][i]), "v"(fx[1][j]));
      }
#undef UNIT0
#undef RDW
#undef RDX
    }
#undef DMA
    if (p.store) {                                   // accumulator layout: lane = row m, 4 consecutive n per register quad
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = m0 + wm * 128 + j * 16 + (lane & 15), n = n0 + wn * 128 + i * 16 + 4 * (lane >> 4);
          if (m < p.M && n + 3 < p.N) *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.N + n) = acc[i][j];
        }
    } else {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sacc == 12345.678f) p.out[0] = sacc;
    }
  }
}

// ---- the same wave tiling on a 4-slot ring of BK = 32 stages (32 KiB each): three stages in flight across the barrier
__device__ __forceinline__ int kswz32(int row) { return (row >> 2) & 3; }

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))) void agpr_gemm_deep(P p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = (p.N + 255) / 256;
  const int nk = p.K / 32;
  const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)LDS_PTR(smem);
  uint32_t aw0, ax0;
  {
    const int rw = wn * 128 + (lane & 15), rx = wm * 128 + (lane & 15), cq = lane >> 4;
    aw0 = lds0 + rw * 64 + ((cq ^ kswz32(rw)) << 4);
    ax0 = lds0 + (256 + rx) * 64 + ((cq ^ kswz32(rx)) << 4);
  }
  for (int lt = blockIdx.x; lt < ntiles; lt += gridDim.x) {
    const int tm = lt / nbn, tn = lt - tm * nbn;
    const int m0 = tm * 256, n0 = tn * 256;
    f32x4 acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    uint32_t so[8];                                   // this wave's 8 LDS-DMA pieces of a stage (16 rows x 64 B each)
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      const int inst = r * 4 + wave;
      const int rr = inst * 16 + (lane >> 2);
      const int chunk = (lane & 3) ^ kswz32(rr);
      int g = rr < 256 ? n0 + rr : m0 + rr - 256;
      const int lim = rr < 256 ? p.N : p.M;
      g = g < lim ? g : lim - 1;
      so[r] = (uint32_t)g * (uint32_t)(p.K * 2) + chunk * 16;
    }
    const uint32_t ldsw = __builtin_amdgcn_readfirstlane(lds0 + wave * 1024);
#define DMA(r, kt)                                                                                                   \
    {                                                                                                                \
      const char* base_ = (const char*)((r) < 4 ? p.w : p.x) + (size_t)(kt) * 64;     /* pieces 0-3 of a wave are W rows */                \
      const uint32_t dst_ = ldsw + ((kt) & 3) * 32768 + (r) * 4096;                                                  \
      uint32_t keep_;                                                                                                \
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0" \
                   : "=&s"(keep_) : "v"(so[r]), "s"(dst_), "s"(base_) : "memory");                                   \
    }
    __builtin_amdgcn_s_barrier();                    // the previous tile's fragment reads are over
#pragma unroll
    for (int st = 0; st < 3; ++st)
      if (st < nk) {
#pragma unroll
        for (int r = 0; r < 8; ++r) DMA(r, st)
      }
    for (int t = 0; t < nk; ++t) {
      // stage t has landed once at most the two younger stages (8 pieces each) are outstanding
      const int rem = nk - 1 - t;
      if (rem >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else if (rem == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      const uint32_t so_l = (uint32_t)((t & 3) * 32768);
      const bool more = t + 3 < nk;
      bf16x8 fw[8], fx[8];
#define RDW(i) fw[i] = lds_read_b128<(i) * 1024>(aw0 + so_l);
#define RDX(j) fx[j] = lds_read_b128<(j) * 1024>(ax0 + so_l);
      RDW(0) RDW(1) RDW(2) RDW(3) RDW(4) RDW(5) RDW(6) RDW(7) RDX(0) RDX(1)
      RDX(2) RDX(3) RDX(4) RDX(5) RDX(6) RDX(7)
#define UNIT(j, PEND)                                                                                                \
      if (more) DMA(j, t + 3)                                                                                        \
      asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(PEND) : "memory");                                                  \
      __builtin_amdgcn_s_setprio(1);                                                                                 \
      _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                  \
        asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[i]), "v"(fx[j]));          \
      __builtin_amdgcn_s_setprio(0);                                                                                 \
      __builtin_amdgcn_sched_barrier(0);
      UNIT(0, 7) UNIT(1, 6) UNIT(2, 5) UNIT(3, 4) UNIT(4, 3) UNIT(5, 2) UNIT(6, 1) UNIT(7, 0)
#undef UNIT
#undef RDW
#undef RDX
    }
#undef DMA
    if (p.store) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const int m = m0 + wm * 128 + j * 16 + (lane & 15), n = n0 + wn * 128 + i * 16 + 4 * (lane >> 4);
          if (m < p.M && n + 3 < p.N) *reinterpret_cast<f32x4*>(p.out + (size_t)m * p.N + n) = acc[i][j];
        }
    } else {
      float sacc = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) sacc += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
      if (sacc == 12345.678f) p.out[0] = sacc;
    }
  }
}

static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 31162, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const bool deep = argc > 4 && atoi(argv[4]) == 1;          // 1: the 4-slot BK = 32 ring
  auto kern = deep ? agpr_gemm_deep : agpr_gemm;
  void *x, *w, *out;
  hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&out, (size_t)M * N * 4);
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
  for (auto& v : hx) v = (uint16_t)(((rand() & 1) << 15) | ((0x7b + (rand() % 6)) << 7) | (rand() & 0x7f));
  for (auto& v : hw) v = (uint16_t)(((rand() & 1) << 15) | ((0x7b + (rand() % 6)) << 7) | (rand() & 0x7f));
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  P p{(const bf16_t*)x, (const bf16_t*)w, (float*)out, M, N, K, 1};
  const int ntiles = ((M + 255) / 256) * ((N + 255) / 256);
  hipMemset(out, 0, (size_t)M * N * 4);
  hipLaunchKernelGGL(kern, dim3(256), dim3(256), 131072, 0, p, ntiles);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed: %s\n", hipGetErrorString(hipGetLastError())); return 1; }
  std::vector<float> ho((size_t)M * N);
  hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost);
  double worst = 0.0;
  for (int s = 0; s < 2000; ++s) {
    const int m = rand() % M, n = (rand() % (N / 4)) * 4 + (rand() & 3);
    double ref = 0.0;
    for (int k = 0; k < K; ++k) ref += (double)bf2f(hx[(size_t)m * K + k]) * bf2f(hw[(size_t)n * K + k]);
    const double err = fabs(ho[(size_t)m * N + n] - ref) / (1.0 + fabs(ref));
    worst = err > worst ? err : worst;
  }
  printf("check: worst |err| / (1 + |ref|) on 2000 sampled elements = %.2e\n", worst);
  p.store = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 131072, 0, p, ntiles);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 100;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(kern, dim3(256), dim3(256), 131072, 0, p, ntiles);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
  printf("%s", deep ? "[4-slot BK=32 ring] " : "[2-slot BK=64 ring] ");
  printf("agpr 4-wave 128x128 wave tiles, loop only: M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s-equivalent (%d tiles, %.2f us per 256x256x64 step per workgroup)\n",
         M, N, K, us, 2.0 * M * N * K / us / 1e6, ntiles, us / ((ntiles + 255) / 256) / (K / 64));
  return 0;
}
