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
    __builtin_amdgcn_s_barrier();                    // the previous tile's fragment reads are over
    stage(p, n0, m0, 0, smem, wave, lane);
    for (int t = 0; t < nk; ++t) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < nk) stage(p, n0, m0, (t + 1) * 64, smem + ((t + 1) & 1) * 65536, wave, lane);
      const uint32_t so = (uint32_t)((t & 1) * 65536);
      bf16x8 fw[2][8], fx[2][8];
#define RD(buf, ks)                                                                                                  \
      {                                                                                                              \
        fw[buf][0] = lds_read_b128<0 * 2048>(aw0[ks] + so); fw[buf][1] = lds_read_b128<1 * 2048>(aw0[ks] + so);      \
        fw[buf][2] = lds_read_b128<2 * 2048>(aw0[ks] + so); fw[buf][3] = lds_read_b128<3 * 2048>(aw0[ks] + so);      \
        fw[buf][4] = lds_read_b128<4 * 2048>(aw0[ks] + so); fw[buf][5] = lds_read_b128<5 * 2048>(aw0[ks] + so);      \
        fw[buf][6] = lds_read_b128<6 * 2048>(aw0[ks] + so); fw[buf][7] = lds_read_b128<7 * 2048>(aw0[ks] + so);      \
        fx[buf][0] = lds_read_b128<0 * 2048>(ax0[ks] + so); fx[buf][1] = lds_read_b128<1 * 2048>(ax0[ks] + so);      \
        fx[buf][2] = lds_read_b128<2 * 2048>(ax0[ks] + so); fx[buf][3] = lds_read_b128<3 * 2048>(ax0[ks] + so);      \
        fx[buf][4] = lds_read_b128<4 * 2048>(ax0[ks] + so); fx[buf][5] = lds_read_b128<5 * 2048>(ax0[ks] + so);      \
        fx[buf][6] = lds_read_b128<6 * 2048>(ax0[ks] + so); fx[buf][7] = lds_read_b128<7 * 2048>(ax0[ks] + so);      \
      }
#define MM(buf)                                                                                                      \
      _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                                  \
        _Pragma("unroll") for (int i = 0; i < 8; ++i)                                                                \
          asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(fw[buf][i]), "v"(fx[buf][j]));
      RD(0, 0)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      RD(1, 1)                                       // second 32-deep sub-step's fragments fly under the first one's 64 MFMAs
      MM(0)
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      MM(1)
#undef RD
#undef MM
    }
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

static float bf2f(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 31162, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  void *x, *w, *out;
  hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&out, (size_t)M * N * 4);
  std::vector<uint16_t> hx((size_t)M * K), hw((size_t)N * K);
  for (auto& v : hx) v = (uint16_t)(((rand() & 1) << 15) | ((0x7b + (rand() % 6)) << 7) | (rand() & 0x7f));
  for (auto& v : hw) v = (uint16_t)(((rand() & 1) << 15) | ((0x7b + (rand() % 6)) << 7) | (rand() & 0x7f));
  hipMemcpy(x, hx.data(), hx.size() * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice);
  hipFuncSetAttribute((const void*)agpr_gemm, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  P p{(const bf16_t*)x, (const bf16_t*)w, (float*)out, M, N, K, 1};
  const int ntiles = ((M + 255) / 256) * ((N + 255) / 256);
  hipMemset(out, 0, (size_t)M * N * 4);
  hipLaunchKernelGGL(agpr_gemm, dim3(256), dim3(256), 131072, 0, p, ntiles);
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
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(agpr_gemm, dim3(256), dim3(256), 131072, 0, p, ntiles);
  hipDeviceSynchronize();
  hipEventRecord(e0, 0);
  const int it = 100;
  for (int i = 0; i < it; ++i) hipLaunchKernelGGL(agpr_gemm, dim3(256), dim3(256), 131072, 0, p, ntiles);
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
  printf("agpr 4-wave 128x128 wave tiles, loop only: M=%d N=%d K=%d: %.1f us  %.1f TFLOP/s-equivalent (%d tiles, %.2f us per 256x256x64 step per workgroup)\n",
         M, N, K, us, 2.0 * M * N * K / us / 1e6, ntiles, us / ((ntiles + 255) / 256) / (K / 64));
  return 0;
}
