// Stand-alone timing of one GEMM shape through the C ABI (not part of the product).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude tools/exp/exp_gemm.hip -o gpurun_out/exp_gemm
// The ablation switches of rounds 1-2 (-DUNIMM_EXP=k: no staging, no epilogue, shared tiles, ... ; DESIGN.md 5 / 5b list
// their results) lived inside csrc/gemm.hip up to commit 25f86ac and were removed from the product source in round 3.
#define UNIMM_EXP 0
#include "../../unimm_amd/csrc/gemm.hip"
#include <cstdio>
#include <cstdlib>
#include <vector>

int main(int argc, char** argv) {
  const int M = argc > 1 ? atoi(argv[1]) : 31162, N = argc > 2 ? atoi(argv[2]) : 3072, K = argc > 3 ? atoi(argv[3]) : 768;
  const int epi = argc > 4 ? atoi(argv[4]) : 0, cfg = argc > 5 ? atoi(argv[5]) : 0;
  const bool f32 = epi == UNIMM_EPI_BIAS_DROP_RESID;
  void *x, *w, *bias, *aux, *out, *out2;
  hipMalloc(&x, (size_t)M * K * 2); hipMalloc(&w, (size_t)N * K * 2); hipMalloc(&bias, N * 4);
  hipMalloc(&aux, (size_t)M * N * 4); hipMalloc(&out, (size_t)M * N * 4); hipMalloc(&out2, (size_t)M * N * 2);
  std::vector<uint16_t> h((size_t)(M > N ? M : N) * K);
  // random data of both signs over several binades (zero / sign-constant operands run at a higher clock and flatter the kernel)
  for (size_t i = 0; i < h.size(); ++i) h[i] = (uint16_t)(((rand() & 1) << 15) | ((0x7b + (rand() % 6)) << 7) | (rand() & 0x7f));
  hipMemcpy(x, h.data(), (size_t)M * K * 2, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), (size_t)N * K * 2, hipMemcpyHostToDevice);
  hipMemset(bias, 0, N * 4);
  { std::vector<uint16_t> ha((size_t)M * N * 2); for (size_t i = 0; i < ha.size(); ++i) ha[i] = (uint16_t)(0x3f00 + (rand() & 0xff)); hipMemcpy(aux, ha.data(), ha.size() * 2, hipMemcpyHostToDevice); }
  unimm_gemm_nt_args a{};
  a.x = x; a.w = w; a.bias = (const float*)bias; a.aux = aux; a.out = out; a.out2 = epi == UNIMM_EPI_BIAS_GELU_DG ? out2 : nullptr;
  a.M = M; a.N = N; a.K = K; a.ldx = K; a.ldw = K; a.ldaux = N; a.ldo = N; a.epilogue = epi; a.out_f32 = f32;
  a.drop_thr = 0; a.drop_scale = 1.f;
  a.tile = cfg < 0 ? 0 : cfg;
  if (cfg < 0) {   // TN: dw[N,K] += x1[M,N]^T x2[M,K]  (x reused as dy when N <= K, sizes are what matter)
    void *dy, *dw, *db;
    hipMalloc(&dy, (size_t)M * N * 2); hipMalloc(&dw, (size_t)N * K * 4); hipMalloc(&db, N * 4);
    hipMemset(dy, 0x3c, (size_t)M * N * 2); hipMemset(dw, 0, (size_t)N * K * 4); hipMemset(db, 0, N * 4);
    unimm_gemm_tn_args g{};
    g.dy = dy; g.x = x; g.dw = (float*)dw; g.dbias = cfg == -2 ? (float*)db : nullptr;
    g.M = M; g.N = N; g.K = K; g.lddy = N; g.ldx = K; g.lddw = K;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) if (unimm_gemm_tn(&g, nullptr) != 0) { printf("launch failed\n"); return 1; }
    hipDeviceSynchronize();
    hipEventRecord(e0, nullptr);
    const int it = 200;
    for (int i = 0; i < it; ++i) unimm_gemm_tn(&g, nullptr);
    hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / it;
    printf("EXP=%d TN M=%d N=%d K=%d dbias=%d: %.1f us  %.1f TFLOP/s-equivalent\n", UNIMM_EXP, M, N, K, cfg == -2, us, 2.0 * M * N * K / us / 1e6);
    return 0;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) { int rc = unimm_gemm_nt(&a, nullptr); if (rc != 0) { printf("launch failed rc=%d hip=%s\n", rc, hipGetErrorString(hipGetLastError())); return 1; } }
  hipDeviceSynchronize();
  hipEventRecord(e0, nullptr);
  const int it = 200;
  for (int i = 0; i < it; ++i) unimm_gemm_nt(&a, nullptr);
  hipEventRecord(e1, nullptr); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / it;
  printf("EXP=%d M=%d N=%d K=%d epi=%d cfg=%d: %.1f us  %.1f TFLOP/s-equivalent\n", UNIMM_EXP, M, N, K, epi, cfg, us, 2.0 * M * N * K / us / 1e6);
  return 0;
}
