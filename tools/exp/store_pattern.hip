// Does the shape of a wave's 16-byte stores matter for fp32 output rows?  (stand-alone probe, not part of the product)
//   hipcc --offload-arch=gfx950 -O2 tools/exp/store_pattern.hip -o tools/exp/_bin/store_pattern
// A GEMM epilogue lane owns 8 consecutive fp32 columns of a row (32 B) and writes them with TWO 16-byte stores, so one store
// instruction covers 8 lanes x 16 B with 16-B gaps (half of each 128-B line); the alternative gives a lane columns 4l..4l+3 and
// 32+4l..: each instruction then writes whole 128-B lines.  Same bytes, [M, 768] fp32, non-temporal stores.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((ext_vector_type(4))) float f32x4;

template <int MODE>
__global__ __launch_bounds__(256) void wr(float* out, int M, int N) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  const int nwaves = gridDim.x * 4;
  const int tiles_n = N / 64;
  // a wave writes [8 rows][64 cols] per iteration (lane >> 3 = row, lane & 7 = column group)
  for (long t = wave; t < (long)(M / 8) * tiles_n; t += nwaves) {
    const int rb = (int)(t / tiles_n) * 8, cb = (int)(t % tiles_n) * 64;
    float* row = out + (size_t)(rb + (lane >> 3)) * N + cb;
    const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
    if (MODE == 0) {          // lane owns 8 consecutive columns: two stores 16 B apart
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(row + (lane & 7) * 8));
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(row + (lane & 7) * 8 + 4));
    } else if (MODE == 1) {   // lane owns columns 4l..4l+3 and 32+4l..: whole lines per instruction
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(row + (lane & 7) * 4));
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(row + 32 + (lane & 7) * 4));
    } else if (MODE == 2) {   // as 0, default cache policy
      *reinterpret_cast<f32x4*>(row + (lane & 7) * 8) = v;
      *reinterpret_cast<f32x4*>(row + (lane & 7) * 8 + 4) = v;
    } else {                  // as 1, default cache policy
      *reinterpret_cast<f32x4*>(row + (lane & 7) * 4) = v;
      *reinterpret_cast<f32x4*>(row + 32 + (lane & 7) * 4) = v;
    }
  }
}

template <int MODE> float run(float* d, int M, int N, int grid) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL(wr<MODE>, dim3(grid), dim3(256), 0, 0, d, M, N);
  hipEventRecord(a);
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(wr<MODE>, dim3(grid), dim3(256), 0, 0, d, M, N);
  hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b);
  return ms / 20.f;
}

int main() {
  const int M = 31168, N = 768;
  float* d; hipMalloc(&d, (size_t)M * N * 4);
  for (int grid : {256, 512, 2048}) {
    const float t0 = run<0>(d, M, N, grid), t1 = run<1>(d, M, N, grid), t2 = run<2>(d, M, N, grid), t3 = run<3>(d, M, N, grid);
    const double gb = (double)M * N * 4 / 1e9;
    printf("grid %4d: 8-consecutive-columns nt %6.1f us (%5.2f TB/s) | whole-line nt %6.1f us (%5.2f TB/s) | 8-consecutive default %6.1f us (%5.2f) | whole-line default %6.1f us (%5.2f)\n",
           grid, t0 * 1e3, gb / t0, t1 * 1e3, gb / t1, t2 * 1e3, gb / t2, t3 * 1e3, gb / t3);
  }
  return 0;
}
