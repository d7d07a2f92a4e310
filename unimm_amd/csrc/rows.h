// Row helpers shared by the row kernels (rowops.hip) and the fp32x3-mode kernels (x3ops.hip).
#pragma once
#include "common.h"

namespace {

constexpr int MAXC = 2;          // 8-element chunks per lane: hidden sizes up to 1024
#ifndef UNIMM_RED_BLOCKS
#define UNIMM_RED_BLOCKS 512
#endif
constexpr int RED_BLOCKS = UNIMM_RED_BLOCKS;  // row-kernel grid for kernels that emit per-block column partials

// ------------------------------------------------------------------------------------------------
// row helpers: a wave owns one row of H elements as 8-element (16-byte) chunks, chunk c = lane + 64*i
// ------------------------------------------------------------------------------------------------
struct Row8 { float v[MAXC][8]; };

__device__ __forceinline__ void load_row_bf16(const bf16_t* __restrict__ p, int H, int lane, Row8& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c * 8 < H) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(p + c * 8);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        r.v[i][2 * j] = __uint_as_float(raw[j] << 16);
        r.v[i][2 * j + 1] = __uint_as_float(raw[j] & 0xffff0000u);
      }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
    }
  }
}

__device__ __forceinline__ void store_row_bf16(bf16_t* __restrict__ p, int H, int lane, const Row8& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c * 8 < H)
      *reinterpret_cast<u32x4*>(p + c * 8) = u32x4{pack2bf(r.v[i][0], r.v[i][1]), pack2bf(r.v[i][2], r.v[i][3]),
                                                   pack2bf(r.v[i][4], r.v[i][5]), pack2bf(r.v[i][6], r.v[i][7])};
  }
}

__device__ __forceinline__ void store_row_f32(float* __restrict__ p, int H, int lane, const Row8& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c * 8 < H) {
      *reinterpret_cast<f32x4*>(p + c * 8) = f32x4{r.v[i][0], r.v[i][1], r.v[i][2], r.v[i][3]};
      *reinterpret_cast<f32x4*>(p + c * 8 + 4) = f32x4{r.v[i][4], r.v[i][5], r.v[i][6], r.v[i][7]};
    }
  }
}

__device__ __forceinline__ void load_vec_f32(const float* __restrict__ p, int H, int lane, Row8& r) {
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = lane + 64 * i;
    if (c * 8 < H) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(p + c * 8);
      const f32x4 b = *reinterpret_cast<const f32x4*>(p + c * 8 + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { r.v[i][j] = a[j]; r.v[i][4 + j] = b[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) r.v[i][j] = 0.f;
    }
  }
}

__device__ __forceinline__ void row_stats(const Row8& x, int H, float& mean, float& rstd, float eps) {
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) s += x.v[i][j];
  mean = wave_sum(s) / (float)H;
  float q = 0.f;
#pragma unroll
  for (int i = 0; i < MAXC; ++i) {
    const int c = (threadIdx.x & 63) + 64 * i;
    if (c * 8 < H) {
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = x.v[i][j] - mean; q += d * d; }
    }
  }
  rstd = rsqrtf(wave_sum(q) / (float)H + eps);
}

}  // namespace
