// What does an out-of-range buffer_load ... lds write into LDS on gfx950?  (stand-alone probe, not part of the product)
//   hipcc --offload-arch=gfx950 -O2 tools/exp/oob_lds_dma.hip -o tools/exp/_bin/oob_lds_dma
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

__global__ void probe(const uint32_t* src, uint32_t* out, int num_bytes, int soff) {
  __shared__ __attribute__((aligned(16))) uint32_t lds[2 * 256];
  for (int i = threadIdx.x; i < 512; i += 64) lds[i] = 0xAAAAAAAAu;
  __syncthreads();
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)(uintptr_t)src);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)((uintptr_t)src >> 32));
  typedef __attribute__((ext_vector_type(4))) uint32_t u4;
  u4 srd = {lo, hi & 0xffffu, (uint32_t)num_bytes, 0x00020000u};
  const uint32_t voff = threadIdx.x * 16;
  const uint32_t ldsbase = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) void*)lds;
  // form 1: voffset only
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)"
               :: "s"(ldsbase), "v"(voff), "s"(srd) : "memory", "m0");
  // form 2: soffset carries part of the offset (is it range-checked?)
  const uint32_t so = __builtin_amdgcn_readfirstlane((uint32_t)soff);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)"
               :: "s"(ldsbase + 1024), "v"(voff), "s"(srd), "s"(so) : "memory", "m0");
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 64) out[i] = lds[i];
}

int main() {
  std::vector<uint32_t> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = 0x1000 + i;
  uint32_t *d, *o;
  hipMalloc(&d, 4096 * 4); hipMalloc(&o, 512 * 4);
  hipMemcpy(d, h.data(), 4096 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d, o, 1000, 512);
  std::vector<uint32_t> r(512);
  hipMemcpy(r.data(), o, 512 * 4, hipMemcpyDeviceToHost);
  printf("form 1 (num_records = 1000 bytes, voffset = lane * 16): dwords 244..255\n");
  for (int i = 244; i < 256; ++i) printf("  lds[%d] = %08x (in-range value would be %08x)\n", i, r[i], 0x1000 + i);
  printf("form 2 (soffset = 512): lane l reads byte 512 + 16 l; dwords 116..127 (byte 976..1023 -> source 1488..)\n");
  for (int i = 116; i < 128; ++i) printf("  lds2[%d] = %08x (in-range value would be %08x)\n", i, r[256 + i], 0x1000 + 128 + i);
  return 0;
}
