// unimm_gemm_nt, tile configuration 15 (Cfg<2, 2, 2, 64, 2, 0, 3>: 64x128, X operand on a three-slot ring: 56 KiB, two workgroups per CU): see gemm_nt.h.
#include "gemm_nt.h"
int unimm_nt_launch_cfg15(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk) {
  return nt_launch_epi<Cfg<2, 2, 2, 64, 2, 0, 3>>(p, epi, out_f32, want_persist, s, sk);
}
