// unimm_gemm_nt, tile configuration 12 (Cfg<2, 4, 6, 64, 2, 0, 3>: 192x256, X operand on a three-slot ring): see gemm_nt.h.
#include "gemm_nt.h"
int unimm_nt_launch_cfg12(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk) {
  return nt_launch_epi<Cfg<2, 4, 6, 64, 2, 0, 3>>(p, epi, out_f32, want_persist, s, sk);
}
