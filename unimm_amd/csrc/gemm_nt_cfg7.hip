// unimm_gemm_nt, tile configuration 7 (Cfg<2, 2, 2, 64, 2>): see gemm_nt.h (device code) and gemm.hip (tile choice).
#include "gemm_nt.h"
int unimm_nt_launch_cfg7(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk) {
  return nt_launch_epi<Cfg<2, 2, 2, 64, 2>>(p, epi, out_f32, want_persist, s, sk);
}
