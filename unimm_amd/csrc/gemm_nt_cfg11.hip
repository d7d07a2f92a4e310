// unimm_gemm_nt, tile configuration 11 (Cfg<1, 4, 8, 32, 3>): see gemm_nt.h (device code) and gemm.hip (tile choice).
#include "gemm_nt.h"
int unimm_nt_launch_cfg11(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk) {
  return nt_launch_epi<Cfg<1, 4, 8, 32, 3>>(p, epi, out_f32, want_persist, s, sk);
}
