// bf16 MFMA GEMMs for the UniMM-UL hot path (gfx950).
//
//   gemm_nt : OUT[M,N] = epi( X[M,K] . W[N,K]^T )      forward linears + all dgrads (with W^T copies)
//   gemm_tn : DW[N,K] += DY[M,N]^T . X[M,K]            weight gradients (split over M, fp32 atomics)
//
// Replaces the nn.Linear / matmul chains of the reference (models/vilbert_dialog.py:386-388, 423,
// 453, 466, 515-517, 552, 582, 595, 659-661, 670-672, 745-748, 950, 965, 983, 1002, 1025, 1070,
// 1087, 1488-1489) and their autograd backward.
//
// Design (CDNA4): 128x128x64 block tile, 4 waves (2x2), each wave a 64x64 sub-tile as 4x4
// v_mfma_f32_16x16x32_bf16 accumulators.  Operand tiles go HBM -> LDS by LDS-DMA
// (global_load_lds_dwordx4, 1 KiB per wave-instruction); the LDS image is lane-linear, so the
// bank-conflict swizzle (16-B chunk index ^= (row>>1)&7) is applied on the per-lane SOURCE address
// and again on the ds_read_b128 side.  The MFMA is issued "swapped" (A operand = W rows, B operand
// = X rows) so that each lane ends up with 4 consecutive output columns of one row: 8-byte bf16 /
// 16-byte fp32 stores and vector loads of bias / residual in the epilogue.  Grid order is
// XCD-aware: each XCD walks a contiguous run of tiles with the N index fastest, so an X row panel is
// fetched from HBM once per XCD and W stays L2/MALL resident.
#include <type_traits>
#include "common.h"
#include <stdlib.h>

namespace {

struct GemmNtParams {
  const bf16_t* x; const bf16_t* w; const float* bias; const void* aux;  // aux: fp32 for DROP_RESID, bf16 otherwise
  void* out; bf16_t* out2;
  int M, N, K, ldx, ldw, ldaux, ldo;
  int gn;            // n-tiles per column group of the tile order (see tile_of)
  DropoutArg drop;
  const float* aux_mean; const float* aux_rstd; const float* aux_gamma; const float* aux_beta;  // DROP_RESID: aux = LayerNorm(aux)
  // split-K (ring-loop tiles only; see nt_split_join): ksplit workgroups share one output tile, each reduces a slice of K,
  // partial tiles meet in `slabs` and the last arriver (ticket in `counters`) runs the epilogue.  ksplit <= 1: off.
  int ksplit; float* slabs; int* counters;
};

// Tile order.  Logical ids run group by group over the n-tiles (gn tile columns per group), inside a
// group over the m-tiles, n fastest.  After the XCD remap every XCD walks a contiguous id range, so the
// ~32 workgroups resident on one XCD cover (32/gn) row panels x gn column panels: both operands'
// panels are shared through that XCD's 4 MiB L2.  gn >= the number of n-tiles = plain row-major order.
__device__ __forceinline__ void tile_of(int lid, int nbm, int nbn, int gn, int& tm, int& tn) {
  const int per_group = nbm * gn;
  const int g = lid / per_group;
  const int r = lid - g * per_group;
  const int gw = (nbn - g * gn) < gn ? (nbn - g * gn) : gn;   // width of this (possibly last, narrower) group
  tm = r / gw;
  tn = g * gn + (r - tm * gw);
}

// Block-tile configurations.  A wave owns a (16*MT)(m) x 64(n) output sub-tile (MT x 4 accumulators of
// v_mfma_f32_16x16x32_bf16); WM x WN waves make the block tile; one ring slot holds one K-step of BK.
// The ring is what hides HBM/L2 latency (~3-4k cycles under load): bytes staged per MFMA-cycle halve
// with a 256x256 tile, so the same LDS covers twice the latency of the 128x128 tile.
//   Cfg<2,2,4,64,2>: 128x128, 4 waves, 68 KiB LDS -> 2 workgroups per CU
//   Cfg<2,4,8,64,2>: 256x256, 8 waves, BK=64, 2-slot ring (128 KiB)
//   Cfg<2,4,6,64,2>: 192x256, 8 waves of 96x64, 2-slot ring (112 KiB): the tile for N = 768 at ~31k rows (489 tiles = 1.91
//                     rounds of 256 CUs, where 256x256 gives 366 tiles = 1.43 rounds and 128x128 is staging-bound)
//   Cfg<2,2,2,64,2>:  64x128, 4 waves of 32x64 (48 KiB, 3 workgroups per CU): twice the waves of the 128x128 tile
//                     for grids that do not fill the chip (per-GPU batches of 30-60 sequences under strong scaling)
template <int WM_, int WN_, int MT_, int BK_, int STAGES_, int PP_ = 0>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, BK = BK_, STAGES = STAGES_;
  static constexpr bool PP = PP_ != 0;                        // ping-pong main loop (nt_mainloop_pp)
  static constexpr int BM = 16 * MT * WM, BN = 64 * WN, NW = WM * WN, THREADS = 64 * NW;
  static constexpr int ROWB = BK * 2;                         // bytes per staged row
  static constexpr int RPI = 1024 / ROWB;                     // rows per LDS-DMA wave-instruction
  static constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  static constexpr int G = (BM + BN) / RPI / NW;              // LDS-DMA wave-instructions per wave per K-step
  static constexpr int JP = MT == 6 ? 3 : (MT < 4 ? MT : 4);  // 16-row sub-tiles of a wave's tile per epilogue pass
  static constexpr int SLAB_ROWS = 16 * JP;                   // rows per pass
  static constexpr int SLAB_BYTES = NW * SLAB_ROWS * 68 * 4;  // epilogue transpose slabs
  static constexpr int LDS = STAGES * STAGE_BYTES > SLAB_BYTES ? STAGES * STAGE_BYTES : SLAB_BYTES;
  static constexpr int WG_PER_CU = LDS <= 53 * 1024 ? 3 : (LDS <= 80 * 1024 ? 2 : 1);
  static constexpr int MIN_WAVES = (WG_PER_CU * NW + 3) / 4;
  static_assert((BM + BN) % (RPI * NW) == 0, "tile rows must split evenly over the waves");
};

// 16-byte chunk swizzle of a staged row (conflict-free ds_read_b128 of 16 rows at one chunk):
// 128-B rows (BK=64): chunk ^= (row>>1)&7 ; 64-B rows (BK=32): chunk ^= (row>>2)&3
template <int BK> __device__ __forceinline__ int kswz(int row) { return BK == 64 ? ((row >> 1) & 7) : ((row >> 2) & 3); }

// Stage one K-step of the block's [W tile (BN rows) | X tile (BM rows)] x BK bf16 into LDS by LDS-DMA: 1 KiB per
// wave-instruction, lane-linear image, swizzle applied on the source side -- with no address arithmetic in the K loop:
// one buffer descriptor per operand with its base at the tile's first row, per-lane byte offsets (row * ld + swizzled
// chunk, constant over K) computed once per tile, the K advance as the instruction's scalar offset.  (Rounds 1-2 formed a
// 64-bit address per LDS-DMA instruction and K step: ~6 VALU + a multiply-add each, on the issue port two waves per SIMD
// share with the MFMAs; the descriptor form is 1-3 % faster on every ring-loop shape, 2-5 % on the 64x128 tile of the
// small per-GPU batches: interleaved A/B of the two builds, round 3.)
template <class C> struct RingStage {
  uint32_t so[C::G];     // per-lane source byte offsets of this wave's G LDS-DMA instructions of a K step
  u32x4 srd_w, srd_x;    // buffer descriptors: W rows from n0, X rows from m0
  uint32_t lds;          // LDS byte address of ring slot 0 (uniform)
};
template <class C>
__device__ __forceinline__ void ring_stage_init(RingStage<C>& st, const GemmNtParams& p, int n0, int m0, uint32_t lds0, int wave, int lane) {
  constexpr int CPR = C::ROWB / 16;
  static_assert(C::BN % (C::RPI * C::NW) == 0, "an LDS-DMA instruction round must not straddle the W / X boundary");
  const uint64_t bw = (uint64_t)(uintptr_t)(p.w + (size_t)n0 * p.ldw), bx = (uint64_t)(uintptr_t)(p.x + (size_t)m0 * p.ldx);
  st.srd_w = u32x4{(uint32_t)bw, (uint32_t)(bw >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  st.srd_x = u32x4{(uint32_t)bx, (uint32_t)(bx >> 32) & 0xffffu, 0xffffffffu, 0x00020000u};
  st.lds = __builtin_amdgcn_readfirstlane(lds0);
#pragma unroll
  for (int r = 0; r < C::G; ++r) {
    const int rr = (r * C::NW + wave) * C::RPI + lane / CPR;   // row in the concatenated [W | X] tile
    const int chunk = (lane % CPR) ^ kswz<C::BK>(rr);
    const bool is_w = r < C::BN / (C::RPI * C::NW);            // compile-time per r
    int g = is_w ? rr : rr - C::BN;                            // row inside the operand's tile
    const int lim = is_w ? p.N - n0 : p.M - m0;
    g = g < lim ? g : lim - 1;                                 // edge rows re-read a valid row; their outputs are never stored
    st.so[r] = (uint32_t)g * (uint32_t)((is_w ? p.ldw : p.ldx) * 2) + (uint32_t)chunk * 16u;
  }
}
template <class C>
__device__ __forceinline__ void ring_stage_one(const RingStage<C>& st, int k0, int slot, int wave, int r) {
  const bool is_w = r < C::BN / (C::RPI * C::NW);
  const uint32_t dst = st.lds + (uint32_t)(slot * C::STAGE_BYTES + (r * C::NW + wave) * 1024);
  const uint32_t soff = (uint32_t)k0 * 2u;
  uint32_t keep;
  if (is_w)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(st.so[r]), "s"(dst), "s"(st.srd_w), "s"(soff) : "memory");
  else
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(st.so[r]), "s"(dst), "s"(st.srd_x), "s"(soff) : "memory");
}
template <class C>
__device__ __forceinline__ void ring_stage_step(const RingStage<C>& st, int k0, int slot, int wave) {
#pragma unroll
  for (int r = 0; r < C::G; ++r) ring_stage_one<C>(st, k0, slot, wave, r);
}

// ---- hand-counted LDS fragment reads (see the main loop of gemm_nt_kernel) -------------------------
template <int OFF> __device__ __forceinline__ bf16x8 lds_read_b128(uint32_t addr) {
  bf16x8 r;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
template <int OFF> __device__ __forceinline__ s16x4 lds_read_tr(uint32_t addr) {
  s16x4 r;
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(r) : "v"(addr), "n"(OFF));
  return r;
}
// acc + sum of the 8 bf16 elements of a fragment: 4 v_dot2c_f32_bf16 against (1, 1)
__device__ __forceinline__ float dot_ones(bf16x8 f, float acc) {
  const bf16x2_t ones2 = {(__bf16)1.0f, (__bf16)1.0f};
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 0, 1), ones2, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 2, 3), ones2, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 4, 5), ones2, acc, false);
  acc = __builtin_amdgcn_fdot2_f32_bf16(__builtin_shufflevector(f, f, 6, 7), ones2, acc, false);
  return acc;
}
// wait until at most N of this wave's LDS reads are outstanding; `a` is tied to the wait so that no
// consumer of the fragment can be scheduled above it
template <int N> __device__ __forceinline__ void lds_wait(bf16x8& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a) : "n"(N)); }

// Issue plan of the fragment pipeline: phase(-1) = [W(0,0..3), X(0), X(1)]; phase(v) = [X(v+2)] +
// [WPN W fragments of the next sub-step while j = v % MT is in [WP0, WP0 + 4/WPN)], all of which are
// requested before the first X fragment of that sub-step.  pending(u) = reads requested after X(u)
// by the time unit u waits for it.
template <int MT, int KS> struct FragPipe {
  static constexpr int U = KS * MT;
  static constexpr bool AFTER = false;   // true: a unit requests its prefetches behind its MFMAs instead of in front of them
  static constexpr int WP0 = MT >= 6 ? 2 : 0, WPN = MT >= 6 ? 1 : 2, WPU = 4 / WPN;   // first unit / frags per unit / units
  static constexpr int npref_w(int v) {
    return (v >= 0 && v / MT + 1 < KS && v % MT >= WP0 && v % MT < WP0 + WPU) ? WPN : 0;
  }
  static constexpr int nx(int v) { return v + 2 < U ? 1 : 0; }
  // requests up to and including phase p (phase -1 = the 6 prologue reads)
  static constexpr int upto(int p) {
    int c = 6;
    for (int v = 0; v <= p; ++v) c += nx(v) + npref_w(v);
    return c;
  }
  static constexpr int pos_x(int u) { return u < 2 ? 4 + u : upto(u - 3); }          // X(u) is the first request of phase u-2
  static constexpr int pos_last_w(int ks) {                                            // last W fragment of sub-step ks
    if (ks == 0) return 3;
    const int vl = (ks - 1) * MT + WP0 + WPU - 1;
    return upto(vl) - 1;
  }
  // requests that may still be outstanding when unit u starts its MFMAs: everything requested so far minus
  // everything up to the LAST request the unit needs (its X fragment; for the first unit of a sub-step also the
  // sub-step's W fragments, which for small MT are requested after that X fragment)
  static constexpr int pending(int u) {
    const int issued = upto(AFTER ? u - 1 : u);
    int need = pos_x(u);
    if (u % MT == 0 && pos_last_w(u / MT) > need) need = pos_last_w(u / MT);
    return issued - 1 - need;
  }
  static constexpr bool ok() {
    for (int u = 0; u < U; ++u)
      if (pending(u) < 0 || pending(u) > 15) return false;
    return true;
  }
  static_assert(MT == 2 || MT == 4 || MT == 6 || MT == 8, "FragPipe: unit plans exist for 2, 4, 6 and 8 sub-tiles");
};


template <int BK> __device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int chunk) {
  typedef __attribute__((address_space(3))) const bf16x8* lds_frag_ptr;
  return *(lds_frag_ptr)LDS_PTR(lds_tile + row * (BK * 2) + ((chunk ^ kswz<BK>(row)) << 4));
}

template <int N> __device__ __forceinline__ void wait_vmcnt() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// Shared epilogue of the NT kernels (inlined): bias / activation / residual / dropout on the wave's
// [16*MT rows][64 columns] accumulator tile, stored row-contiguously.
template <class C, int EPI, bool OUT_F32>
__device__ __forceinline__ void nt_epilogue(const GemmNtParams& p, f32x4 (&acc)[4][C::MT], char* smem, int m0, int n0,
                                            int wm, int wn, int wave, int lane) {
  constexpr int MT = C::MT;
  // ---- epilogue.  The accumulator layout (lane = row m, 4 consecutive n per register quad) would need
  // 16 strided 8-byte stores (+16 such loads of the residual) per lane, which is store-ISSUE bound and
  // cost more than the 12-step main loop of the K=768 GEMMs.  Instead every wave transposes its tile,
  // 64 rows at a time, through its own fp32 LDS slab (the ring is dead by now) and walks it
  // row-contiguously: a lane owns 8 consecutive columns of one row -> 16-byte bias/residual loads and
  // 16-byte stores, 8 full 128-byte row segments per wave-instruction.
  // Outputs (and the once-read residual / multiplier operand) use the NON-TEMPORAL cache policy: the workgroups of a
  // launch reach their epilogues together and a round's 33 MB of output is the size of the eight L2s, so with the
  // default write-back policy the stores of every round waited for evictions (N=3072, K=768, "x aux" epilogue: 215 us
  // -> 187 us with nt stores + nt operand loads, 167 -> 155 us for the 128x128 tile at K=3072; 200-launch averages of
  // alternating builds in one gpurun call).  Nothing re-reads these lines before the next kernel does.
  constexpr int SLAB_LD = 68;                       // floats per slab row (64 + 4 pad)
  float* slab = reinterpret_cast<float*>(smem) + wave * (C::SLAB_ROWS * SLAB_LD);
  const int c0 = (lane & 7) * 8;
  const int n = n0 + wn * 64 + c0;
  const bool ncols_ok_ = n < p.N;
  const bool full_ = (n + 7 < p.N);
  // fp32 outputs: a lane that owns 8 CONSECUTIVE fp32 columns writes them with two 16-byte stores 16 bytes apart, so one
  // store instruction covers half of every 128-byte line it touches: measured 3.2-3.5 TB/s for a [31k, 768] fp32 matrix
  // against 6.0 TB/s when every instruction writes whole lines (tools/exp/store_pattern.hip), and +17.6 us for the fp32
  // output of an N = 768, K = 768 GEMM over its bf16 one (48 MB more at 2.7 TB/s).  In the FAST walk of the plain fp32
  // epilogues a lane therefore owns columns 4l..4l+3 (group A) and 32+4l..32+4l+3 (group B) of the wave's 64: element e of
  // the lane's 8 values is column colA + e (e < 4) or colB + e - 4.  Everywhere else colA = n, colB = n + 4 (the old map).
  constexpr bool SPLITCOL = OUT_F32 && (EPI == UNIMM_EPI_BIAS || EPI == UNIMM_EPI_BIAS_DROP_RESID || EPI == UNIMM_EPI_BIAS_RELU);
  const bool fastw_ = (n0 + wn * 64 + 64 <= p.N) && ((p.ldo * (OUT_F32 ? 4 : 2)) % 16 == 0) &&
                      (!(EPI == UNIMM_EPI_BIAS_DROP_RESID || EPI == UNIMM_EPI_DGELU || EPI == UNIMM_EPI_ADD || EPI == UNIMM_EPI_MUL) ||
                       (p.ldaux * (EPI == UNIMM_EPI_BIAS_DROP_RESID ? 4 : 2)) % 16 == 0);
  const bool splitc = SPLITCOL && fastw_;                         // wave-uniform
  const int colA = splitc ? n0 + wn * 64 + (lane & 7) * 4 : n, colB = splitc ? colA + 32 : n + 4;
  const int cA = colA - (n0 + wn * 64), cB = colB - (n0 + wn * 64);       // the same inside the wave's slab row
  // Residual / multiplier operand: loaded in batches of PF row-walk iterations, one batch ahead of its use (the first
  // before the barrier below; registers: the main loop's fragment registers are dead).  In program order the walk used
  // to reach each load only after the slab reads of its iteration, and because loads and stores share vmcnt (and may
  // retire out of order with respect to each other) the compiler can only wait vmcnt(0): every one of the 16 iterations
  // paid a full load round trip plus the acknowledgement of the previous iteration's stores.  Batched, there is one
  // such drain per batch, and the loads it waits for were requested a whole batch earlier.
  constexpr bool AUX32 = EPI == UNIMM_EPI_BIAS_DROP_RESID;
  constexpr bool AUX16 = EPI == UNIMM_EPI_DGELU || EPI == UNIMM_EPI_ADD || EPI == UNIMM_EPI_MUL;
  constexpr int NIT = 2 * C::JP, NWALK = (MT / C::JP) * NIT;
  constexpr int PF = (AUX32 || AUX16) ? (AUX32 && NIT % 2 == 0 ? NIT / 2 : NIT) : 0;
  constexpr int PFN = PF > 0 ? PF : 1;
  const bool pf_on = PF > 0 && p.aux != nullptr && ((p.ldaux * (AUX32 ? 4 : 2)) % 16 == 0);   // uniform
  const int pf_n = full_ ? n : 0;                     // lanes at the ragged N edge take the scalar path; their prefetch is ignored
  const int pf_a = full_ ? colA : 0, pf_b = full_ ? colB : 4;
  f32x4 pf0[2][PFN], pf1[2][PFN];
  float pfmu[2][PFN], pfrs[2][PFN];
  auto aux_prefetch = [&](int batch, auto sure) {    // requests walk iterations [batch * PF, (batch + 1) * PF)
    if constexpr (PF > 0) {
      if (batch * PF < NWALK && (decltype(sure)::value || pf_on)) {
#pragma unroll
        for (int k = 0; k < PF; ++k) {
          const int g = batch * PF + k;
          const int row = (g % NIT) * 8 + (lane >> 3);
          int m = m0 + wm * 16 * MT + (g / NIT) * 16 * C::JP + row;
          m = m < p.M ? m : p.M - 1;
          if constexpr (AUX32) {
            const float* ap = reinterpret_cast<const float*>(p.aux) + (size_t)m * p.ldaux;
            pf0[batch & 1][k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ap + pf_a));
            pf1[batch & 1][k] = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ap + pf_b));
            if (p.aux_mean != nullptr) { pfmu[batch & 1][k] = p.aux_mean[m]; pfrs[batch & 1][k] = p.aux_rstd[m]; }
          } else {
            const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.aux) + (size_t)m * p.ldaux + pf_n;
            pf0[batch & 1][k] = __builtin_bit_cast(f32x4, __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ap)));
          }
        }
      }
    }
  };
  aux_prefetch(0, std::false_type{});
  __builtin_amdgcn_s_barrier();                      // all waves finished reading the ring
  float b[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int c = e < 4 ? colA + e : colB + e - 4;
    b[e] = (p.bias != nullptr && c < p.N) ? p.bias[c] : 0.f;
  }
  float lg[8], lb[8];                                // LayerNorm-on-the-fly residual (DROP_RESID only)
  const bool aux_ln = EPI == UNIMM_EPI_BIAS_DROP_RESID && p.aux_mean != nullptr;
  if (aux_ln) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int c = e < 4 ? colA + e : colB + e - 4;
      lg[e] = c < p.N ? p.aux_gamma[c] : 0.f;
      lb[e] = c < p.N ? p.aux_beta[c] : 0.f;
    }
  }
  const bool vec_out_ = full_ && ((p.ldo * (OUT_F32 ? 4 : 2)) % 16 == 0);
  const bool vec_aux_ = full_ && ((p.ldaux * (EPI == UNIMM_EPI_BIAS_DROP_RESID ? 4 : 2)) % 16 == 0);
  constexpr int JP = C::JP;                          // 16-row sub-tiles per pass (a 32-row wave tile has only two)
  // GELU epilogues with bf16 outputs do their arithmetic BEFORE the transposition, on the accumulator registers:
  // elementwise math does not care about the layout, and there every lane has 16 x JP independent values in flight
  // (the row-walk below has 8 behind an LDS read per iteration: the erf + exp + rcp chains of the fused GELU / GELU'
  // ran at ~10 cycles per instruction and cost 90 us of the 234 us ff1 GEMM at 240 sequences).  The two results of an
  // element travel through the slab as ONE 32-bit word (bf16 pair: low = output, high = second output).
  constexpr bool PRE = (EPI == UNIMM_EPI_BIAS_GELU || EPI == UNIMM_EPI_BIAS_GELU_DG) && !OUT_F32;
  f32x4 bq[4];                                       // bias of the lane's accumulator columns 16 i + 4 (lane >> 4) + e
  if constexpr (PRE) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int na = n0 + wn * 64 + i * 16 + 4 * (lane >> 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) bq[i][e] = (p.bias != nullptr && na + e < p.N) ? p.bias[na + e] : 0.f;
    }
  }
  // The walk exists twice: FAST = the wave's 64 columns are all inside N and every row pointer is 16-byte aligned (wave-
  // uniform), so no lane ever takes an element-wise path.  Keeping the element-wise loads out of that instance is what
  // lets the batched operand loads above work: with divergent load paths at every join the compiler's conservative
  // vmcnt(0)s also waited for the batch that had just been requested.
  const bool fastw = fastw_;
  auto walk = [&](auto fast_tag) {
  constexpr bool FAST = decltype(fast_tag)::value;
  const bool ncols_ok = FAST ? true : ncols_ok_;
  const bool full = FAST ? true : full_;
  const bool vec_out = FAST ? true : vec_out_;
  const bool vec_aux = FAST ? true : vec_aux_;
#pragma unroll
  for (int pass = 0; pass < MT / JP; ++pass) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < JP; ++j) {
        float* dst = slab + (j * 16 + (lane & 15)) * SLAB_LD + i * 16 + 4 * (lane >> 4);
        if constexpr (PRE) {
          u32x4 w;
          if constexpr (EPI == UNIMM_EPI_BIAS_GELU) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const float x = acc[i][pass * JP + j][e] + bq[i][e];
              w[e] = pack2bf(gelu_erf(x), x);
            }
          } else {                                           // GELU and GELU', packed fp32 math on element pairs
#pragma unroll
            for (int e = 0; e < 4; e += 2) {
              const f32x4 a4 = acc[i][pass * JP + j];
              const f32x2v x = f32x2v{a4[e], a4[e + 1]} + f32x2v{bq[i][e], bq[i][e + 1]};
              f32x2v y, d;
              gelu_and_grad2(x, y, d);
              w[e] = pack2bf(y.x, d.x);
              w[e + 1] = pack2bf(y.y, d.y);
            }
          }
          *reinterpret_cast<u32x4*>(dst) = w;
        } else {
          *reinterpret_cast<f32x4*>(dst) = acc[i][pass * JP + j];
        }
      }
#pragma unroll
    for (int it = 0; it < 2 * JP; ++it) {
      const int row = it * 8 + (lane >> 3);
      const int m = m0 + wm * 16 * MT + pass * 16 * JP + row;
      const f32x4 lo = *reinterpret_cast<const f32x4*>(slab + row * SLAB_LD + cA);
      const f32x4 hi = *reinterpret_cast<const f32x4*>(slab + row * SLAB_LD + cB);
      const int gw = pass * NIT + it;                  // walk index; at the start of a batch request the next one
      if (PF > 0 && gw % PFN == 0) {
        // drain first (this batch's operands, requested a batch ago), THEN request: with the order reversed the
        // compiler's vmcnt(0) in front of the first use would also wait for the loads just issued
        if (FAST || pf_on) __builtin_amdgcn_s_waitcnt(0x0F70);
        aux_prefetch(gw / PFN + 1, fast_tag);
      }
      const f32x4 pfa0 = pf0[(gw / PFN) & 1][gw % PFN], pfa1 = pf1[(gw / PFN) & 1][gw % PFN];
      const float pfamu = pfmu[(gw / PFN) & 1][gw % PFN], pfars = pfrs[(gw / PFN) & 1][gw % PFN];
      (void)pfa1; (void)pfamu; (void)pfars;
      if (m >= p.M || !ncols_ok) continue;
      if constexpr (PRE) {
        uint32_t pw[8];
#pragma unroll
        for (int e = 0; e < 4; ++e) { pw[e] = __float_as_uint(lo[e]); pw[4 + e] = __float_as_uint(hi[e]); }
        bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n;
        bf16_t* up = p.out2 != nullptr ? p.out2 + (size_t)m * p.ldo + n : nullptr;
        if (vec_out) {
          u32x4 ow, uw;
#pragma unroll
          for (int k2 = 0; k2 < 4; ++k2) {
            ow[k2] = __builtin_amdgcn_perm(pw[2 * k2 + 1], pw[2 * k2], 0x05040100u);   // low halves of two words
            uw[k2] = __builtin_amdgcn_perm(pw[2 * k2 + 1], pw[2 * k2], 0x07060302u);   // high halves
          }
          __builtin_nontemporal_store(ow, reinterpret_cast<u32x4*>(op));
          if (up != nullptr) __builtin_nontemporal_store(uw, reinterpret_cast<u32x4*>(up));
        } else {
          for (int e = 0; e < 8; ++e)
            if (n + e < p.N) {
              op[e] = (bf16_t)(pw[e] & 0xffffu);
              if (up != nullptr) up[e] = (bf16_t)(pw[e] >> 16);
            }
        }
        continue;
      }
      float v[8], u[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[e] = lo[e] + b[e]; v[4 + e] = hi[e] + b[4 + e]; }
      if constexpr (EPI == UNIMM_EPI_BIAS_DROP_RESID || EPI == UNIMM_EPI_DGELU || EPI == UNIMM_EPI_ADD || EPI == UNIMM_EPI_MUL) {
        float a[8];
        if constexpr (EPI == UNIMM_EPI_BIAS_DROP_RESID) {   // fp32 residual stream
          const float* ap = reinterpret_cast<const float*>(p.aux) + (size_t)m * p.ldaux;
          if (vec_aux) {
            f32x4 r0, r1;
            if (PF > 0) { r0 = pfa0; r1 = pfa1; }
            else {
              r0 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ap + colA));
              r1 = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(ap + colB));
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[e] = r0[e]; a[4 + e] = r1[e]; }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (n + e < p.N) ? ap[n + e] : 0.f;      // (ragged edge: never the split map)
          }
          if (aux_ln) {
            const bool pfd = PF > 0 && (FAST || pf_on);
            const float mu = pfd ? pfamu : p.aux_mean[m], rs = pfd ? pfars : p.aux_rstd[m];
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (a[e] - mu) * rs * lg[e] + lb[e];
          }
          if (p.drop.thr != 0u) {
            const uint32_t kb = (SPLITCOL && FAST)
                ? (drop_bits4(p.drop, (uint32_t)m, (uint32_t)p.N, (uint32_t)colA) | (drop_bits4(p.drop, (uint32_t)m, (uint32_t)p.N, (uint32_t)colB) << 4))
                : drop_bits8(p.drop, (uint32_t)m, (uint32_t)p.N, (uint32_t)n);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = ((kb >> e) & 1u) ? v[e] * p.drop.scale : 0.f;
          }
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] += a[e];
        } else {
          const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.aux) + (size_t)m * p.ldaux + n;
          if (vec_aux) {
            const u32x4 raw = PF > 0 ? __builtin_bit_cast(u32x4, pfa0)
                                     : __builtin_nontemporal_load(reinterpret_cast<const u32x4*>(ap));
#pragma unroll
            for (int e = 0; e < 4; ++e) { a[2 * e] = __uint_as_float(raw[e] << 16); a[2 * e + 1] = __uint_as_float(raw[e] & 0xffff0000u); }
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] = (n + e < p.N) ? bf2f(ap[e]) : 0.f;
          }
          if constexpr (EPI == UNIMM_EPI_DGELU) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_grad(a[e]);
          } else if constexpr (EPI == UNIMM_EPI_MUL) {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] *= a[e];
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += a[e];
          }
        }
      }
      if constexpr (EPI == UNIMM_EPI_BIAS_GELU || EPI == UNIMM_EPI_BIAS_GELU_DG) {
        if constexpr (EPI == UNIMM_EPI_BIAS_GELU) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { u[e] = v[e]; v[e] = gelu_erf(v[e]); }
        } else {
#pragma unroll
          for (int e = 0; e < 8; ++e) { const float x = v[e]; gelu_and_grad(x, v[e], u[e]); }
        }
        if (p.out2 != nullptr) {
          bf16_t* up = p.out2 + (size_t)m * p.ldo + n;
          if (full && (p.ldo % 8) == 0)
            __builtin_nontemporal_store(u32x4{pack2bf(u[0], u[1]), pack2bf(u[2], u[3]), pack2bf(u[4], u[5]), pack2bf(u[6], u[7])},
                                        reinterpret_cast<u32x4*>(up));
          else
            for (int e = 0; e < 8; ++e) if (n + e < p.N) up[e] = f2bf(u[e]);
        }
      }
      if constexpr (EPI == UNIMM_EPI_BIAS_RELU) {
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (OUT_F32) {
        float* op = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo;
        if (vec_out) {
          __builtin_nontemporal_store(f32x4{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4*>(op + colA));
          __builtin_nontemporal_store(f32x4{v[4], v[5], v[6], v[7]}, reinterpret_cast<f32x4*>(op + colB));
        } else {
          for (int e = 0; e < 8; ++e) if (n + e < p.N) op[n + e] = v[e];
        }
      } else {
        bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n;
        if (vec_out) {
          const u32x4 pk = u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
          __builtin_nontemporal_store(pk, reinterpret_cast<u32x4*>(op));
        } else
          for (int e = 0; e < 8; ++e) if (n + e < p.N) op[e] = f2bf(v[e]);
      }
    }
  }
  };  // walk
  if (fastw) walk(std::true_type{}); else walk(std::false_type{});
}

// ------------------------------------------------------------------------------------------------
// Ping-pong main loop of the 256x256x64 tile (Cfg<2,4,8,64,2,1>).
//
// The loop above runs all 8 waves through the same fragment stream in lock-step: the two waves of a SIMD want the
// LDS, the LDS-DMA issue slots and the matrix pipe at the same moments, and the step-top vmcnt(0) + barrier drains the
// whole pipeline once per K-step.  Here the waves form two groups, G0 = waves 0-3 (rows 0-127 of the tile) and
// G1 = waves 4-7 (rows 128-255) - one wave of each per SIMD - that run the SAME program one barrier apart (G1 passes
// one extra barrier first), so that in every barrier interval one group issues nothing but its 16 MFMAs (one
// quadrant of its 128x64 accumulator tile x the whole 64-deep K-tile) while the other requests the fragments of its
// next quadrant and issues its share of the LDS-DMA staging (the 8-phase structure of the CDNA guide's 256^2 GEMM).
// Per K-tile and wave: 4 phases = 8 barriers,
//   phase 0: read X(m-half 0: 8 fragments) + W(n-half 0: 4)  | MFMA quadrant (m0, n0)
//   phase 1: read W(n-half 1: 4)                              | MFMA quadrant (m0, n1)
//   phase 2: read X(m-half 1: 8, same registers)              | MFMA quadrant (m1, n1)
//   phase 3: no reads (W(n-half 0) stays in registers)        | MFMA quadrant (m1, n0)
// Staging: the K-tile's 64 KiB are four 16 KiB half-tiles [W rows 0-127 | W 128-255 | X 0-127 | X 128-255]; every phase
// stages ONE half-tile (each wave 2 LDS-DMA instructions, in its read interval), phase P the half-tile (P+1) % 4 of
// K-tile (P+1) / 4 + 1.  Two buffers; a half-tile is restaged at least two barrier intervals after the lgkmcnt(0) that
// retired its last fragment read in BOTH groups (WAR), and a K-tile is certified by ONE counted wait per K-tile: in
// phase 3 every wave, after issuing that phase's 2 DMAs, waits vmcnt(2) - all DMAs of the next K-tile are older - and
// the barrier that follows orders them before the first reads of the next K-tile in either group (RAW).  Proof sketch
// with I_k = the interval before barrier k; G0 reads phase P in I_2P and computes it in I_2P+1, G1 one interval later:
//   buffer b of K-tile t: last W reads in phase 4t+1 (G1: I_8t+3, retired at the start of I_8t+4), last X reads in phase
//   4t+2 (G0 I_8t+4 / G1 I_8t+5, retired at the start of I_8t+5 / I_8t+6); restaged by phases 4t+3 (W0: I_8t+6, I_8t+7),
//   4t+4 (W1), 4t+5 (X0: I_8t+10, I_8t+11), 4t+6 (X1); certified in phase 4t+7 (I_8t+14 / I_8t+15), first read of K-tile
//   t+2 in I_8t+16.
// ------------------------------------------------------------------------------------------------
// Staging addresses of one wave: half-tile ht (0,1 = W rows 0-127 / 128-255; 2,3 = X) is 16 wave-instructions of 8 rows
// x 128 B; this wave issues two of them, inst = group * 8 + (wave & 3) * 2 + r.  Sources are a 32-bit byte offset per
// lane (row * ld + swizzled 16-B chunk, constant over K) on top of a scalar base that advances 128 B per K-tile.
struct PpStage {
  uint32_t so[4][2];       // per-lane source byte offsets
  uint32_t lds;            // LDS byte address of this wave's first instruction slot in half-tile 0 of buffer 0 (uniform)
};

template <class C>
__device__ __forceinline__ void pp_stage_init(PpStage& st, const GemmNtParams& p, int n0, int m0, uint32_t lds0, int wave, int lane) {
  const int inst0 = (wave >> 2) * 8 + (wave & 3) * 2;
  st.lds = __builtin_amdgcn_readfirstlane(lds0 + inst0 * 1024);
#pragma unroll
  for (int ht = 0; ht < 4; ++ht)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int rl = (inst0 + r) * 8 + (lane >> 3);            // row inside the half-tile
      const int chunk = (lane & 7) ^ kswz<64>(rl);
      const int rr = (ht & 1) * 128 + rl;                      // row inside the W (or X) tile
      int g = (ht < 2 ? n0 : m0) + rr;
      const int lim = ht < 2 ? p.N : p.M;
      g = g < lim ? g : lim - 1;                               // edge rows re-read a valid row; their outputs are never stored
      st.so[ht][r] = (uint32_t)g * (uint32_t)((ht < 2 ? p.ldw : p.ldx) * 2) + chunk * 16;
    }
}

// Issued through inline asm: the compiler must not know that LDS-DMA is in flight.  With the builtin it tracks the pending
// "VMEM write to LDS" across the loop's back edge and puts an s_waitcnt vmcnt(0) in front of the first MFMA cluster of every
// K-tile (seen in the .s), which drains the half-tile that was just requested and serialises staging with compute once
// per K-tile.  Completion is counted by hand (wait_vmcnt in nt_mainloop_pp).
template <class C>
__device__ __forceinline__ void pp_stage(const PpStage& st, const GemmNtParams& p, int kt, int ht) {
  const char* base = (const char*)(ht < 2 ? p.w : p.x) + (size_t)kt * 128;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint32_t dst = st.lds + (kt & 1) * 65536 + ht * 16384 + r * 1024;
    uint32_t keep;                                             // m0 is the compiler's: hand it back as found
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(st.so[ht][r]), "s"(dst), "s"(base) : "memory");
  }
}

template <class C>
__device__ __forceinline__ void nt_mainloop_pp(const GemmNtParams& p, char* smem, int m0, int n0, f32x4 (&acc)[4][C::MT]) {
  static_assert(C::WM == 2 && C::WN == 4 && C::MT == 8 && C::BK == 64 && C::STAGES == 2, "ping-pong loop: 256x256x64, 8 waves");
  constexpr int RB = 128;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int wm = wave >> 2, wn = wave & 3;
  const int nk = p.K / 64;
  // fragment addresses in buffer 0 (second 32-deep sub-step = a second base: the swizzle is an XOR)
  uint32_t aw0[2], ax0[2];
  PpStage st;
  {
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)LDS_PTR(smem);
    pp_stage_init<C>(st, p, n0, m0, lds0, wave, lane);
    const int rw = wn * 64 + (lane & 15), rx = wm * 128 + (lane & 15), cq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      aw0[ks] = lds0 + rw * RB + (((ks * 4 + cq) ^ kswz<64>(rw)) << 4);
      ax0[ks] = lds0 + (256 + rx) * RB + (((ks * 4 + cq) ^ kswz<64>(rx)) << 4);
    }
  }
  // prologue: K-tile 0 (4 half-tiles) and half-tile 0 of K-tile 1
#pragma unroll
  for (int ht = 0; ht < 4; ++ht) pp_stage<C>(st, p, 0, ht);
  // vmcnt(0) through the builtin, not asm and not vmcnt(2): a full wait the compiler can SEE retires the stores it still
  // tracks from the previous tile's epilogue (persistent kernel); with anything less it protects their data registers
  // with a vmcnt(0) of its own in front of the first MFMA cluster of every K-tile, which drains the staging pipeline.
  if (nk > 1) pp_stage<C>(st, p, 1, 0);
  __builtin_amdgcn_s_waitcnt(0x0F70);
  __builtin_amdgcn_s_barrier();                        // K-tile 0 has landed for every wave
  if (wm == 1) __builtin_amdgcn_s_barrier();           // G1 runs one barrier behind G0 (matched by G0's first loop barrier)
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 fx[2][4], fw[2][4];
  // phase P stages half-tile (P + 1) % 4 of K-tile (P + 1) / 4 + 1
#define UNIMM_PP_STAGE(P_)                                                                       \
  {                                                                                              \
    const int u_ = (P_) + 1, kt_ = (u_ >> 2) + 1;                                                \
    if (kt_ < nk) pp_stage<C>(st, p, kt_, u_ & 3);                                               \
  }
#define UNIMM_PP_MFMA(JH, IH)                                                                    \
  {                                                                                              \
    __builtin_amdgcn_s_setprio(1);                                                               \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                             \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                           \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                         \
          acc[2 * (IH) + ii][4 * (JH) + jj] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(           \
              fw[ks][2 * (IH) + ii], fx[ks][jj], acc[2 * (IH) + ii][4 * (JH) + jj], 0, 0, 0);    \
    __builtin_amdgcn_s_setprio(0);                                                               \
  }
#define UNIMM_PP_READ_X(JH)                                                                      \
  {                                                                                              \
    fx[0][0] = lds_read_b128<((JH) * 4 + 0) * 16 * RB>(ax[0]); fx[0][1] = lds_read_b128<((JH) * 4 + 1) * 16 * RB>(ax[0]); \
    fx[0][2] = lds_read_b128<((JH) * 4 + 2) * 16 * RB>(ax[0]); fx[0][3] = lds_read_b128<((JH) * 4 + 3) * 16 * RB>(ax[0]); \
    fx[1][0] = lds_read_b128<((JH) * 4 + 0) * 16 * RB>(ax[1]); fx[1][1] = lds_read_b128<((JH) * 4 + 1) * 16 * RB>(ax[1]); \
    fx[1][2] = lds_read_b128<((JH) * 4 + 2) * 16 * RB>(ax[1]); fx[1][3] = lds_read_b128<((JH) * 4 + 3) * 16 * RB>(ax[1]); \
  }
#define UNIMM_PP_READ_W(IH)                                                                      \
  {                                                                                              \
    fw[0][2 * (IH)] = lds_read_b128<(2 * (IH)) * 16 * RB>(aw[0]); fw[0][2 * (IH) + 1] = lds_read_b128<(2 * (IH) + 1) * 16 * RB>(aw[0]); \
    fw[1][2 * (IH)] = lds_read_b128<(2 * (IH)) * 16 * RB>(aw[1]); fw[1][2 * (IH) + 1] = lds_read_b128<(2 * (IH) + 1) * 16 * RB>(aw[1]); \
  }
#define UNIMM_PP_SYNC_READS()                                                                    \
  __builtin_amdgcn_s_barrier();                                                                  \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fx[0][0]), "+v"(fx[0][1]), "+v"(fx[0][2]), "+v"(fx[0][3]), "+v"(fx[1][0]), \
               "+v"(fx[1][1]), "+v"(fx[1][2]), "+v"(fx[1][3]), "+v"(fw[0][0]), "+v"(fw[0][1]), "+v"(fw[0][2]), "+v"(fw[0][3]), \
               "+v"(fw[1][0]), "+v"(fw[1][1]), "+v"(fw[1][2]), "+v"(fw[1][3]));                 \
  __builtin_amdgcn_sched_barrier(0);

  for (int t = 0; t < nk; ++t) {
    const uint32_t off = (uint32_t)((t & 1) * 65536);
    const uint32_t aw[2] = {aw0[0] + off, aw0[1] + off}, ax[2] = {ax0[0] + off, ax0[1] + off};
    const int P = 4 * t;
    // ---- phase 0
    UNIMM_PP_READ_W(0)
    UNIMM_PP_READ_X(0)
    UNIMM_PP_STAGE(P)
    UNIMM_PP_SYNC_READS()
    UNIMM_PP_MFMA(0, 0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 1
    UNIMM_PP_READ_W(1)
    UNIMM_PP_STAGE(P + 1)
    UNIMM_PP_SYNC_READS()
    UNIMM_PP_MFMA(0, 1)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2
    UNIMM_PP_READ_X(1)
    UNIMM_PP_STAGE(P + 2)
    UNIMM_PP_SYNC_READS()
    UNIMM_PP_MFMA(1, 1)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: no fragment reads; certify K-tile t+1 (every DMA older than this phase's two has landed)
    {
      const int u_ = P + 4, kt_ = (u_ >> 2) + 1;
      if (kt_ < nk) { pp_stage<C>(st, p, kt_, 0); wait_vmcnt<2>(); } else wait_vmcnt<0>();
    }
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    UNIMM_PP_MFMA(1, 0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wm == 0) __builtin_amdgcn_s_barrier();           // G0 waits for G1's last phase (barrier counts match again)
#undef UNIMM_PP_STAGE
#undef UNIMM_PP_MFMA
#undef UNIMM_PP_READ_X
#undef UNIMM_PP_READ_W
#undef UNIMM_PP_SYNC_READS
}

// Split-K join.  The small per-GPU batches of a split global batch (30-60 sequences: ~4-8k rows) give a K = 2304 / 3072
// GEMM a few hundred tiles whose 36-48 step reductions are one dependent chain per workgroup: the chip is half empty and
// the kernel's time is the chain's.  With ksplit workgroups per tile each reduces 1 / ksplit of K; every one stores its
// fp32 partial tile to its own slab (register order, fully coalesced) and takes a ticket; the LAST arriver adds the other
// slabs to the partial it still holds and runs the normal epilogue.  Hand-off (placement-independent; MI355X_MICROARCH.md
// "inter-workgroup visibility"): write-through slab stores -> every wave's vmcnt(0) -> workgroup barrier -> lane 0: relaxed
// agent-scope ticket; the last arriver: agent-scope acquire, vmcnt(0), barrier, loads.  Returns false for the
// workgroups that are done.
template <class C>
__device__ __forceinline__ bool nt_split_join(const GemmNtParams& p, f32x4 (&acc)[4][C::MT], char* smem, int tile, int split) {
  constexpr int MT = C::MT, TILE_F4 = C::NW * 4 * MT * 64;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  f32x4* mine = reinterpret_cast<f32x4*>(p.slabs) + ((size_t)tile * p.ksplit + split) * TILE_F4;
  // WRITE-THROUGH (sc0 sc1) 16-byte stores: the bytes leave the XCD's L2 as they are written, so the publisher needs no
  // agent-scope release (buffer_wbl2 writes back EVERY dirty line of the L2, and with a few hundred workgroups publishing
  // 32 KiB each at the same time that fence cost more than the reduction it saved: 57 vs 33 us for K = 3072 at 3.9k rows).
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      f32x4* dst = mine + ((wave * 4 + i) * MT + j) * 64 + lane;
      asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(dst), "v"(acc[i][j]) : "memory");
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();                                               // (every wave has left the ring: smem[0..3] carries the ticket)
  int* flag = reinterpret_cast<int*>(smem);
  if (tid == 0) {
    const int ticket = __hip_atomic_fetch_add(p.counters + tile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (ticket == p.ksplit - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __hip_atomic_store(p.counters + tile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    }
    *flag = ticket;
  }
  __syncthreads();
  const bool last = *flag == p.ksplit - 1;                        // workgroup-uniform
  __syncthreads();                                               // the flag word is part of the epilogue's slab
  if (!last) return false;
  // The other splits' slabs, in split order, on top of the partial this workgroup still holds in registers.  With two splits
  // (what the engine asks for) the sum does not depend on who arrived last (fp32 addition commutes); with three or four the
  // association does, i.e. the last bits may differ between runs, like the atomically accumulated weight gradients.
  const f32x4* base = reinterpret_cast<const f32x4*>(p.slabs) + (size_t)tile * p.ksplit * TILE_F4;
  for (int sp = 0; sp < p.ksplit; ++sp) {
    if (sp == split) continue;
    const f32x4* other = base + (size_t)sp * TILE_F4;
    // sc0 sc1 loads (L1-bypassing, like the stores that published the bytes): together with the acquire above this is the
    // guide's "write-through stores and loads on both sides" hand-off as well as its "acquire + loads" one.  Eight loads in
    // flight per lane, one wait that names their registers (so no use can move above it).
    static_assert((4 * MT) % 8 == 0, "split-K join: 8 loads per batch");
#pragma unroll
    for (int c = 0; c < 4 * MT; c += 8) {
      f32x4 o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const f32x4* src = other + ((wave * 4 + (c + e) / MT) * MT + (c + e) % MT) * 64 + lane;
        asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(o[e]) : "v"(src) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(o[0]), "+v"(o[1]), "+v"(o[2]), "+v"(o[3]), "+v"(o[4]), "+v"(o[5]), "+v"(o[6]), "+v"(o[7])
                   :: "memory");
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int i = (c + e) / MT, j = (c + e) % MT;
        acc[i][j][0] += o[e][0]; acc[i][j][1] += o[e][1]; acc[i][j][2] += o[e][2]; acc[i][j][3] += o[e][3];
      }
    }
  }
  return true;
}

// One output tile (logical tile id `lid`, already XCD-remapped): ring-staged main loop + epilogue.
template <class C, int EPI, bool OUT_F32>
__device__ __forceinline__ void nt_tile(const GemmNtParams& p, char* smem, int lid) {
  constexpr int BM = C::BM, BN = C::BN, BK = C::BK, S = C::STAGES, G = C::G, MT = C::MT;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  int split = 0;
  if (!C::PP && p.ksplit > 1) { split = lid % p.ksplit; lid /= p.ksplit; }   // the splits of a tile are neighbours in the remapped order
  const int nbn = (p.N + BN - 1) / BN, nbm = (p.M + BM - 1) / BM;
  int tm, tn;
  tile_of(lid, nbm, nbn, p.gn, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wave / C::WN, wn = wave % C::WN;

  f32x4 acc[4][MT];  // [n-subtile i][m-subtile j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  // LDS byte addresses of this lane's fragments in ring slot 0, per 32-deep sub-step (the XOR swizzle
  // makes the second sub-step a second base, not a constant offset)
  uint32_t aw0[BK / 32], ax0[BK / 32];
  {
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)LDS_PTR(smem);
    const int rw = wn * 64 + (lane & 15), rx = wm * 16 * MT + (lane & 15), cq = lane >> 4;
#pragma unroll
    for (int ks = 0; ks < BK / 32; ++ks) {
      aw0[ks] = lds0 + rw * C::ROWB + (((ks * 4 + cq) ^ kswz<BK>(rw)) << 4);
      ax0[ks] = lds0 + (BN + rx) * C::ROWB + (((ks * 4 + cq) ^ kswz<BK>(rx)) << 4);
    }
  }
  if constexpr (C::PP) {
    nt_mainloop_pp<C>(p, smem, m0, n0, acc);
    nt_epilogue<C, EPI, OUT_F32>(p, acc, smem, m0, n0, wm, wn, wave, lane);
    return;
  }
  int nk = p.K / BK, kb = 0;                 // this workgroup reduces K-steps [kb, kb + nk)
  if (p.ksplit > 1) {
    const int per = (nk + p.ksplit - 1) / p.ksplit;
    kb = split * per;
    nk = (nk - kb) < per ? (nk - kb) : per;  // >= 1: the host only splits when (ksplit - 1) * per < K / BK
  }
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  RingStage<C> rst;
  ring_stage_init<C>(rst, p, n0, m0, (uint32_t)(size_t)(__attribute__((address_space(3))) char*)LDS_PTR(smem), wave, lane);
  // One 8-wave workgroup per CU: spread the ring refill over the MFMA units (see the main loop).  With two
  // 4-wave workgroups per CU the other workgroup's MFMAs already cover the issue phase and the later
  // issue only shortens the time the loads have to land (measured 15-20 % slower), so those refill at
  // the top of the step.
  constexpr bool SPREAD = (S == 2 && C::NW == 8);
  // prologue: fill S-1 ring slots
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) ring_stage_step<C>(rst, (kb + s) * BK, s, wave_u);

  for (int t = 0; t < nk; ++t) {
    // K-step t has landed once at most min(S-2, nk-1-t) younger steps are still in flight
    const int rem = nk - 1 - t;
    if constexpr (S == 2) wait_vmcnt<0>();
    else if constexpr (S == 3) { if (rem >= 1) wait_vmcnt<G>(); else wait_vmcnt<0>(); }
    else { if (rem >= 2) wait_vmcnt<2 * G>(); else if (rem == 1) wait_vmcnt<G>(); else wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();            // every wave's loads of step t landed; step t-1 fully read
    __builtin_amdgcn_sched_barrier(0);
    if (t + S - 1 < nk)                      // refill the slot step t-1 used
      if constexpr (!SPREAD) ring_stage_step<C>(rst, (kb + t + S - 1) * BK, (t + S - 1) % S, wave_u);
    const char* tw = smem + (t % S) * C::STAGE_BYTES;
    const char* tx = tw + BN * C::ROWB;
    // Software-pipelined fragment stream.  Left alone, the compiler reads all 12 fragments of a 32-deep
    // sub-step, waits lgkmcnt(0), and only then issues its 32 MFMAs; the 8 waves of the block run in
    // lock-step behind the barrier, so the matrix pipes idle while 96 KiB of fragments cross the LDS
    // array (measured: 1.41 PFLOP/s-equivalent for the MFMA + LDS-read loop alone).  Here a "unit" is
    // one X fragment (16 rows) against the wave's four W fragments = 4 MFMAs; the X fragment of unit
    // u+2 and the W fragments of the next 32-deep sub-step are requested while unit u's MFMAs run.
    // The reads are inline asm with hand-counted s_waitcnt lgkmcnt(N): the compiler's own waitcnt
    // insertion falls back to lgkmcnt(0) here (the outstanding LDS-DMA loads count as "pending flat"
    // accesses), which would expose every prefetch again.  LDS returns in order, so waiting for X(u)
    // also covers every W fragment requested before it.  With SPREAD the refill of the other ring slot
    // is spread over the first G units, one LDS-DMA behind each unit's MFMAs, instead of 8 back-to-back
    // issues (and their address arithmetic) at the top of the step while the matrix pipes wait.
    {
      constexpr int KS = BK / 32, U = KS * MT, RB = C::ROWB;
      constexpr int WP0 = FragPipe<MT, KS>::WP0, WPN = FragPipe<MT, KS>::WPN;
      bf16x8 fw[2][4], fx[3];
      const uint32_t so = (uint32_t)((t % S) * C::STAGE_BYTES);
      uint32_t aw[KS], ax[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { aw[ks] = aw0[ks] + so; ax[ks] = ax0[ks] + so; }
      fw[0][0] = lds_read_b128<0 * 16 * RB>(aw[0]);
      fw[0][1] = lds_read_b128<1 * 16 * RB>(aw[0]);
      fw[0][2] = lds_read_b128<2 * 16 * RB>(aw[0]);
      fw[0][3] = lds_read_b128<3 * 16 * RB>(aw[0]);
      fx[0] = lds_read_b128<0>(ax[0]);
      fx[1] = lds_read_b128<16 * RB>(ax[0]);
      static_assert(FragPipe<MT, KS>::ok(), "fragment pipeline: a unit would start before its fragments are certain");
#define UNIMM_PREFETCH(u)                                                                                    \
        if constexpr ((u) + 2 < U) fx[((u) + 2) % 3] = lds_read_b128<(((u) + 2) % MT) * 16 * RB>(ax[((u) + 2) / MT]); \
        if constexpr (FragPipe<MT, KS>::npref_w(u) > 0) {                                                    \
          constexpr int w_ = (((u) % MT) - WP0) * WPN, kn_ = ((u) / MT + 1 < KS) ? (u) / MT + 1 : 0;         \
          fw[kn_ & 1][w_] = lds_read_b128<w_ * 16 * RB>(aw[kn_]);                                            \
          if constexpr (WPN == 2) fw[kn_ & 1][w_ + 1] = lds_read_b128<(w_ + 1) * 16 * RB>(aw[kn_]);          \
        }
#define UNIMM_UNIT(u)                                                                                        \
      if constexpr ((u) < U) {                                                                               \
        constexpr int ks_ = (u) / MT, j_ = (u) % MT;                                                         \
        if constexpr (!FragPipe<MT, KS>::AFTER) { UNIMM_PREFETCH(u) }                                        \
        lds_wait<FragPipe<MT, KS>::pending(u)>(fx[(u) % 3]);                                                 \
        _Pragma("unroll") for (int i = 0; i < 4; ++i)                                                        \
          acc[i][j_] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[ks_ & 1][i], fx[(u) % 3], acc[i][j_], 0, 0, 0); \
        if constexpr (FragPipe<MT, KS>::AFTER) { UNIMM_PREFETCH(u) }                                         \
        if constexpr (SPREAD && (u) < G) {                                         /* refill, one LDS-DMA per unit */ \
          if (t + 1 < nk) ring_stage_one<C>(rst, (kb + t + 1) * BK, (t + 1) & 1, wave_u, u);               \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
      }
      UNIMM_UNIT(0) UNIMM_UNIT(1) UNIMM_UNIT(2) UNIMM_UNIT(3) UNIMM_UNIT(4) UNIMM_UNIT(5) UNIMM_UNIT(6) UNIMM_UNIT(7)
      UNIMM_UNIT(8) UNIMM_UNIT(9) UNIMM_UNIT(10) UNIMM_UNIT(11) UNIMM_UNIT(12) UNIMM_UNIT(13) UNIMM_UNIT(14) UNIMM_UNIT(15)
#undef UNIMM_UNIT
#undef UNIMM_PREFETCH
    }
  }

  if (p.ksplit > 1) {
    if (!nt_split_join<C>(p, acc, smem, lid, split)) return;
  }
  nt_epilogue<C, EPI, OUT_F32>(p, acc, smem, m0, n0, wm, wn, wave, lane);
}

template <class C, int EPI, bool OUT_F32>
__global__ __launch_bounds__(C::THREADS, C::MIN_WAVES) void gemm_nt_kernel(GemmNtParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (EPI == UNIMM_EPI_BIAS_DROP_RESID) drop_resolve(p.drop);
  nt_tile<C, EPI, OUT_F32>(p, smem, xcd_remap(blockIdx.x, gridDim.x));
}

// Persistent form: one workgroup per CU slot walks tiles blockIdx, blockIdx + grid, ...  A tile's epilogue
// stores are fire-and-forget; here they drain under the NEXT tile's first ring stage instead of holding the
// workgroup (and its CU slot) until they complete and a new workgroup is launched.  grid % 8 == 0 keeps a
// workgroup's tiles on its own XCD's chunk of the tile order.
template <class C, int EPI, bool OUT_F32>
__global__ __launch_bounds__(C::THREADS, C::MIN_WAVES) void gemm_ntp_kernel(GemmNtParams p, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  if constexpr (EPI == UNIMM_EPI_BIAS_DROP_RESID) drop_resolve(p.drop);
  for (int lt = blockIdx.x; lt < ntiles; lt += gridDim.x) {
    if (lt != (int)blockIdx.x) __builtin_amdgcn_s_barrier();   // every wave has left its epilogue slab (it aliases the ring)
    nt_tile<C, EPI, OUT_F32>(p, smem, xcd_remap(lt, ntiles));
  }
}

// ------------------------------------------------------------------------------------------------
// gemm_tn: DW[N,K] += sum_m DY[m,N]^T X[m,K].  Both operands are reduction-strided, so tiles are
// staged [64 m][128 cols] (256-B rows, LDS-DMA, same source-side swizzle idea) and fragments come
// from ds_read_b64_tr_b16 (hardware transposed read): lane group g (16 lanes) reads 4 m-rows x 16
// columns; two reads give the 8 reduction elements of a 16x16x32 fragment.  The reduction index
// order inside a k-step is the same permutation for both operands, so the dot products are exact.
// Grid: tiles(N/128 x K/128) x splits over M; every split adds its fp32 partial with atomics
// (the gradient arena is zeroed once per step, so += is also what batch_multiply accumulation needs).
// ------------------------------------------------------------------------------------------------
constexpr int TK = 64;                       // m rows per step
constexpr int TN_TILE_BYTES = TK * 128 * 2;  // 16 KiB

struct GemmTnParams {
  const bf16_t* dy; const bf16_t* x; float* dw; float* dbias;
  const int32_t* m_dev;   // or NULL: reduction rows actually present (M is then a capacity)
  int M, N, K, lddy, ldx, lddw, rows_per_split;
  int tile0;   // first tile index of this problem inside a grouped launch
  int nsplit;  // splits of this problem that own rows (the others leave at once and never arrive at the tile's counter)
};
// A grouped launch: the weight gradients of one encoder block in ONE grid.  Every launch ends with a
// drain of one fp32 partial tile per resident workgroup (256 x 256 KiB = 67 MB of memory-side atomics,
// ~43 us, whatever the shape), so a block's 4-10 weight gradients pay that tail once instead of once each,
// and the bigger tile pool lets the split count land on a whole number of rounds.
constexpr int TN_MAXG = 48;   // 48 descriptors of 80 bytes + header = 3.8 KiB of kernel arguments (limit 4 KiB)
struct GemmTnGroup {
  int count, total_tiles;
  int splits;            // reduction ranges per tile
  float* slabs;          // workspace: [total_tiles][splits] partial tiles, or NULL = every split adds with fp32 atomics
  int* counters;         // workspace: arrival counter per tile (zero between launches: the last arriver resets it)
  GemmTnParams pr[TN_MAXG];
};

// [64 m][128 c] bf16 tile, 256-B rows = 16 chunks of 16 B.  A transposed read's 32-lane half touches
// 8 rows {b..b+3, b+8..b+11} x one aligned chunk pair; tn_swz moves each of those rows to its own
// chunk pair, so the half-wave covers all 64 banks exactly once.
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }

// Stage one m-step: the block's DY columns [n0, n0+TNB) and X columns [k0, k0+TKB) for rows
// [m0, m0+64), as [64][128]-column sub-tiles of 16 KiB (4 rows x 256 B per LDS-DMA wave-instruction).
template <int NW, int NSUB_A, int NSUB_B>
__device__ __forceinline__ void stage_one_tn(const GemmTnParams& p, int m0, int mend, int n0, int k0, char* stage,
                                             int wave, int lane, int r) {
  const int q = r * NW + wave;
  const int sub = q >> 4, rg = q & 15;
  const int row = rg * 4 + (lane >> 4);
  const int chunk = (lane & 15) ^ tn_swz(row);
  const bool is_a = sub < NSUB_A;
  const bf16_t* g = is_a ? p.dy : p.x;
  const int ld = is_a ? p.lddy : p.ldx;
  const int ncols = is_a ? p.N : p.K;
  const int c0 = is_a ? n0 + sub * 128 : k0 + (sub - NSUB_A) * 128;
  int gm = m0 + row;
  int gc = c0 + chunk * 8;
  // rows past the end re-read the last row (zeroed in LDS before use); columns past
  // round_up(ncols, 8) re-read the last readable chunk (they only feed outputs that are never stored)
  gm = gm < mend ? gm : mend - 1;
  const int cmax = ((ncols + 7) & ~7) - 8;
  gc = gc <= cmax ? gc : cmax;
  const bf16_t* src = g + (size_t)gm * ld + gc;
  __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(stage + sub * TN_TILE_BYTES + rg * 1024), 16, 0, 0);
}
template <int NW, int NSUB_A, int NSUB_B>
__device__ __forceinline__ void stage_step_tn(const GemmTnParams& p, int m0, int mend, int n0, int k0, char* stage,
                                              int wave, int lane) {
  constexpr int PER_WAVE = (NSUB_A + NSUB_B) * 16 / NW;
#pragma unroll
  for (int r = 0; r < PER_WAVE; ++r) stage_one_tn<NW, NSUB_A, NSUB_B>(p, m0, mend, n0, k0, stage, wave, lane, r);
}

// transposed fragment: 16 columns starting at c16 (tile-local, multiple of 16), reduction rows
// mrow0 + 8*(lane>>4) + {0..7}.  Lane i = 4q+p of its 16-lane group addresses row q, cols 4p..4p+3.
__device__ __forceinline__ bf16x8 read_frag_tr(const char* lds_tile, int c16, int mrow0, int lane) {
  const int g = lane >> 4, i = lane & 15, q = i >> 2, pp = i & 3;
  const int col = c16 + 4 * pp;  // element column
  s16x4 lo, hi;
  {
    const int row = mrow0 + 8 * g + q;
    const int chunk = (col >> 3) ^ tn_swz(row);
    lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)LDS_PTR(lds_tile + row * 256 + chunk * 16 + (col & 7) * 2));
  }
  {
    const int row = mrow0 + 8 * g + 4 + q;
    const int chunk = (col >> 3) ^ tn_swz(row);
    hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)LDS_PTR(lds_tile + row * 256 + chunk * 16 + (col & 7) * 2));
  }
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x8 v = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, v);
}

// Partial tile -> gradient (shared by both weight-gradient kernels).  acc[i][j]: n-subtile i, k-subtile j of the wave's
// (16 NT) x 64 piece of the tile at (n0, k0); D[n][k]: lane holds k = .. + (lane & 15) (column), n = .. + 4 (lane >> 4) + e.
template <int NW, int NT>
__device__ __forceinline__ void tn_store_partial(const GemmTnGroup& grp, const GemmTnParams& p, f32x4 (&acc)[NT][4], char* smem,
                                                 int gtile, int split, int n0, int k0, int wn, int wk, int wave, int lane, int tid) {
  // ---- partial tile -> gradient.  Without a workspace every split adds its fp32 partial with memory-side atomics:
  // one 256 KiB tile per resident workgroup at the END of every round, 67 MB at ~1.3 TB/s = ~43 us in which the chip
  // only drains (and 8x the algorithmic write traffic: 222 MB per text-block launch against 28 MB of gradients).
  // With a workspace the splits of a tile meet at a counter instead: each stores its partial to its own SLAB with plain,
  // fully coalesced 16-byte stores (register order: the reducer has the same layout), and whoever arrives LAST reads
  // the other slabs on top of the partial it still holds in registers and adds the sum into the gradient exactly once
  // (one fp32 atomic per element: a tied or concurrently accumulated dw stays safe).  Placement-independent hand-off (guide,
  // "in-launch split-K reduction"): slab stores -> every wave's vmcnt(0) -> workgroup barrier -> lane 0: agent-scope
  // release, vmcnt(0), relaxed agent-scope ticket; the last arriver: agent-scope acquire, vmcnt(0), barrier, plain loads.
  if (grp.slabs != nullptr && p.nsplit > 1) {
    constexpr int TILE_F4 = NW * NT * 4 * 64;                      // f32x4 per partial tile
    f32x4* mine = reinterpret_cast<f32x4*>(grp.slabs) + ((size_t)gtile * grp.splits + split) * TILE_F4;
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) mine[((wave * NT + i) * 4 + j) * 64 + lane] = acc[i][j];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                               // (the ring is dead: smem[0..3] carries the ticket)
    int* flag = reinterpret_cast<int*>(smem);
    if (tid == 0) {
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      const int ticket = __hip_atomic_fetch_add(grp.counters + gtile, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      if (ticket == p.nsplit - 1) {
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(grp.counters + gtile, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
      }
      *flag = ticket;
    }
    __syncthreads();
    if (*flag != p.nsplit - 1) return;                             // block-uniform
    const f32x4* base = reinterpret_cast<const f32x4*>(grp.slabs) + (size_t)gtile * grp.splits * TILE_F4;
    for (int sp = 0; sp < p.nsplit; ++sp) {
      if (sp == split) continue;
      const f32x4* other = base + (size_t)sp * TILE_F4;
#pragma unroll
      for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const f32x4 o = other[((wave * NT + i) * 4 + j) * 64 + lane];
          acc[i][j][0] += o[0]; acc[i][j][1] += o[1]; acc[i][j][2] += o[2]; acc[i][j][3] += o[3];
        }
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wn * 16 * NT + i * 16 + 4 * (lane >> 4) + e;
        if (n >= p.N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + wk * 64 + j * 16 + (lane & 15);
          if (k < p.K) atomicAdd(p.dw + (size_t)n * p.lddw + k, acc[i][j][e]);   // dw may alias another problem's / stream's
        }
      }
    }
    return;
  }
  // D[n][k]: lane holds k = .. + (lane&15) (column), n = .. + 4*(lane>>4) + e (rows)
#pragma unroll
  for (int i = 0; i < NT; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wn * 16 * NT + i * 16 + 4 * (lane >> 4) + e;
      if (n >= p.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (k < p.K) atomicAdd(p.dw + (size_t)n * p.lddw + k, acc[i][j][e]);
      }
    }
  }
}

// Problem that owns grouped tile index `tile`: the last descriptor with tile0 <= tile (binary search: <= 6 dependent
// scalar loads from the kernel-argument segment instead of `count` of them).
__device__ __forceinline__ int tn_find_problem(const GemmTnGroup& grp, int tile) {
  int lo = 0, hi = grp.count - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (grp.pr[mid].tile0 <= tile) lo = mid; else hi = mid - 1;
  }
  return lo;
}

// WN x WK waves; a wave owns (16*NT)(n) x 64(k) of the [TNB x TKB] output tile.
//   <2,2,4>: 128x128 tile, 4 waves, 64 KiB LDS (2 workgroups per CU)      -- small problems
//   <2,4,8>: 256x256 tile, 8 waves, 128 KiB LDS: half the staged bytes per MFMA, so one m-step of
//            compute covers twice the load latency (same reasoning as Cfg<2,4,8,64,2> of gemm_nt)
template <int WN, int WK, int NT>
__global__ __launch_bounds__(64 * WN * WK, (WN * WK * (NT == 4 ? 2 : 1) + 3) / 4) void gemm_tn_kernel(GemmTnGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = WN * WK, TNB = 16 * NT * WN, TKB = 64 * WK;
  constexpr int NSUB_A = TNB / 128, NSUB_B = TKB / 128;
  constexpr int STAGE_BYTES = (NSUB_A + NSUB_B) * TN_TILE_BYTES;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  // split-major order: all tiles of one reduction range run together and share its DY / X rows in L2
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / grp.total_tiles;
  int tile = lid - split * grp.total_tiles;
  int gi = tn_find_problem(grp, tile);
  gi = __builtin_amdgcn_readfirstlane(gi);   // wave-uniform: the descriptor is fetched once, into SGPRs
  const GemmTnParams p = grp.pr[gi];
  const int gtile = tile;                    // tile index inside the grouped launch (slab / counter index)
  tile -= p.tile0;
  const int nbk = (p.K + TKB - 1) / TKB;
  const int tn = tile / nbk, tk = tile - tn * nbk;
  const int n0 = tn * TNB, k0 = tk * TKB;
  const int mbeg = split * p.rows_per_split;
  int mend = mbeg + p.rows_per_split;
  const int mtot = p.m_dev != nullptr ? min(p.M, p.m_dev[0]) : p.M;
  mend = mend < mtot ? mend : mtot;
  if (mbeg >= mend) return;
  const int wn = wave / WK, wk = wave % WK;

  f32x4 acc[NT][4];  // [n-subtile i][k-subtile j]
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of DY, taken from the DY fragments that are in registers anyway: only in
  // the first k-tile column and only in the waves that own k-subtile 0 (one float per n-subtile per lane)
  const bool do_bias = p.dbias != nullptr && tk == 0 && wk == 0;
  float accb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) accb[i] = 0.f;

  // LDS byte address of this lane's transposed reads in ring slot 0 (see read_frag_tr): rows 8g+q,
  // element columns 4p..4p+3 of a 16-column fragment; the fragment's column offset c16 enters the 16-byte
  // chunk field by XOR with the row swizzle, so fragment x of a wave is (base ^ 32x) -- every other term
  // (sub-tile, 32-row sub-step, +4 rows, ring slot) is a disjoint bit field and goes into the immediate.
  uint32_t la, lb;
  {
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int row = 8 * g + q;
    const uint32_t lc = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)LDS_PTR(smem) +
                        row * 256 + ((tn_swz(row) | (pp >> 1)) << 4) + (pp & 1) * 8;
    const int ca = wn * 16 * NT, cb = wk * 64;
    la = (lc ^ (uint32_t)((ca & 127) * 2)) + (ca >> 7) * TN_TILE_BYTES;
    lb = (lc ^ (uint32_t)((cb & 127) * 2)) + (NSUB_A + (cb >> 7)) * TN_TILE_BYTES;
  }
  const int nsteps = (mend - mbeg + TK - 1) / TK;
  stage_step_tn<NW, NSUB_A, NSUB_B>(p, mbeg, mend, n0, k0, smem, wave, lane);

  for (int t = 0; t < nsteps; ++t) {
    const int cur = t & 1;
    const int mt = mbeg + t * TK;
    wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();          // step t landed everywhere; step t-1 fully read
    __builtin_amdgcn_sched_barrier(0);
    constexpr bool SPREAD = NW == 8;   // 8-wave tile: refill spread behind the units (as gemm_nt)
    constexpr int PER_WAVE = (NSUB_A + NSUB_B) * 16 / NW;
    if (!SPREAD && t + 1 < nsteps)
      stage_step_tn<NW, NSUB_A, NSUB_B>(p, mt + TK, mend, n0, k0, smem + (cur ^ 1) * STAGE_BYTES, wave, lane);
    const char* ta = smem + cur * STAGE_BYTES;
    const char* tb = ta + NSUB_A * TN_TILE_BYTES;
    const int valid = mend - mt;  // rows of this step that exist (>= 1); others must contribute 0
    if (valid < TK) {
      // Ragged last step of a split: the staged rows past the end are re-reads of the last row.  Zero them
      // on the DY side (whole 256-byte rows, so the chunk swizzle does not matter) and every product and
      // column sum they feed is zero.  Kept out of the MFMA path: a second code path over the 128
      // accumulator registers made the compiler spill them at the join.
      const int ninv = (TK - valid) * 16 * NSUB_A;          // 16-byte chunks to clear
      for (int idx = tid; idx < ninv; idx += 64 * NW) {
        const int sub = idx / ((TK - valid) * 16), rem = idx - sub * ((TK - valid) * 16);
        *reinterpret_cast<u32x4*>(const_cast<char*>(ta) + sub * TN_TILE_BYTES + (valid + (rem >> 4)) * 256 + (rem & 15) * 16) =
            u32x4{0u, 0u, 0u, 0u};
      }
      __syncthreads();
    }
    {
      // Software-pipelined fragment stream, the same scheme as gemm_nt_kernel's (inline-asm transposed
      // reads with hand-counted lgkmcnt; a "unit" = one DY fragment against the wave's four X fragments =
      // 4 MFMAs; the DY fragment of unit u+2 and the X fragments of the next 32-row sub-step are requested
      // while unit u computes).  Every fragment is two ds_read_b64_tr_b16.
      typedef __attribute__((ext_vector_type(8))) short s16x8;
      bf16x8 fb[2][4], fa[3];
      const uint32_t sa = la + (uint32_t)(cur * STAGE_BYTES), sb = lb + (uint32_t)(cur * STAGE_BYTES);
#define UNIMM_TR(base, x, ks) __builtin_bit_cast(bf16x8, __builtin_shufflevector(                                 \
          lds_read_tr<(ks) * 8192>((base) ^ (uint32_t)((x) * 32)), lds_read_tr<(ks) * 8192 + 1024>((base) ^ (uint32_t)((x) * 32)), \
          0, 1, 2, 3, 4, 5, 6, 7))
      fb[0][0] = UNIMM_TR(sb, 0, 0); fb[0][1] = UNIMM_TR(sb, 1, 0); fb[0][2] = UNIMM_TR(sb, 2, 0); fb[0][3] = UNIMM_TR(sb, 3, 0);
      fa[0] = UNIMM_TR(sa, 0, 0);
      fa[1] = UNIMM_TR(sa, 1, 0);
#define UNIMM_UNIT(u, WITH_BIAS)                                                                             \
      if constexpr ((u) < 2 * NT) {                                                                          \
        constexpr int ks_ = (u) / NT, i_ = (u) % NT;                                                         \
        if constexpr ((u) + 2 < 2 * NT) fa[((u) + 2) % 3] = UNIMM_TR(sa, ((u) + 2) % NT, ((u) + 2) / NT);    \
        if constexpr (FragPipe<NT, 2>::npref_w(u) > 0) {                                                     \
          constexpr int w_ = (i_ - FragPipe<NT, 2>::WP0) * FragPipe<NT, 2>::WPN;                             \
          fb[1][w_] = UNIMM_TR(sb, w_, 1);                                                                   \
          if constexpr (FragPipe<NT, 2>::WPN == 2) fb[1][w_ + 1] = UNIMM_TR(sb, w_ + 1, 1);                  \
        }                                                                                                    \
        lds_wait<2 * FragPipe<NT, 2>::pending(u)>(fa[(u) % 3]);                                              \
        _Pragma("unroll") for (int j = 0; j < 4; ++j)                                                        \
          acc[i_][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[(u) % 3], fb[ks_][j], acc[i_][j], 0, 0, 0); \
        if constexpr (WITH_BIAS)                  accb[i_] = dot_ones(fa[(u) % 3], accb[i_]);             \
        if constexpr (SPREAD && (u) < PER_WAVE) {                                                             \
          if (t + 1 < nsteps)                                                                                \
            stage_one_tn<NW, NSUB_A, NSUB_B>(p, mt + TK, mend, n0, k0, smem + (cur ^ 1) * STAGE_BYTES, wave, lane, u); \
        }                                                                                                    \
        __builtin_amdgcn_sched_barrier(0);                                                                   \
      }
#define UNIMM_UNITS(B)                                                                                       \
      UNIMM_UNIT(0, B) UNIMM_UNIT(1, B) UNIMM_UNIT(2, B) UNIMM_UNIT(3, B) UNIMM_UNIT(4, B) UNIMM_UNIT(5, B)      \
      UNIMM_UNIT(6, B) UNIMM_UNIT(7, B) UNIMM_UNIT(8, B) UNIMM_UNIT(9, B) UNIMM_UNIT(10, B) UNIMM_UNIT(11, B)    \
      UNIMM_UNIT(12, B) UNIMM_UNIT(13, B) UNIMM_UNIT(14, B) UNIMM_UNIT(15, B)
      // the column sums of DY (bias gradient) ride on the fragments of 1 wave in 4 of one tile column only; one
      // wave-uniform branch per step picks the unit stream with or without the 4 dependent v_dot2c per unit
      // (if-converted into every wave's stream they cost 19 % of the kernel)
      if (do_bias) { UNIMM_UNITS(true) } else { UNIMM_UNITS(false) }
#undef UNIMM_UNITS
#undef UNIMM_UNIT
#undef UNIMM_TR
    }
  }

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float v = accb[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wn * 16 * NT + i * 16 + (lane & 15);
      if (lane < 16 && n < p.N) atomicAdd(p.dbias + n, v);
    }
  }

  tn_store_partial<NW, NT>(grp, p, acc, smem, gtile, split, n0, k0, wn, wk, wave, lane, tid);
}

// ------------------------------------------------------------------------------------------------
// gemm_tn_pp: the 256x256 weight-gradient tile behind the ping-pong main loop of gemm_nt (nt_mainloop_pp above, same
// phases, same barrier / vmcnt proof with W <-> X columns "B", X <-> DY columns "A").
//
// Round-2 counters of the lock-step loop (profiles/r3_pmc_gemm_tn.txt): waves parked at s_waitcnt / s_barrier 38 % of
// their life, MFMA pipe busy 40 %, LDS array busy 15 %, 77 VALU + 44 SALU instructions per wave and 64-row step next to
// its 64 MFMAs -- two waves per SIMD then want ~2,000 issue cycles in a 2,048-cycle step: the loop was issue-bound and
// drained (vmcnt(0) + barrier + 12 dependent transposed reads) at the top of every step.  Here
//   * the two half-workgroups (n-halves of the tile) run one barrier apart: one issues nothing but its 16 MFMAs per
//     interval while the other requests fragments and issues its two LDS-DMA instructions;
//   * staging is buffer_load_dwordx4 ... lds through one buffer descriptor per operand whose num_records ends at the
//     split's last reduction row: the per-lane offset (row * ld + swizzled chunk) is constant over the steps, the step
//     advance is the scalar offset, and rows past the end come back as ZEROS from the range check (measured:
//     tools/exp/oob_lds_dma.hip) -- no address arithmetic, no row clamps and no ragged-tail code in the loop;
//   * one counted vmcnt per step certifies the next step's 64 KiB.
// LDS: two buffers of four 16 KiB half-tiles [X cols 0-127 | X cols 128-255 | DY cols 0-127 | DY cols 128-255] x 64 rows.
// ------------------------------------------------------------------------------------------------
struct TnPpStage {
  uint32_t so[4][2];   // per-lane source byte offsets inside the split's row range (row * ld + swizzled chunk)
  uint32_t lds;        // LDS byte address of this wave's first instruction slot in half-tile 0 of buffer 0 (uniform)
};

__device__ __forceinline__ void tn_pp_stage_init(TnPpStage& st, const GemmTnParams& p, int n0, int k0, uint32_t lds0, int wave, int lane) {
  const int inst0 = (wave >> 2) * 8 + (wave & 3) * 2;          // two of a half-tile's 16 wave-instructions (4 rows x 256 B each)
  st.lds = __builtin_amdgcn_readfirstlane(lds0 + inst0 * 1024);
#pragma unroll
  for (int ht = 0; ht < 4; ++ht)
#pragma unroll
    for (int r = 0; r < 2; ++r) {
      const int row = (inst0 + r) * 4 + (lane >> 4);           // reduction row inside the 64-row step
      const int chunk = (lane & 15) ^ tn_swz(row);
      const bool is_b = ht < 2;
      const int ncols = is_b ? p.K : p.N, ld = is_b ? p.ldx : p.lddy;
      int gc = (is_b ? k0 : n0) + (ht & 1) * 128 + chunk * 8;
      const int cmax = ((ncols + 7) & ~7) - 8;                  // columns past round_up(ncols, 8) re-read the last readable chunk
      gc = gc <= cmax ? gc : cmax;                              // (they only feed outputs that are never stored)
      st.so[ht][r] = (uint32_t)row * (uint32_t)(ld * 2) + (uint32_t)gc * 2u;
    }
}

// Issued through inline asm (see pp_stage): the compiler must not see LDS-DMA in flight.  soff = step * 64 rows in bytes.
__device__ __forceinline__ void tn_pp_stage(const TnPpStage& st, const u32x4& srd, uint32_t soff, int buf, int ht) {
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    const uint32_t dst = st.lds + buf * 65536 + ht * 16384 + r * 1024;
    uint32_t keep;                                             // m0 is the compiler's: hand it back as found
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(st.so[ht][r]), "s"(dst), "s"(srd), "s"(soff) : "memory");
  }
}

__global__ __launch_bounds__(512, 2) void gemm_tn_pp_kernel(GemmTnGroup grp) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int NW = 8, NT = 8, TNB = 256, TKB = 256;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lid = xcd_remap(blockIdx.x, gridDim.x);            // split-major order, as gemm_tn_kernel
  const int split = lid / grp.total_tiles;
  int tile = lid - split * grp.total_tiles;
  int gi = tn_find_problem(grp, tile);
  gi = __builtin_amdgcn_readfirstlane(gi);
  const GemmTnParams p = grp.pr[gi];
  const int gtile = tile;
  tile -= p.tile0;
  const int nbk = (p.K + TKB - 1) / TKB;
  const int tn = tile / nbk, tk = tile - tn * nbk;
  const int n0 = tn * TNB, k0 = tk * TKB;
  const int mbeg = split * p.rows_per_split;
  int mend = mbeg + p.rows_per_split;
  const int mtot = p.m_dev != nullptr ? min(p.M, p.m_dev[0]) : p.M;
  mend = mend < mtot ? mend : mtot;
  if (mbeg >= mend) return;
  const int wn = wave >> 2, wk = wave & 3;                      // G0 = waves 0-3 (n-half 0), G1 = waves 4-7 (n-half 1)
  const int nk = (mend - mbeg + TK - 1) / TK;

  f32x4 acc[NT][4];
#pragma unroll
  for (int i = 0; i < NT; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_bias = p.dbias != nullptr && tk == 0 && wk == 0;   // wave-uniform (wave comes from readfirstlane)
  float accb[NT];
#pragma unroll
  for (int i = 0; i < NT; ++i) accb[i] = 0.f;

  // buffer descriptors: base = first reduction row of this split, num_records = the split's bytes (range check = zero fill)
  u32x4 srd_a, srd_b;
  uint32_t step_a, step_b;
  {
    const uint64_t ba = (uint64_t)(uintptr_t)(p.dy + (size_t)mbeg * p.lddy), bb = (uint64_t)(uintptr_t)(p.x + (size_t)mbeg * p.ldx);
    const uint32_t rows = (uint32_t)(mend - mbeg);
    srd_a = u32x4{(uint32_t)ba, (uint32_t)(ba >> 32) & 0xffffu, rows * (uint32_t)p.lddy * 2u, 0x00020000u};
    srd_b = u32x4{(uint32_t)bb, (uint32_t)(bb >> 32) & 0xffffu, rows * (uint32_t)p.ldx * 2u, 0x00020000u};
    step_a = (uint32_t)(TK * p.lddy * 2);
    step_b = (uint32_t)(TK * p.ldx * 2);
  }
  TnPpStage st;
  uint32_t aa0[8], ab0[4];                                      // transposed-read addresses of the wave's fragments, buffer 0
  {
    const uint32_t lds0 = (uint32_t)(size_t)(__attribute__((address_space(3))) char*)LDS_PTR(smem);
    tn_pp_stage_init(st, p, n0, k0, lds0, wave, lane);
    const int g = lane >> 4, q = (lane & 15) >> 2, pp = lane & 3;
    const int row = 8 * g + q;
    const uint32_t lc = lds0 + row * 256 + ((tn_swz(row) | (pp >> 1)) << 4) + (pp & 1) * 8;
    const uint32_t la = lc + (2 + wn) * 16384;                  // DY half-tile of this wave's n-half
    const uint32_t lb = (lc ^ (uint32_t)((wk & 1) * 128)) + (wk >> 1) * 16384;
#pragma unroll
    for (int x = 0; x < 8; ++x) aa0[x] = la ^ (uint32_t)(x * 32);
#pragma unroll
    for (int x = 0; x < 4; ++x) ab0[x] = lb ^ (uint32_t)(x * 32);
  }
#define UNIMM_TN_STAGE(KT, HT)                                                                                  \
  {                                                                                                             \
    if ((HT) < 2) tn_pp_stage(st, srd_b, (uint32_t)(KT) * step_b, (KT) & 1, (HT));                              \
    else tn_pp_stage(st, srd_a, (uint32_t)(KT) * step_a, (KT) & 1, (HT));                                       \
  }
  // prologue: step 0 (4 half-tiles) and the two X half-tiles of step 1
  UNIMM_TN_STAGE(0, 0) UNIMM_TN_STAGE(0, 1) UNIMM_TN_STAGE(0, 2) UNIMM_TN_STAGE(0, 3)
  if (nk > 1) { UNIMM_TN_STAGE(1, 0) UNIMM_TN_STAGE(1, 1) }
  __builtin_amdgcn_s_waitcnt(0x0F70);                           // vmcnt(0), visible to the compiler (see nt_mainloop_pp)
  __builtin_amdgcn_s_barrier();                                 // step 0 has landed for every wave
  if (wn == 1) __builtin_amdgcn_s_barrier();                    // G1 runs one barrier behind G0
  __builtin_amdgcn_sched_barrier(0);

  bf16x8 fa[2][4], fb[2][4];
#define UNIMM_TN_RD(addr, ks) __builtin_bit_cast(bf16x8, __builtin_shufflevector(                               \
      lds_read_tr<(ks) * 8192>(addr), lds_read_tr<(ks) * 8192 + 1024>(addr), 0, 1, 2, 3, 4, 5, 6, 7))
#define UNIMM_TN_READ_A(JH)                                                                                     \
  {                                                                                                             \
    fa[0][0] = UNIMM_TN_RD(aa[4 * (JH) + 0], 0); fa[0][1] = UNIMM_TN_RD(aa[4 * (JH) + 1], 0);                   \
    fa[0][2] = UNIMM_TN_RD(aa[4 * (JH) + 2], 0); fa[0][3] = UNIMM_TN_RD(aa[4 * (JH) + 3], 0);                   \
    fa[1][0] = UNIMM_TN_RD(aa[4 * (JH) + 0], 1); fa[1][1] = UNIMM_TN_RD(aa[4 * (JH) + 1], 1);                   \
    fa[1][2] = UNIMM_TN_RD(aa[4 * (JH) + 2], 1); fa[1][3] = UNIMM_TN_RD(aa[4 * (JH) + 3], 1);                   \
  }
#define UNIMM_TN_READ_B(IH)                                                                                     \
  {                                                                                                             \
    fb[0][2 * (IH)] = UNIMM_TN_RD(ab[2 * (IH)], 0); fb[0][2 * (IH) + 1] = UNIMM_TN_RD(ab[2 * (IH) + 1], 0);     \
    fb[1][2 * (IH)] = UNIMM_TN_RD(ab[2 * (IH)], 1); fb[1][2 * (IH) + 1] = UNIMM_TN_RD(ab[2 * (IH) + 1], 1);     \
  }
#define UNIMM_TN_SYNC_READS()                                                                                   \
  __builtin_amdgcn_s_barrier();                                                                                 \
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(fa[0][0]), "+v"(fa[0][1]), "+v"(fa[0][2]), "+v"(fa[0][3]), "+v"(fa[1][0]), \
               "+v"(fa[1][1]), "+v"(fa[1][2]), "+v"(fa[1][3]), "+v"(fb[0][0]), "+v"(fb[0][1]), "+v"(fb[0][2]), "+v"(fb[0][3]), \
               "+v"(fb[1][0]), "+v"(fb[1][1]), "+v"(fb[1][2]), "+v"(fb[1][3]));                                 \
  __builtin_amdgcn_sched_barrier(0);
#define UNIMM_TN_MFMA(JH, IH)                                                                                   \
  {                                                                                                             \
    __builtin_amdgcn_s_setprio(1);                                                                              \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj)                                                          \
        _Pragma("unroll") for (int ii = 0; ii < 2; ++ii)                                                        \
          acc[4 * (JH) + jj][2 * (IH) + ii] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(                          \
              fa[ks][jj], fb[ks][2 * (IH) + ii], acc[4 * (JH) + jj][2 * (IH) + ii], 0, 0, 0);                   \
    __builtin_amdgcn_s_setprio(0);                                                                              \
  }
  // bias gradient = column sums of DY: on the fragments of ONE wave in four of one tile column, behind its MFMAs
#define UNIMM_TN_BIAS(JH)                                                                                       \
  if (do_bias) {                                                                                                \
    _Pragma("unroll") for (int ks = 0; ks < 2; ++ks)                                                            \
      _Pragma("unroll") for (int jj = 0; jj < 4; ++jj) accb[4 * (JH) + jj] = dot_ones(fa[ks][jj], accb[4 * (JH) + jj]); \
  }

  for (int t = 0; t < nk; ++t) {
    const uint32_t off = (uint32_t)((t & 1) * 65536);
    uint32_t aa[8], ab[4];
#pragma unroll
    for (int x = 0; x < 8; ++x) aa[x] = aa0[x] + off;
#pragma unroll
    for (int x = 0; x < 4; ++x) ab[x] = ab0[x] + off;
    // Staging runs as far ahead as the two buffers allow: the X half-tiles of step t+2 go out in phase 3 of step t (their
    // buffer's X fragments were last read in phase 1), the DY half-tiles of step t+1 in phases 0 and 1 (last read in
    // phase 2 of step t-1); a half-tile has 0.5-1.25 steps to land (0.25-1.0 with one half-tile per phase).
    // ---- phase 0 (stages DY half-tile 0 of step t+1)
    UNIMM_TN_READ_B(0)
    UNIMM_TN_READ_A(0)
    if (t + 1 < nk) UNIMM_TN_STAGE(t + 1, 2)
    UNIMM_TN_SYNC_READS()
    UNIMM_TN_MFMA(0, 0)
    UNIMM_TN_BIAS(0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 1
    UNIMM_TN_READ_B(1)
    if (t + 1 < nk) UNIMM_TN_STAGE(t + 1, 3)
    UNIMM_TN_SYNC_READS()
    UNIMM_TN_MFMA(0, 1)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 2
    UNIMM_TN_READ_A(1)
    UNIMM_TN_SYNC_READS()
    UNIMM_TN_MFMA(1, 1)
    UNIMM_TN_BIAS(1)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    // ---- phase 3: no fragment reads; certify step t+1 (every DMA older than this phase's four has landed)
    if (t + 2 < nk) { UNIMM_TN_STAGE(t + 2, 0) UNIMM_TN_STAGE(t + 2, 1) wait_vmcnt<4>(); } else wait_vmcnt<0>();
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    UNIMM_TN_MFMA(1, 0)
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
  }
  if (wn == 0) __builtin_amdgcn_s_barrier();                    // G0 waits for G1's last phase (barrier counts match again)
#undef UNIMM_TN_STAGE
#undef UNIMM_TN_RD
#undef UNIMM_TN_READ_A
#undef UNIMM_TN_READ_B
#undef UNIMM_TN_SYNC_READS
#undef UNIMM_TN_MFMA
#undef UNIMM_TN_BIAS

  if (do_bias) {
#pragma unroll
    for (int i = 0; i < NT; ++i) {
      float v = accb[i];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wn * 16 * NT + i * 16 + (lane & 15);
      if (lane < 16 && n < p.N) atomicAdd(p.dbias + n, v);
    }
  }
  tn_store_partial<NW, NT>(grp, p, acc, smem, gtile, split, n0, k0, wn, wk, wave, lane, tid);
}

// ------------------------------------------------------------------------------------------------
// Launch profiler (bench.py's `roofline` block): when enabled, every GEMM launch is bracketed by HIP
// events on the launch stream; unimm_prof_collect() sums elapsed time and algorithmic FLOPs per kernel
// variant.  Off by default (no events, no overhead).
// ------------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int variant; int tag; double flops; };
constexpr int PROF_MAX = 1 << 16;
// One variant per kernel SYMBOL (what rocprofv3 --stats lists): gemm_nt / gemm_ntp x tile x epilogue x output type, and
// the three weight-gradient kernels.  NT: ((persistent * 16 + tile code) * 8 + epilogue) * 2 + out_f32 (tile code = the
// unimm_gemm_nt_args.tile code of the configuration: 1, 3, 6, 7, 8); TN: 512 = gemm_tn_pp, 513 = gemm_tn<2,4,8> (the
// lock-step loop, tools only), 514 = gemm_tn<2,2,4>.
constexpr int PROF_VARIANTS = 516;
constexpr int PROF_TN0 = 512;
template <class C> constexpr int nt_tile_code() {
  return C::PP ? 8 : (C::MT == 8 ? 3 : (C::MT == 6 ? 6 : (C::MT == 2 ? (C::STAGES == 3 ? 9 : 7) : (C::STAGES == 3 ? 10 : 1))));
}
bool g_prof_on = false;
bool g_prof_tn_only = false;        // unimm_prof_enable(2): only the weight-gradient launches (2 event records per launch
                                    // are host time; a rank whose step is launch-rate-bound should not pay them 340 times)
ProfRec* g_prof = nullptr;
int g_prof_n = 0;
// Caller-side tag of the launches that follow (unimm_prof_tag): bench.py's `roofline.coattention_gemms` = the GEMM launches
// the engine issues from inside a BertConnectionLayer (models/vilbert_dialog.py:655-783), forward and backward.
constexpr int PROF_TAGS = 8;
int g_prof_tag = 0;
double g_tag_ms[PROF_TAGS], g_tag_flops[PROF_TAGS], g_tag_union_ms[PROF_TAGS];
int g_tag_count[PROF_TAGS];

inline ProfRec* prof_begin(int variant, double flops, hipStream_t s) {
  if (!g_prof_on || g_prof_n >= PROF_MAX || (g_prof_tn_only && variant < PROF_TN0)) return nullptr;
  ProfRec* r = &g_prof[g_prof_n];
  if (r->a == nullptr) {
    if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) return nullptr;
  }
  r->variant = variant; r->flops = flops; r->tag = g_prof_tag;
  hipEventRecord(r->a, s);
  ++g_prof_n;
  return r;
}
inline void prof_end(ProfRec* r, hipStream_t s) { if (r) hipEventRecord(r->b, s); }

// Per-call tuning of unimm_gemm_nt (unimm_gemm_nt_args.tile = gn * 1000 + p * 100 + cfg; 0 = everything automatic).
struct NtTune { int cfg, persist, gn; };
inline bool nt_tune_decode(int code, NtTune& t) {
  if (code < 0 || code > 999 * 1000 + 999) return false;
  const int pc = (code % 1000) / 100;
  t.persist = pc == 0 ? -1 : (pc == 1 ? 1 : 0);           // x1xx persistent, x2xx one workgroup per tile, else automatic
  t.cfg = code % 100;
  t.gn = code / 1000;
  return pc <= 2 && (t.cfg == 0 || t.cfg == 1 || t.cfg == 3 || t.cfg == 6 || t.cfg == 7 || t.cfg == 8 || t.cfg == 9 || t.cfg == 10);
}

template <class C, int EPI, bool F32> constexpr auto pick_nt_kernel() { return &gemm_nt_kernel<C, EPI, F32>; }

inline int cu_count() {
  static int n = 0;
  if (n == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n = prop.multiProcessorCount;
    if (n <= 0) n = 256;
    if (n < 8) n = 8;     // the slot arithmetic below works in multiples of the 8 XCDs
  }
  return n;
}

struct NtSplit { int want; void* ws; long ws_bytes; };     // want: 0 / 1 = off, >= 2 = that many splits, -1 = the library's choice

template <class C, int EPI>
int launch_nt_cfg(const GemmNtParams& p_in, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk = NtSplit{0, nullptr, 0}) {
  GemmNtParams p = p_in;
  int nwg = ((p.M + C::BM - 1) / C::BM) * ((p.N + C::BN - 1) / C::BN);
  p.ksplit = 1; p.slabs = nullptr; p.counters = nullptr;
  if (!C::PP && sk.want != 0 && sk.want != 1 && sk.ws != nullptr) {
    // Split only grids that leave the chip under-filled (every split workgroup resident at once) and reductions long enough
    // to pay for the join (>= 8 K-steps per split).
    const int slots = (cu_count() & ~7) * C::WG_PER_CU, nk = p.K / C::BK;
    int ks = sk.want > 1 ? sk.want : (nwg > 0 ? slots / nwg : 1);
    ks = ks > 4 ? 4 : ks;
    while (ks > 1 && nk / ks < 8) --ks;
    if (ks > 1) {
      const int per = (nk + ks - 1) / ks;
      while (ks > 1 && (ks - 1) * per >= nk) --ks;                 // no empty split
    }
    constexpr long CBYTES = 16384, TILE_BYTES = (long)C::NW * 4 * C::MT * 64 * 16;
    if (ks > 1 && nwg <= CBYTES / 4 && CBYTES + (long)nwg * ks * TILE_BYTES <= sk.ws_bytes) {
      p.ksplit = ks;
      p.counters = reinterpret_cast<int*>(sk.ws);
      p.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(sk.ws) + CBYTES);
      nwg *= ks;
      want_persist = 0;
    }
  }
  auto k32 = pick_nt_kernel<C, EPI, true>();
  auto k16 = pick_nt_kernel<C, EPI, false>();
  if (C::LDS > 64 * 1024) {
    static bool done32 = false, done16 = false;
    bool& done = out_f32 ? done32 : done16;
    if (!done) {
      const void* fn = out_f32 ? (const void*)k32 : (const void*)k16;
      if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return UNIMM_E_HIP;
      done = true;
    }
  }
  bool persist = false;
  {
    const int slots = (cu_count() & ~7) * C::WG_PER_CU;
    persist = want_persist != 0 && nwg > slots && slots > 0;
    if (persist) {
      auto p32 = gemm_ntp_kernel<C, EPI, true>;
      auto p16 = gemm_ntp_kernel<C, EPI, false>;
      if (C::LDS > 64 * 1024) {
        static bool pdone32 = false, pdone16 = false;
        bool& pdone = out_f32 ? pdone32 : pdone16;
        if (!pdone) {
          const void* fn = out_f32 ? (const void*)p32 : (const void*)p16;
          if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, C::LDS) != hipSuccess) return UNIMM_E_HIP;
          pdone = true;
        }
      }
      ProfRec* pr = prof_begin(((16 + nt_tile_code<C>()) * 8 + EPI) * 2 + (out_f32 ? 1 : 0), 2.0 * p.M * (double)p.N * p.K, s);
      if (out_f32) hipLaunchKernelGGL(p32, dim3(slots), dim3(C::THREADS), C::LDS, s, p, nwg);
      else hipLaunchKernelGGL(p16, dim3(slots), dim3(C::THREADS), C::LDS, s, p, nwg);
      prof_end(pr, s);
      UNIMM_CHECK_LAUNCH();
      return UNIMM_OK;
    }
  }
  ProfRec* pr = prof_begin((nt_tile_code<C>() * 8 + EPI) * 2 + (out_f32 ? 1 : 0), 2.0 * p.M * (double)p.N * p.K, s);
  if (out_f32) hipLaunchKernelGGL(k32, dim3(nwg), dim3(C::THREADS), C::LDS, s, p);
  else hipLaunchKernelGGL(k16, dim3(nwg), dim3(C::THREADS), C::LDS, s, p);
  prof_end(pr, s);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

template <int EPI>
int launch_nt(const GemmNtParams& p, bool out_f32, const NtTune& tune, hipStream_t s, const NtSplit& sk = NtSplit{0, nullptr, 0}) {
  int cfg = tune.cfg;
  const int wp = tune.persist;
  if (cfg == 0) {
    // Tile choice = the configuration with the smallest modelled time: rounds of `slots` workgroups, a round of a tile
    // with W workgroups per CU costs area x W / eff, the last (partial) round only the workgroups per CU it really
    // has.  eff (per-flop efficiency against the 256x256 tile, from interleaved micro-benchmarks at K = 768..3072):
    // 192x256 0.95, 128x128 0.85.  At ~31k rows: N = 3072 -> 256x256 (5.72 rounds), N = 2304 / 768 -> 192x256 (5.73 /
    // 1.91 rounds instead of 4.29 / 1.43; 112 vs 116 us, 135 vs 151 us and vs 141 us for 128x128); the image side at
    // 8,880 rows: N = 1024 -> 192x256 (30.6 vs 37.3 us), N = 3072 -> 256x256 (58.5 vs 65.9 us).  Grids too small to
    // give every CU 1.5 workgroups of 128x128 take the 64x128 tile (three workgroups per CU: per-GPU batches of 30).
    const int cus = cu_count() & ~7;
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (t128 < 384) {
      cfg = 7;
    } else {
      struct Cand { int cfg, bm, bn, w; double eff; };
      // 256x256 = the ping-pong loop (configuration 8; 3 = the lock-step ring, reachable through unimm_gemm_set_tile for A/B runs)
      const Cand cand[3] = {{8, 256, 256, 1, 1.0}, {6, 192, 256, 1, 0.95}, {1, 128, 128, 2, 0.85}};
      double best = 1e30;
      for (const Cand& c : cand) {
        const long tiles = (long)((p.M + c.bm - 1) / c.bm) * ((p.N + c.bn - 1) / c.bn);
        const long slots = (long)cus * c.w;
        const double area = (double)c.bm * c.bn / 65536.0;
        const long rem = tiles % slots;
        const double cost = ((double)(tiles / slots) * c.w + (rem ? (double)((rem + cus - 1) / cus) : 0.0)) * area / c.eff;
        if (cost < best) { best = cost; cfg = c.cfg; }
      }
    }
  }
  // The ping-pong loop addresses its LDS-DMA sources with a 32-bit byte offset per lane on the un-offset operand base
  // (pp_stage_init): past 4 GiB of operand rows that offset would wrap, so such problems take the lock-step ring (size_t math).
  if (cfg == 8 && ((size_t)p.M * (size_t)p.ldx * 2 >= ((size_t)1 << 32) || (size_t)p.N * (size_t)p.ldw * 2 >= ((size_t)1 << 32))) cfg = 3;
  if (cfg == 8) return launch_nt_cfg<Cfg<2, 4, 8, 64, 2, 1>, EPI>(p, out_f32, wp, s);
  // (configurations 2, 4, 5 - the BK = 32 rings of 4 and 5 slots, measured slower in round 1 - are not instantiated:
  //  a third of this file's compile time; unimm_gemm_nt rejects them, DESIGN.md 5 has the numbers)
  if (cfg == 6) return launch_nt_cfg<Cfg<2, 4, 6, 64, 2>, EPI>(p, out_f32, wp, s);
  if (cfg == 7) return launch_nt_cfg<Cfg<2, 2, 2, 64, 2>, EPI>(p, out_f32, wp, s, sk);
  // 3-slot rings for the small tiles (two K-steps in flight): grids that leave the chip under-filled run one or two
  // workgroups per CU whose K loop is a chain of exposed L2 round trips (0.69 us per 64-deep step at 3.9k rows, 15 % of a
  // CU's MFMA rate); 9 = 64x128 (72 KiB: two workgroups per CU), 10 = 128x128 (96 KiB: one per CU)
  if (cfg == 9) return launch_nt_cfg<Cfg<2, 2, 2, 64, 3>, EPI>(p, out_f32, wp, s, sk);
  if (cfg == 10) return launch_nt_cfg<Cfg<2, 2, 4, 64, 3>, EPI>(p, out_f32, wp, s, sk);
  if (cfg == 3) return launch_nt_cfg<Cfg<2, 4, 8, 64, 2>, EPI>(p, out_f32, wp, s);
  return launch_nt_cfg<Cfg<2, 2, 4, 64, 2>, EPI>(p, out_f32, wp, s, sk);
}

}  // namespace

extern "C" int unimm_gemm_nt(const unimm_gemm_nt_args* a, void* stream) {
  if (a == nullptr || a->x == nullptr || a->w == nullptr || a->out == nullptr) return UNIMM_E_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % 64) != 0) return UNIMM_E_SHAPE;
  if ((a->ldx % 8) || (a->ldw % 8) || a->ldx < a->K || a->ldw < a->K || a->ldo < a->N) return UNIMM_E_ALIGN;
  if (((uintptr_t)a->x | (uintptr_t)a->w | (uintptr_t)a->out) & 15) return UNIMM_E_ALIGN;
  if (a->out_f32 ? (a->ldo % 4) : (a->ldo % 4)) return UNIMM_E_ALIGN;
  const bool needs_aux = a->epilogue == UNIMM_EPI_BIAS_DROP_RESID || a->epilogue == UNIMM_EPI_DGELU ||
                         a->epilogue == UNIMM_EPI_ADD || a->epilogue == UNIMM_EPI_MUL;
  if (needs_aux && (a->aux == nullptr || (a->ldaux % 4) || a->ldaux < a->N)) return UNIMM_E_ARG;
  if (a->epilogue == UNIMM_EPI_BIAS_DROP_RESID && !a->out_f32) return UNIMM_E_ARG;  // residual stream is fp32
  GemmNtParams p;
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = a->bias; p.aux = a->aux;
  p.out = a->out; p.out2 = (bf16_t*)a->out2;
  p.M = a->M; p.N = a->N; p.K = a->K; p.ldx = a->ldx; p.ldw = a->ldw; p.ldaux = a->ldaux; p.ldo = a->ldo;
  p.drop.key = a->drop_key; p.drop.thr = a->drop_thr; p.drop.scale = a->drop_scale; p.drop.salt = a->drop_salt;
  const int n_ln = (a->aux_mean != nullptr) + (a->aux_rstd != nullptr) + (a->aux_gamma != nullptr) + (a->aux_beta != nullptr);
  if (n_ln != 0 && (n_ln != 4 || a->epilogue != UNIMM_EPI_BIAS_DROP_RESID)) return UNIMM_E_ARG;
  p.aux_mean = a->aux_mean; p.aux_rstd = a->aux_rstd; p.aux_gamma = a->aux_gamma; p.aux_beta = a->aux_beta;
  NtTune tune;
  if (!nt_tune_decode(a->tile, tune)) return UNIMM_E_ARG;
  p.gn = tune.gn > 0 ? tune.gn : 4;   // 4 tile columns per group: +1 % at 240 sequences, +5-8 % at 30 over 6 (A/B, two-stream schedule)
  hipStream_t s = (hipStream_t)stream;
  const bool f32 = a->out_f32 != 0;
  p.ksplit = 1; p.slabs = nullptr; p.counters = nullptr;
  if (a->splitk != 0 && a->splitk != 1 && (a->splitk_ws == nullptr || a->splitk_ws_bytes < 32768 || ((uintptr_t)a->splitk_ws & 255) ||
                                           a->splitk > 8 || a->splitk < -1))
    return UNIMM_E_ARG;
  const NtSplit sk{a->splitk, a->splitk_ws, (long)a->splitk_ws_bytes};
  switch (a->epilogue) {
    case UNIMM_EPI_BIAS: return launch_nt<UNIMM_EPI_BIAS>(p, f32, tune, s, sk);
    case UNIMM_EPI_BIAS_GELU: return launch_nt<UNIMM_EPI_BIAS_GELU>(p, f32, tune, s, sk);
    case UNIMM_EPI_BIAS_DROP_RESID: return launch_nt<UNIMM_EPI_BIAS_DROP_RESID>(p, f32, tune, s, sk);
    case UNIMM_EPI_BIAS_RELU: return launch_nt<UNIMM_EPI_BIAS_RELU>(p, f32, tune, s, sk);
    case UNIMM_EPI_DGELU: return launch_nt<UNIMM_EPI_DGELU>(p, f32, tune, s, sk);
    case UNIMM_EPI_ADD: return launch_nt<UNIMM_EPI_ADD>(p, f32, tune, s, sk);
    case UNIMM_EPI_MUL: return launch_nt<UNIMM_EPI_MUL>(p, f32, tune, s, sk);
    case UNIMM_EPI_BIAS_GELU_DG: return launch_nt<UNIMM_EPI_BIAS_GELU_DG>(p, f32, tune, s, sk);
    default: return UNIMM_E_ARG;
  }
}

namespace {

int check_tn(const unimm_gemm_tn_args* a) {
  if (a->dy == nullptr || a->x == nullptr || a->dw == nullptr) return UNIMM_E_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return UNIMM_E_SHAPE;
  if ((a->lddy % 8) || (a->ldx % 8) || a->lddy < a->N || a->ldx < a->K || a->lddw < a->K) return UNIMM_E_ALIGN;
  if (((uintptr_t)a->dy | (uintptr_t)a->x) & 15) return UNIMM_E_ALIGN;
  return UNIMM_OK;
}

// (M >= 1024: with launches grouped over several blocks the 256x256 ping-pong tile also wins at the ~4k rows of a 30-sequence
// batch -- the 128x128 class it used to take there was chosen when one block's ~110 tiles had to fill the chip)
inline bool tn_is_big(const unimm_gemm_tn_args* a) { return a->N >= 256 && a->K >= 256 && a->M >= 1024; }

// One launch of `count` (<= TN_MAXG) problems that all use the same tile size.
int launch_tn_group(const unimm_gemm_tn_args* const* a, int count, bool big, bool shared, bool legacy_loop, void* ws,
                    int64_t ws_bytes, hipStream_t s) {
  GemmTnGroup g;
  const int tb = big ? 256 : 128;
  int tiles = 0, max_m = 0;
  double flops = 0.0;
  for (int i = 0; i < count; ++i) {
    GemmTnParams& p = g.pr[i];
    p.dy = (const bf16_t*)a[i]->dy; p.x = (const bf16_t*)a[i]->x; p.dw = a[i]->dw; p.dbias = a[i]->dbias;
    p.m_dev = a[i]->m_dev;
    p.M = a[i]->M; p.N = a[i]->N; p.K = a[i]->K; p.lddy = a[i]->lddy; p.ldx = a[i]->ldx; p.lddw = a[i]->lddw;
    p.tile0 = tiles;
    tiles += ((p.N + tb - 1) / tb) * ((p.K + tb - 1) / tb);
    max_m = p.M > max_m ? p.M : max_m;
    flops += 2.0 * p.M * (double)p.N * p.K;
  }
  g.count = count; g.total_tiles = tiles;
  // Split the reduction (M) `splits` ways, the same for every problem of the group.  Workgroups run in
  // rounds of `slots` (256 CUs x 1 or 2 resident); cost ~ rounds(s) / s, so pick the s that wastes the
  // least of its last round (270 workgroups ran 2x as long as 243), smallest s on ties (fewer partial
  // tiles to drain), keeping >= 1024 reduction rows per split of the longest problem.
  const int slots = big ? 256 : 512;
  int max_s = max_m / 1024;
  max_s = max_s < 1 ? 1 : (max_s > 32 ? 32 : max_s);
  int splits = 1;
  double best = 1e30;
  // cost of a split count: rounds x (steps of one workgroup + its fixed cost: pipeline fill and partial-tile
  // drain, ~8 steps' worth; 40 when the caller says the launch shares the chip with another stream's kernels
  // (the shared_chip argument of unimm_gemm_tn_grouped_ws): an under-filled round is then not idle time, so fewer, longer workgroups and fewer
  // partial tiles win: 51.5 -> 50.9 ms per step at 240 sequences, neutral at 30-120; alone on the chip it costs 10 %); the fixed term only matters for short reductions (per-GPU batches of 30-60
  // sequences under strong scaling), where three rounds of 17-step workgroups lose to one round of 61
  const double steps = (double)max_m / TK;
  for (int sp = 1; sp <= max_s; ++sp) {
    const int rounds = (tiles * sp + slots - 1) / slots;
    const double cost = rounds * (steps / sp + (shared ? 40.0 : 8.0));
    if (cost < best * 0.98) { best = cost; splits = sp; }
  }
  for (int i = 0; i < count; ++i) {
    int rps = (g.pr[i].M + splits - 1) / splits;
    g.pr[i].rows_per_split = ((rps + TK - 1) / TK) * TK;
    g.pr[i].nsplit = (g.pr[i].M + g.pr[i].rows_per_split - 1) / g.pr[i].rows_per_split;
  }
  // workspace layout: [counters: one int per tile in a FIXED 16 KiB region: launches with different tile counts share
  // the workspace, and the slabs of one must never cover the (zero-between-launches) counters of another]
  // [slabs: tiles x splits partial tiles of tb x tb floats]
  g.splits = splits;
  g.slabs = nullptr;
  g.counters = nullptr;
  bool dyn_rows = false;                       // a device-side row count may leave splits without rows: they would never
  for (int i = 0; i < count; ++i) dyn_rows = dyn_rows || g.pr[i].m_dev != nullptr;   // reach the tile's arrival counter
  if (ws != nullptr && splits > 1 && !dyn_rows) {
    const int64_t cbytes = 16384;
    const int64_t need = cbytes + (int64_t)tiles * splits * tb * tb * 4;
    if (need <= ws_bytes && tiles <= 4096) {
      g.counters = reinterpret_cast<int*>(ws);
      g.slabs = reinterpret_cast<float*>(reinterpret_cast<char*>(ws) + cbytes);
    }
  }
  ProfRec* pr = prof_begin(PROF_TN0 + (big ? (legacy_loop ? 1 : 0) : 2), flops, s);
  if (big) {
    auto kern = legacy_loop ? gemm_tn_kernel<2, 4, 8> : gemm_tn_pp_kernel;
    static bool attr_done[2] = {false, false};
    if (!attr_done[legacy_loop]) {
      if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 8 * TN_TILE_BYTES) != hipSuccess)
        return UNIMM_E_HIP;
      attr_done[legacy_loop] = true;
    }
    hipLaunchKernelGGL(kern, dim3(tiles * splits), dim3(512), 8 * TN_TILE_BYTES, s, g);
  } else {
    hipLaunchKernelGGL((gemm_tn_kernel<2, 2, 4>), dim3(tiles * splits), dim3(256), 4 * TN_TILE_BYTES, s, g);
  }
  prof_end(pr, s);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

}  // namespace

extern "C" int unimm_gemm_tn_grouped_ws(const unimm_gemm_tn_args* a, int32_t count, int32_t shared_chip, void* ws,
                                        int64_t ws_bytes, void* stream) {
  if (a == nullptr || count <= 0 || (ws != nullptr && (ws_bytes < 0 || ((uintptr_t)ws & 255)))) return UNIMM_E_ARG;
  const bool shared = (shared_chip & 1) != 0;
  const bool legacy_loop = (shared_chip & 2) != 0;   // A/B only: the lock-step loop of rounds 1-2 for the 256x256 tile
  for (int i = 0; i < count; ++i) {
    const int rc = check_tn(a + i);
    if (rc != UNIMM_OK) return rc;
  }
  hipStream_t s = (hipStream_t)stream;
  const unimm_gemm_tn_args* sel[TN_MAXG];
  for (int pass = 0; pass < 2; ++pass) {           // big-tile problems share launches; so do the small ones
    const bool big = pass == 0;
    int n = 0;
    for (int i = 0; i < count; ++i) {
      if (tn_is_big(a + i) != big) continue;
      sel[n++] = a + i;
      if (n == TN_MAXG) {
        const int rc = launch_tn_group(sel, n, big, shared, legacy_loop, ws, ws_bytes, s);
        if (rc != UNIMM_OK) return rc;
        n = 0;
      }
    }
    if (n > 0) {
      const int rc = launch_tn_group(sel, n, big, shared, legacy_loop, ws, ws_bytes, s);
      if (rc != UNIMM_OK) return rc;
    }
  }
  return UNIMM_OK;
}

extern "C" int unimm_gemm_tn_grouped(const unimm_gemm_tn_args* a, int32_t count, void* stream) {
  return unimm_gemm_tn_grouped_ws(a, count, 0, nullptr, 0, stream);
}

extern "C" int unimm_gemm_tn(const unimm_gemm_tn_args* a, void* stream) {
  if (a == nullptr) return UNIMM_E_ARG;
  return unimm_gemm_tn_grouped(a, 1, stream);
}

extern "C" int unimm_prof_enable(int32_t on) {
  if (on && g_prof == nullptr) {
    g_prof = (ProfRec*)calloc(PROF_MAX, sizeof(ProfRec));
    if (g_prof == nullptr) return UNIMM_E_HIP;
  }
  g_prof_on = on != 0;
  g_prof_tn_only = on == 2;
  g_prof_n = 0;
  return UNIMM_OK;
}

// Synchronises the recorded events (call after the timed region) and fills, per variant:
// ms[v] = summed launch durations, flops[v] = summed algorithmic FLOPs, count[v] = launches.
extern "C" int unimm_prof_collect(double* ms, double* flops, int32_t* count, int32_t nvar) {
  if (!ms || !flops || !count || nvar < PROF_VARIANTS) return UNIMM_E_ARG;
  for (int v = 0; v < nvar; ++v) { ms[v] = 0.0; flops[v] = 0.0; count[v] = 0; }
  for (int t = 0; t < PROF_TAGS; ++t) { g_tag_ms[t] = 0.0; g_tag_flops[t] = 0.0; g_tag_count[t] = 0; g_tag_union_ms[t] = 0.0; }
  // per tag: the UNION of the launches' [start, end] intervals on a common time axis (the first record's start event): launches
  // of two streams that run side by side are counted once -- the wall time during which at least one tagged GEMM was executing
  struct Iv { double a, b; int tag; };
  Iv* iv = g_prof_n > 0 ? (Iv*)malloc(sizeof(Iv) * (size_t)g_prof_n) : nullptr;
  int niv = 0;
  for (int i = 0; i < g_prof_n; ++i) {
    ProfRec& r = g_prof[i];
    if (hipEventSynchronize(r.b) != hipSuccess) return UNIMM_E_HIP;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return UNIMM_E_HIP;
    if (r.variant < 0 || r.variant >= nvar) return UNIMM_E_ARG;
    ms[r.variant] += t; flops[r.variant] += r.flops; count[r.variant] += 1;
    if (r.variant < PROF_TN0 && r.tag >= 0 && r.tag < PROF_TAGS) {      // NT launches only: grouped TN launches mix layers
      g_tag_ms[r.tag] += t; g_tag_flops[r.tag] += r.flops; g_tag_count[r.tag] += 1;
      float ta = 0.f;
      if (iv != nullptr && r.tag > 0 && hipEventElapsedTime(&ta, g_prof[0].a, r.a) == hipSuccess) iv[niv++] = Iv{(double)ta, (double)ta + t, r.tag};
    }
  }
  for (int tag = 1; tag < PROF_TAGS && niv > 0; ++tag) {               // sort by start (insertion sort per tag: a few hundred records)
    int m = 0;
    for (int i = 0; i < niv; ++i) if (iv[i].tag == tag) { Iv x = iv[i]; iv[i] = iv[m]; iv[m] = x; ++m; }
    for (int i = 1; i < m; ++i) { Iv x = iv[i]; int j = i - 1; while (j >= 0 && iv[j].a > x.a) { iv[j + 1] = iv[j]; --j; } iv[j + 1] = x; }
    double end = -1e300, total = 0.0;
    for (int i = 0; i < m; ++i) {
      if (iv[i].a > end) { total += iv[i].b - iv[i].a; end = iv[i].b; }
      else if (iv[i].b > end) { total += iv[i].b - end; end = iv[i].b; }
    }
    g_tag_union_ms[tag] = total;
    // (records of other tags stay behind index m: restore nothing, every tag re-partitions the array)
  }
  free(iv);
  g_prof_n = 0;
  return UNIMM_OK;
}

extern "C" int unimm_prof_tag(int32_t tag) {
  if (tag < 0 || tag >= PROF_TAGS) return UNIMM_E_ARG;
  g_prof_tag = tag;
  return UNIMM_OK;
}

extern "C" int unimm_prof_tagged(double* ms, double* flops, int32_t* count, double* union_ms, int32_t ntags) {
  if (!ms || !flops || !count || ntags < 1) return UNIMM_E_ARG;
  for (int t = 0; t < ntags; ++t) {
    ms[t] = t < PROF_TAGS ? g_tag_ms[t] : 0.0; flops[t] = t < PROF_TAGS ? g_tag_flops[t] : 0.0; count[t] = t < PROF_TAGS ? g_tag_count[t] : 0;
    if (union_ms != nullptr) union_ms[t] = t < PROF_TAGS ? g_tag_union_ms[t] : 0.0;
  }
  return UNIMM_OK;
}
