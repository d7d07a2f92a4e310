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
//
// Round 5: the NT kernels are compiled as one translation unit per block-tile configuration (gemm_nt_cfg*.hip, 32 kernels
// each: 8 epilogues x 2 output types x {one workgroup per tile, persistent}) so that the library builds in parallel; this
// header holds the device code and the per-configuration launcher, gemm.hip the tile choice, the weight-gradient kernels
// and the launch profiler.
#pragma once
#include <type_traits>
#include "common.h"
#include <stdlib.h>

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
struct NtSplit { int want; void* ws; long ws_bytes; };     // want: 0 / 1 = off, >= 2 = that many splits, -1 = the library's choice

// launch profiler and device query (defined in gemm.hip)
struct ProfRec;
ProfRec* unimm_prof_begin(int variant, double flops, hipStream_t s);
void unimm_prof_end(ProfRec* r, hipStream_t s);
int unimm_cu_count();

namespace {

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
//   Cfg<2,4,6,64,2,0,3>: 192x256 with the X operand on a ring of THREE slots (W on two): X(t+2) is requested while step t
//                     computes.  In the step the X operand of a K >= 2304 GEMM was just written by the previous kernel and
//                     streams from HBM (W, a few MB, stays in L2): one K step of prefetch does not cover that latency
//                     (profiles/r5h_cold_operand_microbench.txt: +15-22 %), and 2 x 32 + 3 x 24 = 136 KiB is what the LDS has
template <int WM_, int WN_, int MT_, int BK_, int STAGES_, int PP_ = 0, int XS_ = 0>
struct Cfg {
  static constexpr int WM = WM_, WN = WN_, MT = MT_, BK = BK_, STAGES = STAGES_;
  static constexpr int XS = XS_ ? XS_ : STAGES_;              // ring slots of the X operand (W: STAGES)
  static constexpr bool PP = PP_ != 0;                        // ping-pong main loop (nt_mainloop_pp)
  static constexpr int BM = 16 * MT * WM, BN = 64 * WN, NW = WM * WN, THREADS = 64 * NW;
  static constexpr int ROWB = BK * 2;                         // bytes per staged row
  static constexpr int RPI = 1024 / ROWB;                     // rows per LDS-DMA wave-instruction
  static constexpr int STAGE_BYTES = (BM + BN) * ROWB;
  static constexpr int G = (BM + BN) / RPI / NW;              // LDS-DMA wave-instructions per wave per K-step
  static constexpr int JP = MT == 6 ? 3 : (MT < 4 ? MT : 4);  // 16-row sub-tiles of a wave's tile per epilogue pass
  static constexpr int SLAB_ROWS = 16 * JP;                   // rows per pass
  static constexpr int SLAB_BYTES = NW * SLAB_ROWS * 68 * 4;  // epilogue transpose slabs
  static constexpr int WB = BN * ROWB, XB = BM * ROWB;         // bytes of a staged W / X tile
  static constexpr int GW = BN / RPI / NW, GX = BM / RPI / NW; // LDS-DMA wave-instructions per wave per K-step, by operand
  static constexpr int RING_BYTES = STAGES * WB + XS * XB;
  static constexpr int LDS = RING_BYTES > SLAB_BYTES ? RING_BYTES : SLAB_BYTES;
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
// The same with the operands on rings of their own ([W slot 0 | W slot 1 | X slot 0 | X slot 1 | X slot 2], Cfg::XS): instruction
// r of the wave's G belongs to W (r < GW: K step kw, slot kw_slot) or to X (K step kx, slot kx_slot); `on` = that step exists.
template <class C>
__device__ __forceinline__ void ring_stage_one_x(const RingStage<C>& st, int r, int k0, int slot, int wave) {
  const bool is_w = r < C::GW;
  const uint32_t dst = st.lds + (uint32_t)(is_w ? slot * C::WB + (r * C::NW + wave) * 1024
                                                : C::STAGES * C::WB + slot * C::XB + ((r - C::GW) * C::NW + wave) * 1024);
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
  constexpr bool X3 = C::XS == 3 && S == 2;  // the X operand runs one K step further ahead than W (Cfg<...,3>)
  static_assert(C::XS == S || X3, "separate operand rings: two W slots + three X slots");
  if constexpr (X3) {
    // prologue, in this order (vmcnt is in-order): W(0), X(0), then X(1) -- which the first wait leaves in flight
#pragma unroll
    for (int r = 0; r < G; ++r) ring_stage_one_x<C>(rst, r, kb * BK, 0, wave_u);
    if (nk > 1) {
#pragma unroll
      for (int r = C::GW; r < G; ++r) ring_stage_one_x<C>(rst, r, (kb + 1) * BK, 1, wave_u);
    }
  } else {
  // prologue: fill S-1 ring slots
#pragma unroll
  for (int s = 0; s < S - 1; ++s)
    if (s < nk) ring_stage_step<C>(rst, (kb + s) * BK, s, wave_u);
  }

  for (int t = 0; t < nk; ++t) {
    // K-step t has landed once at most min(S-2, nk-1-t) younger steps are still in flight
    const int rem = nk - 1 - t;
    if constexpr (X3) { if (rem >= 1) wait_vmcnt<C::GX>(); else wait_vmcnt<0>(); }   // W(t), X(t) landed; X(t+1) may be in flight
    else if constexpr (S == 2) wait_vmcnt<0>();
    else if constexpr (S == 3) { if (rem >= 1) wait_vmcnt<G>(); else wait_vmcnt<0>(); }
    else { if (rem >= 2) wait_vmcnt<2 * G>(); else if (rem == 1) wait_vmcnt<G>(); else wait_vmcnt<0>(); }
    __builtin_amdgcn_s_barrier();            // every wave's loads of step t landed; step t-1 fully read
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (X3 && !SPREAD) {           // 4-wave tiles refill at the top of the step: W(t+1) first, then X(t+2)
      if (t + 1 < nk) {
#pragma unroll
        for (int r = 0; r < C::GW; ++r) ring_stage_one_x<C>(rst, r, (kb + t + 1) * BK, (t + 1) & 1, wave_u);
      }
      if (t + 2 < nk) {
#pragma unroll
        for (int r = C::GW; r < G; ++r) ring_stage_one_x<C>(rst, r, (kb + t + 2) * BK, (t + 2) % 3, wave_u);
      }
    } else if (t + S - 1 < nk)               // refill the slot step t-1 used
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
      const uint32_t so = (uint32_t)((t % S) * (X3 ? C::WB : C::STAGE_BYTES));
      // (X3: ax0 points at the X tile of the lock-step layout, BN rows behind the W tile of slot 0; X slot x starts at S * WB + x * XB)
      const uint32_t sx = X3 ? (uint32_t)((S - 1) * C::WB + (t % 3) * C::XB) : so;
      uint32_t aw[KS], ax[KS];
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) { aw[ks] = aw0[ks] + so; ax[ks] = ax0[ks] + sx; }
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
          if constexpr (X3) {                                                      /* W(t+1) first, then X(t+2) */    \
            if ((u) < C::GW ? (t + 1 < nk) : (t + 2 < nk))                                                           \
              ring_stage_one_x<C>(rst, u, (kb + t + ((u) < C::GW ? 1 : 2)) * BK, (u) < C::GW ? ((t + 1) & 1) : ((t + 2) % 3), wave_u); \
          } else {                                                                                                   \
            if (t + 1 < nk) ring_stage_one<C>(rst, (kb + t + 1) * BK, (t + 1) & 1, wave_u, u);             \
          }                                                                                                          \
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
#ifdef UNIMM_STAGGER_US
  // tools/exp/desync_steady_state.py (variant build only): half of every XCD's workgroups start UNIMM_STAGGER_US late, so that
  // their tile boundaries (epilogue bursts) fall between the other half's for the rest of the launch
  if ((blockIdx.x >> 3) & 1) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)(UNIMM_STAGGER_US) * 100) __builtin_amdgcn_s_sleep(32);
  }
#endif
  for (int lt = blockIdx.x; lt < ntiles; lt += gridDim.x) {
    if (lt != (int)blockIdx.x) __builtin_amdgcn_s_barrier();   // every wave has left its epilogue slab (it aliases the ring)
    nt_tile<C, EPI, OUT_F32>(p, smem, xcd_remap(lt, ntiles));
  }
}

template <class C> constexpr int nt_tile_code() {
  return C::PP ? 8 : (C::MT == 8 ? 3 : (C::MT == 6 ? (C::XS == 3 ? 12 : 6) : (C::MT == 2 ? (C::STAGES == 3 ? 9 : (C::XS == 3 ? 15 : 7)) : (C::STAGES == 3 ? 10 : (C::XS == 3 ? 14 : 1)))));
}

template <class C, int EPI, bool F32> constexpr auto pick_nt_kernel() { return &gemm_nt_kernel<C, EPI, F32>; }

template <class C, int EPI>
int launch_nt_cfg(const GemmNtParams& p_in, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk = NtSplit{0, nullptr, 0}) {
  GemmNtParams p = p_in;
  int nwg = ((p.M + C::BM - 1) / C::BM) * ((p.N + C::BN - 1) / C::BN);
  p.ksplit = 1; p.slabs = nullptr; p.counters = nullptr;
  if (!C::PP && sk.want != 0 && sk.want != 1 && sk.ws != nullptr) {
    // Split only grids that leave the chip under-filled (every split workgroup resident at once) and reductions long enough
    // to pay for the join (>= 8 K-steps per split).
    const int slots = (unimm_cu_count() & ~7) * C::WG_PER_CU, nk = p.K / C::BK;
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
    const int slots = (unimm_cu_count() & ~7) * C::WG_PER_CU;
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
      ProfRec* pr = unimm_prof_begin(((16 + nt_tile_code<C>()) * 8 + EPI) * 2 + (out_f32 ? 1 : 0), 2.0 * p.M * (double)p.N * p.K, s);
      if (out_f32) hipLaunchKernelGGL(p32, dim3(slots), dim3(C::THREADS), C::LDS, s, p, nwg);
      else hipLaunchKernelGGL(p16, dim3(slots), dim3(C::THREADS), C::LDS, s, p, nwg);
      unimm_prof_end(pr, s);
      UNIMM_CHECK_LAUNCH();
      return UNIMM_OK;
    }
  }
  ProfRec* pr = unimm_prof_begin((nt_tile_code<C>() * 8 + EPI) * 2 + (out_f32 ? 1 : 0), 2.0 * p.M * (double)p.N * p.K, s);
  if (out_f32) hipLaunchKernelGGL(k32, dim3(nwg), dim3(C::THREADS), C::LDS, s, p);
  else hipLaunchKernelGGL(k16, dim3(nwg), dim3(C::THREADS), C::LDS, s, p);
  unimm_prof_end(pr, s);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

// every epilogue of one configuration (the body of a gemm_nt_cfg*.hip translation unit)
template <class C>
int nt_launch_epi(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk) {
  switch (epi) {
    case UNIMM_EPI_BIAS: return launch_nt_cfg<C, UNIMM_EPI_BIAS>(p, out_f32, want_persist, s, sk);
    case UNIMM_EPI_BIAS_GELU: return launch_nt_cfg<C, UNIMM_EPI_BIAS_GELU>(p, out_f32, want_persist, s, sk);
    case UNIMM_EPI_BIAS_DROP_RESID: return launch_nt_cfg<C, UNIMM_EPI_BIAS_DROP_RESID>(p, out_f32, want_persist, s, sk);
    case UNIMM_EPI_BIAS_RELU: return launch_nt_cfg<C, UNIMM_EPI_BIAS_RELU>(p, out_f32, want_persist, s, sk);
    case UNIMM_EPI_DGELU: return launch_nt_cfg<C, UNIMM_EPI_DGELU>(p, out_f32, want_persist, s, sk);
    case UNIMM_EPI_ADD: return launch_nt_cfg<C, UNIMM_EPI_ADD>(p, out_f32, want_persist, s, sk);
    case UNIMM_EPI_MUL: return launch_nt_cfg<C, UNIMM_EPI_MUL>(p, out_f32, want_persist, s, sk);
    case UNIMM_EPI_BIAS_GELU_DG: return launch_nt_cfg<C, UNIMM_EPI_BIAS_GELU_DG>(p, out_f32, want_persist, s, sk);
    default: return UNIMM_E_ARG;
  }
}

}  // namespace
