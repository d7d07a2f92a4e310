// bf16 MFMA GEMMs for the UniMM-UL hot path (gfx950): tile choice and C ABI of unimm_gemm_nt (its kernels: gemm_nt.h, one
// translation unit per tile configuration), the weight-gradient kernels (gemm_tn, gemm_tn_pp) and the launch profiler.
#include "gemm_nt.h"

struct ProfRec { hipEvent_t a, b; int variant; int tag; double flops; };
int unimm_nt_launch_cfg1(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg3(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg6(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg7(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg8(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg9(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg10(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg12(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg14(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);
int unimm_nt_launch_cfg15(const GemmNtParams& p, int epi, bool out_f32, int want_persist, hipStream_t s, const NtSplit& sk);

int unimm_cu_count() {
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


namespace {
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
  int overwrite;          // dw = (not +=): the caller knows dw is zero and has no other contributor (plain stores, no atomic drain)
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
  if (p.overwrite != 0 && p.nsplit == 1) {
    // The tile has ONE contributor and the gradient is known to be zero (unimm_gemm_tn_args.overwrite): plain stores instead of
    // 256 KiB of memory-side atomics per workgroup -- the drain at the end of every round of a grouped launch (67 MB at the
    // chip's ~1.3 TB/s atomic rate = 51 us of a ~680 us round at 240 sequences) shrinks to a store burst.
#pragma unroll
    for (int i = 0; i < NT; ++i) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wn * 16 * NT + i * 16 + 4 * (lane >> 4) + e;
        if (n >= p.N) continue;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int k = k0 + wk * 64 + j * 16 + (lane & 15);
          if (k < p.K) __builtin_nontemporal_store(acc[i][j][e], p.dw + (size_t)n * p.lddw + k);
        }
      }
    }
    return;
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

#ifdef UNIMM_TN_TRACE
// tools/exp/tn_drift.py (variant build only): wall-clock stamps of every workgroup at reduction steps 0, 32, 64, ... --
// how far apart the workgroups that share operand panels through one XCD's L2 drift over a launch.
__device__ unsigned long long* g_tn_trace = nullptr;
extern "C" int unimm_debug_tn_trace(void* buf) {
  return hipMemcpyToSymbol(HIP_SYMBOL(g_tn_trace), &buf, sizeof(buf)) == hipSuccess ? 0 : 1;
}
#endif

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
  // Bias gradient = column sums of DY.  Every tile of a tile row (same tn) and all four waves of an n-half stage / hold the SAME
  // DY fragments, so the work is dealt out instead of sitting on one wave of one tile in nbk (which made that wave the pole of
  // every phase of its workgroup: the same launch without bias gradients ran 11 % faster, profiles/r5g_*): tile column tk sums
  // the reduction steps t with t % nbk == tk, and in such a step wave wk sums fragment wk of each n-quarter -- 16 v_dot2c per wave
  // in one step out of nbk instead of 64 in every step.  The partial sums meet in dbias by atomics (as before).
  const bool do_bias = p.dbias != nullptr;                          // wave-uniform
  float accb[2] = {0.f, 0.f};                                       // column sums of DY fragments 4 JH + wk (JH = 0, 1)
  int tb = tk;                                                      // next reduction step whose column sums are this tile's

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
  if (bias_now) {                                                                                               \
    if (wk == 0) { accb[JH] = dot_ones(fa[0][0], accb[JH]); accb[JH] = dot_ones(fa[1][0], accb[JH]); }          \
    else if (wk == 1) { accb[JH] = dot_ones(fa[0][1], accb[JH]); accb[JH] = dot_ones(fa[1][1], accb[JH]); }     \
    else if (wk == 2) { accb[JH] = dot_ones(fa[0][2], accb[JH]); accb[JH] = dot_ones(fa[1][2], accb[JH]); }     \
    else { accb[JH] = dot_ones(fa[0][3], accb[JH]); accb[JH] = dot_ones(fa[1][3], accb[JH]); }                  \
  }

#ifdef UNIMM_TN_TRACE
  unsigned long long* const tn_trace = g_tn_trace;
#endif
  for (int t = 0; t < nk; ++t) {
#ifdef UNIMM_TN_TRACE
    if (tn_trace != nullptr && tid == 0 && (t & 31) == 0 && (t >> 5) < 30) {
      tn_trace[(size_t)blockIdx.x * 32 + (t >> 5)] = wall_clock64();
      if (t == 0) {
        tn_trace[(size_t)blockIdx.x * 32 + 30] = (unsigned long long)__builtin_amdgcn_s_getreg(6164);   // HW_REG_XCC_ID[3:0]
        tn_trace[(size_t)blockIdx.x * 32 + 31] = ((unsigned long long)gi << 32) | (unsigned)(tn * 4096 + tk);
      }
    }
#endif
    const bool bias_now = do_bias && t == tb;                       // wave-uniform
    if (t == tb) tb += nbk;
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
    for (int jh = 0; jh < 2; ++jh) {
      float v = accb[jh];
      v += __shfl_xor(v, 16, 64);
      v += __shfl_xor(v, 32, 64);
      const int n = n0 + wn * 16 * NT + (4 * jh + wk) * 16 + (lane & 15);
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
constexpr int PROF_MAX = 1 << 16;
// One variant per kernel SYMBOL (what rocprofv3 --stats lists): gemm_nt / gemm_ntp x tile x epilogue x output type, and
// the three weight-gradient kernels.  NT: ((persistent * 16 + tile code) * 8 + epilogue) * 2 + out_f32 (tile code = the
// unimm_gemm_nt_args.tile code of the configuration: 1, 3, 6, 7, 8); TN: 512 = gemm_tn_pp, 513 = gemm_tn<2,4,8> (the
// lock-step loop, tools only), 514 = gemm_tn<2,2,4>.
constexpr int PROF_VARIANTS = 516;
constexpr int PROF_TN0 = 512;
bool g_prof_on = false;
bool g_prof_tn_only = false;        // unimm_prof_enable(2): only the weight-gradient launches (2 event records per launch
                                    // are host time; a rank whose step is launch-rate-bound should not pay them 340 times)
ProfRec* g_prof = nullptr;
int g_prof_n = 0;
// Caller-side tag of the launches that follow (unimm_prof_tag): bench.py's `roofline.coattention_gemms` = the GEMM launches
// the engine issues from inside a BertConnectionLayer (models/vilbert_dialog.py:655-783), forward and backward.
constexpr int PROF_TAGS = 8;
int g_prof_tag = 0;                  // (the profiler is a single-threaded measurement aid: enable / tag / launch / collect from ONE host thread)
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
  return pc <= 2 && (t.cfg == 0 || t.cfg == 1 || t.cfg == 3 || t.cfg == 6 || t.cfg == 7 || t.cfg == 8 || t.cfg == 9 || t.cfg == 10 || t.cfg == 12 || t.cfg == 14 || t.cfg == 15);
}


int launch_nt(const GemmNtParams& p, int epi, bool out_f32, const NtTune& tune, hipStream_t s, const NtSplit& sk) {
  int cfg = tune.cfg;
  const NtSplit nosplit{0, nullptr, 0};
  const int wp = tune.persist;
  if (cfg == 0) {
    // Tile choice = the configuration with the smallest modelled time: rounds of `slots` workgroups, a round of a tile
    // with W workgroups per CU costs area x W / eff, the last (partial) round only the workgroups per CU it really
    // has.  eff (per-flop efficiency against the 256x256 tile, from interleaved micro-benchmarks at K = 768..3072):
    // 192x256 0.95, 128x128 0.85.  At ~31k rows: N = 3072 -> 256x256 (5.72 rounds), N = 2304 / 768 -> 192x256 (5.73 /
    // 1.91 rounds instead of 4.29 / 1.43; 112 vs 116 us, 135 vs 151 us and vs 141 us for 128x128); the image side at
    // 8,880 rows: N = 1024 -> 192x256 (30.6 vs 37.3 us), N = 3072 -> 256x256 (58.5 vs 65.9 us).  Grids too small to
    // give every CU 1.5 workgroups of 128x128 take the 64x128 tile (three workgroups per CU: per-GPU batches of 30).
    const int cus = unimm_cu_count() & ~7;
    const long t128 = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (t128 < 384) {
      cfg = 7;
    } else {
      struct Cand { int cfg, bm, bn, w; double eff; };
      // 256x256 = the ping-pong loop (configuration 8; 3 = the lock-step ring, reachable through unimm_gemm_set_tile for A/B runs)
#ifndef UNIMM_CFG192
#define UNIMM_CFG192 12       // 192x256 with the X operand on a three-slot ring (6 = both operands on two slots: A/B builds)
#endif
      const Cand cand[3] = {{8, 256, 256, 1, 1.0}, {UNIMM_CFG192, 192, 256, 1, 0.95}, {1, 128, 128, 2, 0.85}};
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
  if (cfg == 8) return unimm_nt_launch_cfg8(p, epi, out_f32, wp, s, nosplit);
  // (configurations 2, 4, 5 - the BK = 32 rings of 4 and 5 slots, measured slower in round 1 - are not instantiated:
  //  a third of this file's compile time; unimm_gemm_nt rejects them, DESIGN.md 5 has the numbers)
  if (cfg == 6) return unimm_nt_launch_cfg6(p, epi, out_f32, wp, s, nosplit);
  if (cfg == 7) return unimm_nt_launch_cfg7(p, epi, out_f32, wp, s, sk);
  // 3-slot rings for the small tiles (two K-steps in flight): grids that leave the chip under-filled run one or two
  // workgroups per CU whose K loop is a chain of exposed L2 round trips (0.69 us per 64-deep step at 3.9k rows, 15 % of a
  // CU's MFMA rate); 9 = 64x128 (72 KiB: two workgroups per CU), 10 = 128x128 (96 KiB: one per CU)
  if (cfg == 9) return unimm_nt_launch_cfg9(p, epi, out_f32, wp, s, sk);
  if (cfg == 10) return unimm_nt_launch_cfg10(p, epi, out_f32, wp, s, sk);
  if (cfg == 3) return unimm_nt_launch_cfg3(p, epi, out_f32, wp, s, nosplit);
  if (cfg == 12) return unimm_nt_launch_cfg12(p, epi, out_f32, wp, s, nosplit);
  if (cfg == 14) return unimm_nt_launch_cfg14(p, epi, out_f32, wp, s, sk);
  if (cfg == 15) return unimm_nt_launch_cfg15(p, epi, out_f32, wp, s, sk);
  return unimm_nt_launch_cfg1(p, epi, out_f32, wp, s, sk);
}

}  // namespace

ProfRec* unimm_prof_begin(int variant, double flops, hipStream_t s) { return prof_begin(variant, flops, s); }
void unimm_prof_end(ProfRec* r, hipStream_t s) { prof_end(r, s); }

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
  if (a->epilogue < 0 || a->epilogue > UNIMM_EPI_BIAS_GELU_DG) return UNIMM_E_ARG;
  return launch_nt(p, a->epilogue, f32, tune, s, sk);
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
  // Longest reduction first.  Workgroups are dispatched in grid order as CUs free up (greedy list scheduling), so with the
  // problems sorted by descending row count the short tiles (the image side's 8.9k rows, the decoder's ~5k beside the text
  // side's ~31k: a one-stream step queues them all into one launch) fill the tail of the launch instead of sitting in front of
  // long tiles that then start late.  Stable: problems of one row count keep the caller's order.
  const unimm_gemm_tn_args* sorted[TN_MAXG];
  for (int i = 0; i < count; ++i) sorted[i] = a[i];
#ifndef UNIMM_TN_NOSORT
  for (int i = 1; i < count; ++i) {
    const unimm_gemm_tn_args* x = sorted[i];
    int j = i - 1;
    while (j >= 0 && sorted[j]->M < x->M) { sorted[j + 1] = sorted[j]; --j; }
    sorted[j + 1] = x;
  }
#endif
  a = sorted;
  for (int i = 0; i < count; ++i) {
    GemmTnParams& p = g.pr[i];
    p.dy = (const bf16_t*)a[i]->dy; p.x = (const bf16_t*)a[i]->x; p.dw = a[i]->dw; p.dbias = a[i]->dbias;
    p.m_dev = a[i]->m_dev;
    p.overwrite = a[i]->overwrite;
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
  auto fail = [&](int rc) { free(iv); g_prof_n = 0; return rc; };        // nothing half-collected is left behind
  for (int i = 0; i < g_prof_n; ++i) {
    ProfRec& r = g_prof[i];
    if (hipEventSynchronize(r.b) != hipSuccess) return fail(UNIMM_E_HIP);
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return fail(UNIMM_E_HIP);
    if (r.variant < 0 || r.variant >= nvar) return fail(UNIMM_E_ARG);
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
