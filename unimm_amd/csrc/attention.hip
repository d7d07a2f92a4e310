// Fused masked attention cores for the UniMM-UL hot path (gfx950): forward, dQ and dK/dV kernels.
//
// Replaces softmax(Q K^T / sqrt(d) + mask) . V with attention-prob dropout, in the four shapes of
// the reference: text self-attention (T=256, d=64, dense per-sequence mask;
// models/vilbert_dialog.py:390-410), visual self-attention (R=37, d=128, key-padding mask; :519-539)
// and both directions of the co-attention (:681-721).  No score matrix ever reaches HBM.
//
// Structure (one workgroup = one (sequence, head); one wave = one 32-query tile):
//   * K and V of the head are staged once into LDS (<= 256 keys x d fits: 64 KiB at d=64).
//   * S^T = K . Q^T with v_mfma_f32_32x32x16_bf16 ("swapped" product): the query sits on the lane,
//     keys in the 16 accumulator registers, so row max / row sum are register reductions plus one
//     cross-half exchange, and the whole 256-key row stays in registers (exact softmax, no online
//     rescaling needed at T <= 256).
//   * masks are bit-packed (1 bit per (query,key), 8 KiB per sequence instead of 512 KiB int64);
//     masked scores get the reference's additive -10000 (not -inf), so fully masked rows reproduce
//     softmax(raw scores) exactly as models/vilbert_dialog.py:1418 does.
//   * P (bf16) feeds O^T = V^T . P^T directly from the accumulator registers (its key index is the
//     MFMA's reduction index); V^T fragments come from ds_read_b64_tr_b16.
#include <type_traits>
#include "common.h"

namespace {

struct AttnParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  bf16_t* o; float* lse;
  const uint32_t* mask;  // [B][mask rows][nw] words, bit j of word t = key 32t+j may be attended
  // variable-length ("unpadded") mode: sequence b owns rows [off[b], off[b]+len[b]) of the packed q / kv
  // matrices; NULL = fixed layout (rows b*T .. b*T+T-1).  Tq/Tk stay the PADDED lengths: they index the
  // mask, lse and dropout counters, so a packed run reproduces the padded run bit for bit on valid rows.
  const int* q_off; const int* q_len; const int* k_off; const int* k_len;
  // forward only: a SHARED key/value segment (rows [ks_off[b], ks_off[b] + ks_len[b]) of the same k / v matrices) spliced into
  // sequence b's keys after its first ks_ins private rows: key position j is private row j (j < ks_ins), shared row j - ks_ins,
  // or private row j - ks_len[b].  The sequence then has k_len[b] + ks_len[b] keys; mask words index key POSITIONS.  What the
  // candidates of one dialog round use to attend the round's context rows, computed once (unimm_amd/scoring.py).
  const int* ks_off; const int* ks_len; int ks_ins;
  // or NULL: a permutation of the sequences; workgroups take their (sequence, head) items in THIS order.  With variable lengths
  // the longest sequences first: the dispatcher hands items to free CU slots in grid order, so the launch's tail is then made of
  // the shortest items instead of whatever the batch order put last (the step's plan writes it: unimm_plan_build).
  const int* order;
  int B, H, Tq, Tk, ldq, ldk, ldv, ldo;
  int mask_q_stride, mask_b_stride;  // in words; q stride 0 = one row per sequence (key padding)
  int parts;                          // workgroups per (sequence, head): each owns a contiguous run of 32-row tiles
  float scale;
  DropoutArg drop;
};

constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;

// ---- LDS images -------------------------------------------------------------------------------
// Rows of D bf16 (2D bytes = D/8 chunks of 16 B), 16-B chunk index XOR-swizzled per row with ONE
// function that keeps both kinds of read conflict-free: ds_read_b128 row fragments (a 16-lane group
// reads 16 rows at one chunk) and ds_read_b64_tr_b16 transposed fragments (a 32-lane half reads
// 4 rows x 64 B).  D=128 is the guide's image (b); D=64 is its 128-B-row analogue.
template <int D> __device__ __forceinline__ int swz(int row) {
  if constexpr (D == 64) { const int t = row >> 1; return (t & 7) ^ ((t & 1) << 2); }
  else return ((row & 3) << 2) | ((row >> 2) & 3);
}

// Cooperative stage of `nrows_pad` (multiple of 32) rows of one head; rows >= nrows are zero-filled.
// LDS-DMA: one wave-instruction lands 1 KiB lane-linearly (4 rows of D=128, 8 rows of D=64), so lane l
// of a group supplies the 16-byte chunk that belongs at physical position l of those rows -- the chunk
// swizzle is applied on the SOURCE address.  Every load of both images is in flight before the first
// one returns (the register-staged version paid a load->ds_write round trip per 16 bytes per thread).
// The caller waits vmcnt(0) before its barrier.
template <int D>
__device__ __forceinline__ void stage_head(const bf16_t* __restrict__ g, int ld, int nrows, int nrows_pad,
                                           char* img, int tid, int nthreads) {
  constexpr int CPR = D / 8;        // 16-byte chunks per row
  constexpr int RPI = 64 / CPR;     // rows per wave-instruction
  const int wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
  const int rl = lane / CPR, pc = lane % CPR;
  for (int grp = wave; grp < nrows_pad / RPI; grp += nwaves) {
    const int row = grp * RPI + rl;
    if (row < nrows) {
      const bf16_t* src = g + (size_t)row * ld + ((pc ^ swz<D>(row)) << 3);
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(img + grp * 1024), 16, 0, 0);
    } else {
      *reinterpret_cast<u32x4*>(img + grp * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
}
// The same with the rows of a spliced key sequence (AttnParams.ks_*): base = first element of the head in row 0 of the matrix.
template <int D>
__device__ __forceinline__ void stage_head_seg(const bf16_t* __restrict__ base, int ld, size_t kbase, size_t sbase, int ins, int slen,
                                               int nrows, int nrows_pad, char* img, int tid, int nthreads) {
  constexpr int CPR = D / 8, RPI = 64 / CPR;
  const int wave = tid >> 6, lane = tid & 63, nwaves = nthreads >> 6;
  const int rl = lane / CPR, pc = lane % CPR;
  for (int grp = wave; grp < nrows_pad / RPI; grp += nwaves) {
    const int row = grp * RPI + rl;
    if (row < nrows) {
      const size_t grow = row < ins ? kbase + row : (row < ins + slen ? sbase + (row - ins) : kbase + (row - slen));
      const bf16_t* src = base + grow * ld + ((pc ^ swz<D>(row)) << 3);
      __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(img + grp * 1024), 16, 0, 0);
    } else {
      *reinterpret_cast<u32x4*>(img + grp * 1024 + lane * 16) = u32x4{0u, 0u, 0u, 0u};
    }
  }
}
__device__ __forceinline__ void stage_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

template <int D> __device__ __forceinline__ bf16x8 read_row_frag(const char* img, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(img + row * (2 * D) + ((chunk ^ swz<D>(row)) << 4));
}

// transposed 32x32x16 operand (permuted reduction order, cdna guide 3 "accumulator tile as operand"):
// element j of lane (r = lane&31, h = lane>>5) = img[rowbase + 8*(j>>2) + 4h + (j&3)][col0 + r]
template <int D>
__device__ __forceinline__ bf16x8 read_tr_frag(const char* img, int rowbase, int col0, int lane) {
  const int h = lane >> 5, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int col = col0 + 16 * ((lane >> 4) & 1) + 4 * pp;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x4 part[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int row = rowbase + 8 * jj + 4 * h + qq;
    const int pch = (col >> 3) ^ swz<D>(row);
    part[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)LDS_PTR(img + row * (2 * D) + pch * 16 + (col & 7) * 2));
  }
  s16x8 v = {part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
  return __builtin_bit_cast(bf16x8, v);
}

__device__ __forceinline__ bf16x8 pack8(const float* p) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u4;
  u4 w = {pack2bf(p[0], p[1]), pack2bf(p[2], p[3]), pack2bf(p[4], p[5]), pack2bf(p[6], p[7])};
  return __builtin_bit_cast(bf16x8, w);
}

__device__ __forceinline__ int key_of_reg(int reg, int h) { return (reg & 3) + 8 * (reg >> 2) + 4 * h; }

// ------------------------------------------------------------------------------------------------
// forward
// ------------------------------------------------------------------------------------------------
// Row store of a transposed 32x32 accumulator set (lane = row, registers = D values): the MFMA layout leaves a lane 4
// consecutive values of every 8 -- its partner lane ^ 32 (same row) holds the other 4 -- so the direct form is one 8-byte piece
// per lane and group of 8: 2 D / 8 store instructions per row tensor, and the kernels' result stores are store-ISSUE bound
// (with them compiled out the text backward drops 275 -> 235 us, the 37-region sides 226 -> 186 / 245 -> 185: twice what the
// bytes need).  Partner lanes trade pieces with v_permlane32_swap: a lane then owns all 8 values of every second group and
// writes them as ONE 16-byte store -- half the instructions.  Executed by all lanes (the swap), stored where `ok`.
template <int D>
__device__ __forceinline__ void store_acc_row(bf16_t* __restrict__ g, const f32x16 (&a)[D / 32], float s, int h, bool ok) {
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int qp = 0; qp < 2; ++qp) {
      const uint32_t a0 = pack2bf(a[dt][8 * qp] * s, a[dt][8 * qp + 1] * s), a1 = pack2bf(a[dt][8 * qp + 2] * s, a[dt][8 * qp + 3] * s);
      const uint32_t b0 = pack2bf(a[dt][8 * qp + 4] * s, a[dt][8 * qp + 5] * s), b1 = pack2bf(a[dt][8 * qp + 6] * s, a[dt][8 * qp + 7] * s);
      // swap(x, y): x's lanes 32..63 <-> y's lanes 0..31.  h = 0 lanes end with (own group 2 qp | partner's group 2 qp),
      // h = 1 lanes with (partner's group 2 qp + 1 | own group 2 qp + 1): 8 consecutive values either way
      const auto w0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
      const auto w1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
      if (ok) *reinterpret_cast<u32x4*>(g + 32 * dt + 16 * qp + 8 * h) = u32x4{w0[0], w1[0], w0[1], w1[1]};
    }
}

template <int D, int NKT>
__global__ __launch_bounds__(512, 2) void attn_fwd_kernel(AttnParams p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KPAD = NKT * 32;
  char* kimg = smem;
  char* vimg = smem + KPAD * 2 * D;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x / p.parts, part = blockIdx.x % p.parts;
  const int b = p.order != nullptr ? p.order[item / p.H] : item / p.H, head = item % p.H;
  const int nwv = blockDim.x >> 6;                     // waves of this workgroup: wave w owns query tiles w, w + nwv, ...
  const int r = lane & 31, h = lane >> 5;

  const int Ts_b = p.ks_len ? p.ks_len[b] : 0;                                  // spliced shared keys (AttnParams.ks_*)
  int Tk_b = (p.k_len ? p.k_len[b] : p.Tk) + Ts_b;
  Tk_b = Tk_b < KPAD ? Tk_b : KPAD;                                             // (the images hold KPAD rows; the host checks the sum)
  const int Tq_b = p.q_len ? p.q_len[b] : p.Tq;
  if (part * (int)(blockDim.x >> 6) * 32 >= Tq_b) return;   // whole workgroup is padding: nothing staged
  const size_t qbase = p.q_off ? (size_t)p.q_off[b] : (size_t)b * p.Tq;
  const size_t kbase = p.k_off ? (size_t)p.k_off[b] : (size_t)b * p.Tk;
  const size_t sbase = p.ks_off ? (size_t)p.ks_off[b] : 0;
  const int kpad_b = ((Tk_b + 31) & ~31) < KPAD ? ((Tk_b + 31) & ~31) : KPAD;   // key tiles past it are skipped
  stage_head_seg<D>(p.k + head * D, p.ldk, kbase, sbase, p.ks_ins, Ts_b, Tk_b, kpad_b, kimg, tid, blockDim.x);
  stage_head_seg<D>(p.v + head * D, p.ldv, kbase, sbase, p.ks_ins, Ts_b, Tk_b, kpad_b, vimg, tid, blockDim.x);

  // Q fragments straight from HBM (each element is used once per key tile, by this wave only) and the mask words
  // of the query row; the first tile's are requested before the staging barrier
  int wt = part * nwv + wave;
  int q0, qrow;
  bool qvalid;
  bf16x8 qf[D / 16];
  uint32_t mw[NKT];
#define UNIMM_LOAD_QTILE()                                                                                    \
  {                                                                                                          \
    q0 = wt * 32;                                                                                            \
    qrow = q0 + r;                                                                                           \
    qvalid = qrow < Tq_b;                                                                                    \
    if (!qvalid) qrow = Tq_b - 1;                                                                            \
    const bf16_t* qg_ = p.q + (qbase + qrow) * p.ldq + head * D;                                             \
    _Pragma("unroll") for (int ks = 0; ks < D / 16; ++ks) qf[ks] = *reinterpret_cast<const bf16x8*>(qg_ + 16 * ks + 8 * h); \
    const uint32_t* mp_ = p.mask + (size_t)b * p.mask_b_stride + (size_t)qrow * p.mask_q_stride;            \
    _Pragma("unroll") for (int t = 0; t < NKT; ++t) mw[t] = mp_[t];                                          \
  }
  UNIMM_LOAD_QTILE()
  stage_wait();
  __syncthreads();
  // A wave walks its query tiles one after the other.  With 8 tiles on 4 waves, two workgroups fit a CU and every
  // resident wave has work: with one wave per tile, the tiles past a sequence's length were waves that exited at
  // once and left the CU (one 8-wave workgroup at these register counts) half empty for the average dialog.
  constexpr bool WALK = D == 64;     // D = 128 instances are at their register limit: one tile per wave there
  for (bool first = true; wt * 32 < Tq_b && (WALK || first); first = false, wt += nwv * p.parts) {
    if (WALK && !first) UNIMM_LOAD_QTILE()

  // ---- S^T = K . Q^T, all (non-padding) key tiles kept in registers
  f32x16 s[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    f32x16 acc = {};
    if (32 * t < Tk_b) {
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const bf16x8 kf = read_row_frag<D>(kimg, 32 * t + r, 2 * ks + h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc, 0, 0, 0);
      }
    }
    s[t] = acc;
  }

  // ---- mask + softmax over the lane's 16*NKT keys and its partner's (lane ^ 32), in the log2 domain:
  // v = s * (scale * log2 e) + (masked ? -10000 * log2 e : 0) is ONE fma per score -- the additive mask (reference
  // :1418) is the sign-extended inverted mask bit ANDed with the constant's bit pattern (v_bfe_i32 + v_and) instead of
  // shift / and / compare / select / add -- and the "key exists" test runs only on the sequence's last, partial tile.
  const float c1 = p.scale * LOG2E;
  constexpr float MOFF = -10000.0f * LOG2E;
  // (opaque copy: the 128 "key exists" compares are invariant over this wave's query tiles, and hoisted out of the walk
  // the compiler parked their lane masks in SGPRs, spilled them to VGPR lanes and read two lanes back per score)
  int tk_here = Tk_b;
  asm volatile("" : "+v"(tk_here));
  float mx = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (32 * t >= Tk_b) continue;                          // (wave-uniform) tile holds padding keys only
    const uint32_t wn = ~(mw[t] >> (4 * h));               // bit kk set = key kk (+ 4h folded in) is masked
    auto scores = [&](auto partial) {
#pragma unroll
      for (int e = 0; e < 16; e += 2) {                    // registers e, e + 1 = keys kk, kk + 1: one packed fma for the pair
        const int kk = (e & 3) + 8 * (e >> 2);
        f32x2v madd;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)wn, kk + u, 1) & __builtin_bit_cast(uint32_t, MOFF);
          madd[u] = __uint_as_float(mb);
          if constexpr (decltype(partial)::value)           // padded keys do not exist
            madd[u] = (32 * t + kk + u + 4 * h) < tk_here ? madd[u] : -INFINITY;
        }
        const f32x2v v = f32x2v{s[t][e], s[t][e + 1]} * c1 + madd;
        s[t][e] = v.x; s[t][e + 1] = v.y;
        mx = fmaxf(mx, fmaxf(v.x, v.y));
      }
    };
    if (32 * (t + 1) > Tk_b) scores(std::true_type{}); else scores(std::false_type{});
  }
  mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
  f32x2v sum2 = {0.f, 0.f};
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (32 * t >= Tk_b) continue;
#pragma unroll
    for (int e = 0; e < 16; e += 2) {
      const f32x2v a = f32x2v{s[t][e], s[t][e + 1]} - mx;
      const float p0 = __builtin_amdgcn_exp2f(a.x), p1 = __builtin_amdgcn_exp2f(a.y);
      s[t][e] = p0; s[t][e + 1] = p1;
      sum2 += f32x2v{p0, p1};
    }
  }
  float sum = sum2.x + sum2.y;
  sum += __shfl_xor(sum, 32, 64);
  // the dropout scale 1 / (1 - p) rides on the normalisation: the keep test below is a plain select
  const float inv = (p.drop.thr != 0u ? p.drop.scale : 1.0f) / sum;

  if (p.drop.thr != 0u) {
    // one hash per two neighbouring keys (registers e, e+1 with e even hold keys k, k+1 with k even)
    // linear stage of the hash: per-lane constant (query row, lane half) + a compile-time multiple of M1 per (tile, register pair)
    const uint32_t wl = drop_lin(p.drop, drop_wbase(((uint32_t)b * p.H + head) * p.Tq + (uint32_t)qrow, (uint32_t)p.Tk, 0u) + 2u * (uint32_t)h);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      if (32 * t >= Tk_b) continue;
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const uint32_t w = drop_fin(p.drop, wl + (uint32_t)((32 * t + key_of_reg(e, 0)) >> 1) * DROP_M1);
        s[t][e] = drop_keep(p.drop, w, 0u) ? s[t][e] : 0.0f;
        s[t][e + 1] = drop_keep(p.drop, w, 1u) ? s[t][e + 1] : 0.0f;
      }
    }
  }

  // ---- O^T = V^T . P^T
  f32x16 o[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt) o[dt] = f32x16{};
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (32 * t >= Tk_b) continue;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      float pv[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) pv[j] = s[t][8 * ss + j];
      const bf16x8 pf = pack8(pv);
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        const bf16x8 vf = read_tr_frag<D>(vimg, 32 * t + 16 * ss, 32 * dt, lane);
        o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o[dt], 0, 0, 0);
      }
    }
  }

  store_acc_row<D>(p.o + (qbase + (qvalid ? qrow : 0)) * p.ldo + head * D, o, inv, h, qvalid);
  if (qvalid) {
    if (p.lse != nullptr && h == 0) p.lse[((size_t)b * p.H + head) * p.Tq + qrow] = (mx + __log2f(sum)) * LN2;
  }
  }   // query tiles of this wave
#undef UNIMM_LOAD_QTILE
}


// ------------------------------------------------------------------------------------------------
// backward, part 1: dQ (query on the lane, same walk as the forward) + delta = rowsum(dO o O)
//   P = exp(S' - lse);  dP = dO . V^T (dropout mask re-generated);  dS = scale * P o (dP - delta)
//   dQ^T += K^T . dS^T    (K^T fragments: transposed reads of the same K image)
// ------------------------------------------------------------------------------------------------
struct AttnBwdParams {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* o; const bf16_t* dout;
  const float* lse; float* delta;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  const uint32_t* mask;
  const int* q_off; const int* q_len; const int* k_off; const int* k_len;
  const int* order;      // as AttnParams.order
  int B, H, Tq, Tk, ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int mask_q_stride, mask_b_stride;
  int parts;
  float scale;
  DropoutArg drop;
};

template <int D, int NKT>
__global__ __launch_bounds__(512, 2) void attn_bwd_dq_kernel(AttnBwdParams p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int KPAD = NKT * 32;
  char* kimg = smem;
  char* vimg = smem + KPAD * 2 * D;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x / p.parts, part = blockIdx.x % p.parts;
  const int b = p.order != nullptr ? p.order[item / p.H] : item / p.H, head = item % p.H;
  const int wt = part * (blockDim.x >> 6) + wave;
  const int r = lane & 31, h = lane >> 5;

  const int Tq_b = p.q_len ? p.q_len[b] : p.Tq, Tk_b = p.k_len ? p.k_len[b] : p.Tk;
  if (part * (int)(blockDim.x >> 6) * 32 >= Tq_b) return;
  const size_t qbase = p.q_off ? (size_t)p.q_off[b] : (size_t)b * p.Tq;
  const size_t kbase = p.k_off ? (size_t)p.k_off[b] : (size_t)b * p.Tk;
  const int kpad_b = ((Tk_b + 31) & ~31) < KPAD ? ((Tk_b + 31) & ~31) : KPAD;
  stage_head<D>(p.k + kbase * p.ldk + head * D, p.ldk, Tk_b, kpad_b, kimg, tid, blockDim.x);
  stage_head<D>(p.v + kbase * p.ldv + head * D, p.ldv, Tk_b, kpad_b, vimg, tid, blockDim.x);

  int qrow = wt * 32 + r;
  const bool qvalid = qrow < Tq_b;
  if (!qvalid) qrow = Tq_b - 1;
  const size_t grow = qbase + qrow;
  const bf16_t* qg = p.q + grow * p.ldq + head * D;
  const bf16_t* dog = p.dout + grow * p.lddo + head * D;
  const bf16_t* og = p.o + grow * p.ldo + head * D;
  bf16x8 qf[D / 16], dof[D / 16];
  float delta = 0.f;
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(qg + 16 * ks + 8 * h);
    dof[ks] = *reinterpret_cast<const bf16x8*>(dog + 16 * ks + 8 * h);
    const bf16x8 of = *reinterpret_cast<const bf16x8*>(og + 16 * ks + 8 * h);
#pragma unroll
    for (int j = 0; j < 8; ++j) delta += (float)dof[ks][j] * (float)of[j];
  }
  delta += __shfl_xor(delta, 32, 64);
  const size_t stat = ((size_t)b * p.H + head) * p.Tq + qrow;
  const float lse_l = p.lse[stat] * LOG2E;
  if (qvalid && h == 0) p.delta[stat] = delta;

  uint32_t mw[NKT];
  {
    const uint32_t* mp = p.mask + (size_t)b * p.mask_b_stride + (size_t)qrow * p.mask_q_stride;
#pragma unroll
    for (int t = 0; t < NKT; ++t) mw[t] = mp[t];
  }
  stage_wait();
  __syncthreads();
  if (wt * 32 >= Tq_b) return;

  f32x16 dq[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt) dq[dt] = f32x16{};
  const uint32_t dwl = drop_lin(p.drop, drop_wbase(((uint32_t)b * p.H + head) * p.Tq + (uint32_t)qrow, (uint32_t)p.Tk, 0u) + 2u * (uint32_t)h);

  // Per score (same log2-domain form as the forward): arg = s * c1 + (masked ? -10000 log2 e : 0) - lse log2 e is one
  // fma, because -lse / scale is what the S accumulator STARTS from; the dropout select carries 1 / (1 - p) and feeds an
  // fma with -delta; the softmax scale is applied once to dQ at the end instead of to every dS.
  const float c1 = p.scale * LOG2E;
  constexpr float MOFF = -10000.0f * LOG2E;
  const float s0 = -lse_l / c1;
  const bool dropping = p.drop.thr != 0u;
  const uint32_t thr16 = dropping ? (p.drop.thr >> 16) : 0u;
  const float dsc = dropping ? p.drop.scale : 1.0f;
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    if (32 * t >= Tk_b) break;
    f32x16 sacc, dpacc = {};
#pragma unroll
    for (int e = 0; e < 16; ++e) sacc[e] = s0;
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const bf16x8 kf = read_row_frag<D>(kimg, 32 * t + r, 2 * ks + h);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], sacc, 0, 0, 0);
      const bf16x8 vf = read_row_frag<D>(vimg, 32 * t + r, 2 * ks + h);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, dof[ks], dpacc, 0, 0, 0);
    }
    const uint32_t wn = ~(mw[t] >> (4 * h));
    float ds[16];
    uint32_t dw[8];   // dropout words: one hash per two neighbouring keys
    if (dropping) {
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) dw[e2] = drop_fin(p.drop, dwl + (uint32_t)((32 * t + key_of_reg(2 * e2, 0)) >> 1) * DROP_M1);
    } else {
#pragma unroll
      for (int e2 = 0; e2 < 8; ++e2) dw[e2] = 0u;          // with thr16 = 0 every field "keeps", at scale 1
    }
    // Straight-line per score (no per-element branch: without dropout the keep test is simply always true), one
    // instance with the "key exists" select on every tile: the unrolled tile loop is 8 copies of this already, and a
    // second, select-free copy per tile made the D = 128 instances slower (code size and registers).
#pragma unroll
    for (int e = 0; e < 16; e += 2) {                      // registers e, e + 1 = keys kk, kk + 1: packed fp32 math on the pair
      const int kk = (e & 3) + 8 * (e >> 2);
      f32x2v madd, tkv;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)wn, kk + u, 1) & __builtin_bit_cast(uint32_t, MOFF);
        madd[u] = (32 * t + kk + u + 4 * h) < Tk_b ? __uint_as_float(mb) : -INFINITY;   // padded keys: P = 0
        const uint32_t field = u ? (dw[e >> 1] >> 16) : (dw[e >> 1] & 0xffffu);
        tkv[u] = field >= thr16 ? dsc : 0.0f;
      }
      const f32x2v arg = f32x2v{sacc[e], sacc[e + 1]} * c1 + madd;
      f32x2v pe;
      pe.x = __builtin_amdgcn_exp2f(arg.x); pe.y = __builtin_amdgcn_exp2f(arg.y);
      const f32x2v dsv = pe * (f32x2v{dpacc[e], dpacc[e + 1]} * tkv - delta);
      ds[e] = dsv.x; ds[e + 1] = dsv.y;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 dsf = pack8(ds + 8 * ss);
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        const bf16x8 ktf = read_tr_frag<D>(kimg, 32 * t + 16 * ss, 32 * dt, lane);
        dq[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dsf, dq[dt], 0, 0, 0);
      }
    }
  }
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dq[dt][e] *= p.scale;

  store_acc_row<D>(p.dq + (qvalid ? grow : 0) * p.lddq + head * D, dq, 1.0f, h, qvalid);
}

// ------------------------------------------------------------------------------------------------
// backward, part 2: dK, dV (key on the lane; one wave = one 32-key tile, walks all query tiles)
//   S = Q . K^T and dP = dO . V^T land as [query regs][key lane]; P / dS feed
//   dV^T += dO^T . P_drop  and  dK^T += Q^T . dS  straight from the accumulators, with the
//   transposed operands read from the Q / dO images.  No atomics, no cross-wave reduction.
// ------------------------------------------------------------------------------------------------
// MAXT = 256: the variant for <= 2 key tiles (37 region keys) -- 2 compute waves + 2 staging helpers, one wave per
// SIMD, so the D = 128 instance gets 512 registers and stops spilling (148 bytes of scratch at 256)
template <int D, int NQT, int MAXT = 512>
__global__ __launch_bounds__(MAXT, MAXT == 256 ? 1 : 2) void attn_bwd_dkv_kernel(AttnBwdParams p) {
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QPAD = NQT * 32;
  char* qimg = smem;
  char* doimg = smem + QPAD * 2 * D;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * QPAD * 2 * D);
  float* del_s = lse_s + QPAD;
  uint32_t* mw_s = reinterpret_cast<uint32_t*>(del_s + QPAD);   // [key tile (wave)][query] mask words
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x / p.parts, part = blockIdx.x % p.parts;
  const int b = p.order != nullptr ? p.order[item / p.H] : item / p.H, head = item % p.H;
  const int wt = part * (blockDim.x >> 6) + wave;      // this wave's 32-key tile
  const int r = lane & 31, h = lane >> 5;

  const int Tq_b = p.q_len ? p.q_len[b] : p.Tq, Tk_b = p.k_len ? p.k_len[b] : p.Tk;
  if (part * (int)(blockDim.x >> 6) * 32 >= Tk_b) return;
  const size_t qbase = p.q_off ? (size_t)p.q_off[b] : (size_t)b * p.Tq;
  const size_t kbase = p.k_off ? (size_t)p.k_off[b] : (size_t)b * p.Tk;
  const int qpad_b = ((Tq_b + 31) & ~31) < QPAD ? ((Tq_b + 31) & ~31) : QPAD;
  stage_head<D>(p.q + qbase * p.ldq + head * D, p.ldq, Tq_b, qpad_b, qimg, tid, blockDim.x);
  stage_head<D>(p.dout + qbase * p.lddo + head * D, p.lddo, Tq_b, qpad_b, doimg, tid, blockDim.x);
  for (int i = tid; i < qpad_b; i += blockDim.x) {
    const size_t stat = ((size_t)b * p.H + head) * p.Tq + i;
    lse_s[i] = i < Tq_b ? -p.lse[stat] * LOG2E : -INFINITY;   // NEGATED; -inf => P = 0 for padded queries
    del_s[i] = i < Tq_b ? p.delta[stat] : 0.f;
  }
  {
    const int nkt_b = (Tk_b + 31) >> 5;
    const uint32_t* mb = p.mask + (size_t)b * p.mask_b_stride;
    for (int i = tid; i < nkt_b * qpad_b; i += blockDim.x) {
      const int kt = i / qpad_b, qi = i - kt * qpad_b;
      const int qc = qi < Tq_b ? qi : Tq_b - 1;
      mw_s[kt * QPAD + qi] = ~mb[(size_t)qc * p.mask_q_stride + kt];   // INVERTED: bit set = masked
    }
  }

  int krow = wt * 32 + r;
  const bool kvalid = krow < Tk_b;
  if (!kvalid) krow = Tk_b - 1;
  const size_t grow = kbase + krow;
  const bf16_t* kg = p.k + grow * p.ldk + head * D;
  const bf16_t* vg = p.v + grow * p.ldv + head * D;
  bf16x8 kf[D / 16], vf[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(kg + 16 * ks + 8 * h);
    vf[ks] = *reinterpret_cast<const bf16x8*>(vg + 16 * ks + 8 * h);
  }
  stage_wait();
  __syncthreads();
  if (wt * 32 >= Tk_b) return;

  f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt) { dk[dt] = f32x16{}; dv[dt] = f32x16{}; }
  const uint32_t* mrow = mw_s + wt * QPAD;
  const uint32_t hbase = ((uint32_t)b * p.H + head) * (uint32_t)p.Tq;
  const uint32_t halfw = ((uint32_t)p.Tk + 1u) >> 1, halfm = halfw * DROP_M1;      // D = 64 dropout words (see below)
  const uint32_t dlane = drop_lin(p.drop, (4u * (uint32_t)h + ((uint32_t)r & 1u)) * halfw + ((uint32_t)(wt * 32 + r) >> 1));
  const float c1 = p.scale * LOG2E;
  constexpr float MOFF = -10000.0f * LOG2E;
  const bool dropping = p.drop.thr != 0u;
  const uint32_t thr16 = dropping ? (p.drop.thr >> 16) : 0u;
  const float dsc = dropping ? p.drop.scale : 1.0f;

#pragma unroll 1
  for (int qt = 0; qt < NQT; ++qt) {
    if (32 * qt >= Tq_b) break;
    f32x16 sacc = {}, dpacc = {};
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const bf16x8 qf = read_row_frag<D>(qimg, 32 * qt + r, 2 * ks + h);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sacc, 0, 0, 0);
      const bf16x8 df = read_row_frag<D>(doimg, 32 * qt + r, 2 * ks + h);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[ks], dpacc, 0, 0, 0);
    }
    float pd[16], ds[16];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const int qb = 32 * qt + 8 * g4 + 4 * h;   // 4 consecutive queries for registers 4*g4 .. 4*g4+3
      const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qb);
      const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + qb);
      const u32x4 w4 = *reinterpret_cast<const u32x4*>(mrow + qb);
      // dropout words: keys k and k^1 (lanes r and r^1) share the hash word of a query row, so each lane of the
      // pair hashes two of the four queries and takes the other two from its neighbour (DPP quad_perm [1,0,3,2])
      uint32_t dwv[4] = {0u, 0u, 0u, 0u};               // without dropout: thr16 = 0, every field "keeps" at scale 1
      uint32_t odd;
      if constexpr (D == 64) {
        odd = (uint32_t)r & 1u;
        if (dropping) {
          // word index (hbase + qb + par [+ 2]) * half + kw, qb = 32 qt + 8 g4 + 4 h: the lane's part of the hash's linear
          // stage (h, par, kw) is loop-invariant, the rest is wave-uniform (scalar multiply)
          const uint32_t par = (uint32_t)r & 1u;
          const uint32_t ua = (hbase + (uint32_t)(32 * qt + 8 * g4)) * halfm;
          const uint32_t wa = drop_fin(p.drop, dlane + ua);
          const uint32_t wb = drop_fin(p.drop, dlane + ua + 2u * halfm);
          const uint32_t oa = (uint32_t)__builtin_amdgcn_mov_dpp((int)wa, 0xB1, 0xF, 0xF, true);
          const uint32_t ob = (uint32_t)__builtin_amdgcn_mov_dpp((int)wb, 0xB1, 0xF, 0xF, true);
          dwv[0] = par ? oa : wa; dwv[1] = par ? wa : oa; dwv[2] = par ? ob : wb; dwv[3] = par ? wb : ob;
        }
      } else {      // D = 128: at its register limit -- one hash per element, inside the loop below, unconditionally
        odd = (uint32_t)krow & 1u;
      }
      // log2-domain score: one fma on top of (-lse log2 e) + (masked ? -10000 log2 e : 0); lanes past the last key need
      // no select -- a key is a lane here, nothing of an invalid lane reaches a valid one and its dK / dV are not stored.
      // Straight-line, two elements at a time (packed fp32 add / fma / mul); the dropout select always runs (dwv above).
      const uint32_t fsh = odd << 4;                       // this lane's 16-bit field of a dropout word
#pragma unroll
      for (int i = 0; i < 4; i += 2) {
        const int e = 4 * g4 + i;
        f32x2v madd, tk;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)w4[i + u], (uint32_t)r, 1) & __builtin_bit_cast(uint32_t, MOFF);
          madd[u] = __uint_as_float(mb);
          uint32_t dwe = dwv[i + u];
          if constexpr (D != 64) {
            int qi = qb + i + u;
            qi = qi < Tq_b ? qi : Tq_b - 1;
            dwe = drop_word(p.drop, drop_wbase(hbase + (uint32_t)qi, (uint32_t)p.Tk, (uint32_t)krow));
          }
          tk[u] = __builtin_amdgcn_ubfe(dwe, fsh, 16) >= thr16 ? dsc : 0.0f;
        }
        const f32x2v arg = f32x2v{sacc[e], sacc[e + 1]} * c1 + (f32x2v{l4[i], l4[i + 1]} + madd);
        f32x2v pe;
        pe.x = __builtin_amdgcn_exp2f(arg.x); pe.y = __builtin_amdgcn_exp2f(arg.y);
        const f32x2v pdv = pe * tk;
        const f32x2v dsv = pe * (f32x2v{dpacc[e], dpacc[e + 1]} * tk - f32x2v{d4[i], d4[i + 1]});   // softmax scale: once, at the end
        pd[e] = pdv.x; pd[e + 1] = pdv.y;
        ds[e] = dsv.x; ds[e + 1] = dsv.y;
      }
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 pf = pack8(pd + 8 * ss);
      const bf16x8 dsf = pack8(ds + 8 * ss);
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        const bf16x8 dotf = read_tr_frag<D>(doimg, 32 * qt + 16 * ss, 32 * dt, lane);
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dv[dt], 0, 0, 0);
        const bf16x8 qtf = read_tr_frag<D>(qimg, 32 * qt + 16 * ss, 32 * dt, lane);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk[dt], 0, 0, 0);
      }
    }
  }

#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt)
#pragma unroll
    for (int e = 0; e < 16; ++e) dk[dt][e] *= p.scale;
  store_acc_row<D>(p.dk + (kvalid ? grow : 0) * p.lddk + head * D, dk, 1.0f, h, kvalid);
  store_acc_row<D>(p.dv + (kvalid ? grow : 0) * p.lddv + head * D, dv, 1.0f, h, kvalid);
}

// ------------------------------------------------------------------------------------------------
// backward in ONE kernel for the text self-attention (D = 64, up to 256 queries x 256 keys): the dK / dV walk above
// (key on the lane, a wave = one 32-key tile, all query tiles) also produces dQ, so S, P, dP, the mask and the dropout
// words are formed once per score instead of twice (they are ~2/3 of the vector work of either backward kernel).
//   dQ^T[d, q] += sum over the wave's 32 keys of K^T[d, k] dS^T[k, q]: K^T fragments are loop-invariant registers (read
//   once, transposed, from HBM; keys past the end as zeros, so their lanes' dS never counts); dS sits in the accumulator
//   layout [query regs][key lane], which is the operand layout for a reduction over QUERIES (dK) -- for the reduction over
//   keys it goes through a wave-private 2 KiB LDS tile ([key][query] bf16: 4 ds_write_b64 per lane, read back as
//   transposed fragments: no barrier, a wave's LDS accesses execute in order);
//   the eight waves' partial dQ tiles meet in an fp32 LDS accumulator [256 q][64 d (+4 pad)] in WAVE ORDER: a turn word
//   per query tile says which wave may add next (wave 0 stores, wave w adds once the word reads w, then sets w + 1) -- a
//   fixed summation order (the result is rounded to bf16: with LDS float atomics in arrival order the rounding differed
//   from run to run, 3e-3 of the gradient scale at the end of backward) and plain 16-byte read / add / write instead of
//   32 ds_add_f32 per lane and tile (LDS float atomics run at a fraction of the plain rate: the first version of this
//   kernel took 3x the time of the two-kernel path).  Scaled, rounded and written once at the end.
//   delta = rowsum(dO o O) is formed while the images are staged (the two-kernel path got it from the dQ kernel).
// LDS: Q / dO images 64 KiB + accumulator 68 KiB + dS tiles 16 KiB + lse / delta / mask words 10 KiB = 158 KiB: one
// workgroup per CU -- which the 228-register dK / dV kernel was already (two waves per SIMD).
// ------------------------------------------------------------------------------------------------
// What the three one-kernel backward forms put into LDS before their first barrier, per query of the item: -lse log2 e (-inf for
// padded queries: P = 0 there), delta = sum_d dO[q, d] O[q, d] (D / 8 consecutive lanes share a row, 16-byte chunks, reduced by
// log2(D / 8) exchanges), and the INVERTED mask words of every (key tile, query) (bit set = masked), row stride `mstride`.
template <int D>
__device__ __forceinline__ void bwd_stage_row_stats(const AttnBwdParams& p, int b, int head, size_t qbase, int Tq_b, int Tk_b, int qpad_b,
                                                    float* lse_s, float* del_s, uint32_t* mw_s, int mstride, int tid, int nthreads) {
  constexpr int LPR = D / 8;                                    // lanes per row
  for (int i = tid; i < qpad_b; i += nthreads) {
    const size_t stat = ((size_t)b * p.H + head) * p.Tq + i;
    lse_s[i] = i < Tq_b ? -p.lse[stat] * LOG2E : -INFINITY;
  }
  for (int i = tid; i < qpad_b * LPR; i += nthreads) {
    const int row = i / LPR, c = i % LPR;
    float part = 0.f;
    if (row < Tq_b) {
      const u32x4 a = *reinterpret_cast<const u32x4*>(p.dout + (qbase + row) * p.lddo + head * D + 8 * c);
      const u32x4 o = *reinterpret_cast<const u32x4*>(p.o + (qbase + row) * p.ldo + head * D + 8 * c);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        part = fmaf(__uint_as_float(a[e] << 16), __uint_as_float(o[e] << 16), part);
        part = fmaf(__uint_as_float(a[e] & 0xffff0000u), __uint_as_float(o[e] & 0xffff0000u), part);
      }
    }
#pragma unroll
    for (int o = 1; o < LPR; o <<= 1) part += __shfl_xor(part, o, 64);
    if (c == 0) del_s[row] = part;
  }
  const int nkt_b = (Tk_b + 31) >> 5;
  const uint32_t* mb = p.mask + (size_t)b * p.mask_b_stride;
  for (int i = tid; i < nkt_b * qpad_b; i += nthreads) {
    const int kt = i / qpad_b, qi = i - kt * qpad_b;
    const int qc = qi < Tq_b ? qi : Tq_b - 1;
    mw_s[kt * mstride + qi] = ~mb[(size_t)qc * p.mask_q_stride + kt];
  }
}

// S and dP of one (32-query tile, 32-key tile) pair in the key-on-the-lane layout (lane (r, h) = key r, register e = query
// key_of_reg-style: 4 consecutive queries per register quad) -> P^T (dropped and scaled: what multiplies dO) and dS^T (before the
// softmax scale), as floats.  q0 = first query of the tile; lse_s (NEGATED, log2 domain) / del_s / mrow (INVERTED mask words of this
// key tile) are LDS arrays indexed by query.  Shared by the three one-kernel backward forms.
struct BwdScoreCtx {
  uint32_t dlane, hbase, halfm, thr16;   // dropout: the lane's part of the counter, (b * H + head) * Tq, words per query row * M1, threshold
  float dsc, c1;                         // keep scale, softmax scale * log2 e
  bool dropping;
};
__device__ __forceinline__ void bwd_score_tile(const AttnBwdParams& p, const BwdScoreCtx& c, const f32x16& sacc, const f32x16& dpacc,
                                               const float* lse_s, const float* del_s, const uint32_t* mrow, int q0, int r, int h,
                                               float (&pd)[16], float (&ds)[16]) {
  constexpr float MOFF = -10000.0f * LOG2E;
  const uint32_t hbase = c.hbase, halfm = c.halfm, dlane = c.dlane, thr16 = c.thr16;
  const float dsc = c.dsc, c1 = c.c1;
  const bool dropping = c.dropping;
#pragma unroll
  for (int g4 = 0; g4 < 4; ++g4) {
    const int qb = q0 + 8 * g4 + 4 * h;   // 4 consecutive queries for registers 4*g4 .. 4*g4+3
    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + qb);
    const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + qb);
    const u32x4 w4 = *reinterpret_cast<const u32x4*>(mrow + qb);
    uint32_t dwv[4] = {0u, 0u, 0u, 0u};
    const uint32_t odd = (uint32_t)r & 1u;
    if (dropping) {       // keys k and k^1 share a hash word: each lane of the pair hashes two of the four queries (see attn_bwd_dkv_kernel)
      const uint32_t ua = (hbase + (uint32_t)(q0 + 8 * g4)) * halfm;
      const uint32_t wa = drop_fin(p.drop, dlane + ua);
      const uint32_t wb = drop_fin(p.drop, dlane + ua + 2u * halfm);
      const uint32_t oa = (uint32_t)__builtin_amdgcn_mov_dpp((int)wa, 0xB1, 0xF, 0xF, true);
      const uint32_t ob = (uint32_t)__builtin_amdgcn_mov_dpp((int)wb, 0xB1, 0xF, 0xF, true);
      dwv[0] = odd ? oa : wa; dwv[1] = odd ? wa : oa; dwv[2] = odd ? ob : wb; dwv[3] = odd ? wb : ob;
    }
    const uint32_t fsh = odd << 4;
#pragma unroll
    for (int i = 0; i < 4; i += 2) {
      const int e = 4 * g4 + i;
      f32x2v madd, tk;
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)w4[i + u], (uint32_t)r, 1) & __builtin_bit_cast(uint32_t, MOFF);
        madd[u] = __uint_as_float(mb);
        tk[u] = __builtin_amdgcn_ubfe(dwv[i + u], fsh, 16) >= thr16 ? dsc : 0.0f;
      }
      const f32x2v arg = f32x2v{sacc[e], sacc[e + 1]} * c1 + (f32x2v{l4[i], l4[i + 1]} + madd);
      f32x2v pe;
      pe.x = __builtin_amdgcn_exp2f(arg.x); pe.y = __builtin_amdgcn_exp2f(arg.y);
      const f32x2v pdv = pe * tk;
      const f32x2v dsv = pe * (f32x2v{dpacc[e], dpacc[e + 1]} * tk - f32x2v{d4[i], d4[i + 1]});
      pd[e] = pdv.x; pd[e + 1] = pdv.y;
      ds[e] = dsv.x; ds[e + 1] = dsv.y;
    }
  }
}

__device__ __forceinline__ bf16x8 read_tr_tile64(const char* tile, int rowbase, int lane) {
  // [32 rows][32 cols] bf16, 64-byte rows: element j of lane (r, h) = tile[rowbase + 8 (j >> 2) + 4 h + (j & 3)][r]
  const int h = lane >> 5, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int col = 16 * ((lane >> 4) & 1) + 4 * pp;
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x4 part[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int row = rowbase + 8 * jj + 4 * h + qq;
    part[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + row * 64 + col * 2));
  }
  s16x8 v = {part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
  return __builtin_bit_cast(bf16x8, v);
}

// The same tile with its 8-byte chunks XOR-swizzled by the row (chunk c of row r sits at c ^ ((r >> 2) & 7)): rows r and r + 4 are
// 256 bytes = one pass over the banks apart, so the 8-byte dS^T stores of 16 lanes (rows r .. r + 15, one chunk) hit four bank
// pairs four times each; swizzled they spread over all of them (SQ_LDS_BANK_CONFLICT was 31 % of the LDS-array cycles of the text
// backward, profiles/r5x_attention_text_kernels_pmc.txt).
#ifndef UNIMM_ATTN_TILE_SWZ
#define UNIMM_ATTN_TILE_SWZ 1      // (A/B builds: 0 = the unswizzled tile)
#endif
__device__ __forceinline__ int tile64_swz(int row) { return UNIMM_ATTN_TILE_SWZ ? (row >> 2) & 7 : 0; }
__device__ __forceinline__ bf16x8 read_tr_tile64s(const char* tile, int rowbase, int lane) {
  const int h = lane >> 5, i = lane & 15, qq = i >> 2, pp = i & 3;
  const int chunk = 4 * ((lane >> 4) & 1) + pp;                 // 8-byte chunk = 4 columns
  typedef __attribute__((ext_vector_type(8))) short s16x8;
  s16x4 part[2];
#pragma unroll
  for (int jj = 0; jj < 2; ++jj) {
    const int row = rowbase + 8 * jj + 4 * h + qq;
    part[jj] = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
        (__attribute__((address_space(3))) s16x4*)LDS_PTR(tile + row * 64 + ((chunk ^ tile64_swz(row)) << 3)));
  }
  s16x8 v = {part[0][0], part[0][1], part[0][2], part[0][3], part[1][0], part[1][1], part[1][2], part[1][3]};
  return __builtin_bit_cast(bf16x8, v);
}

// ------------------------------------------------------------------------------------------------
// forward for FEW queries (the co-attention direction whose queries are the 37 regions: D = 128, <= 64 queries x <= 256 keys,
// models/vilbert_dialog.py:701-721).  In attn_fwd_kernel a wave owns a 32-query tile, so two of the eight waves of an item
// compute (64 + 64 MFMAs and 128 scores per lane each) while six only help staging 128 KiB of K / V, one workgroup per CU:
// 6 x 85 us per step.  Here the KEYS are dealt to the waves, as the backward forms do:
//   phase 1   wave w = key tile w.  Its K rows come straight from HBM as MFMA operands (never staged), S^T = K Q^T for both
//             query tiles (query on the lane, the wave's 32 keys in 16 + 16 registers of the lane pair), the tile's row maximum
//             goes to LDS [wave][query]; after a barrier every wave takes the maximum over the eight, forms p = 2^(v - max),
//             leaves its partial row sum in LDS and its P tile (dropped, bf16, [query][32 keys]) for phase 2;
//   phase 2   OUTPUT-stationary: wave (d slice of 32, query tile) = 4 x 2 reads every key tile's P and V^T fragments and owns
//             O^T[32 d][32 q] over all keys -- no accumulator is shared between waves; row sums are added in wave order.
// Same scores, masks (-10000, reference :1418), dropout words and lse as attn_fwd_kernel; sums associate differently (per key
// tile), i.e. results agree to fp32 rounding, and a packed (variable-length) run equals the padded one bit for bit as there.
// LDS: V image 64 KiB + Q image 16 KiB + P tiles 32 KiB + row statistics 4 KiB = 116 KiB.
// ------------------------------------------------------------------------------------------------
#ifndef UNIMM_ATTN_FEWQ_FWD
#define UNIMM_ATTN_FEWQ_FWD 1      // (A/B builds: 0 = attn_fwd_kernel<128, 8> for this shape, as rounds 1-5)
#endif
__global__ __launch_bounds__(512, 1) void attn_fwd_fewq128_kernel(AttnParams p) {
  constexpr int D = 128, KPAD = 256, QPAD = 64, NW = 8;
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* vimg = smem;
  char* qimg = smem + KPAD * 2 * D;
  char* pt = qimg + QPAD * 2 * D;                                  // [key tile][query tile][32 q][32 keys] bf16
  float* mx_w = reinterpret_cast<float*>(pt + NW * 2 * 2048);       // [wave][QPAD]
  float* sm_w = mx_w + NW * QPAD;
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x;
  const int b = p.order != nullptr ? p.order[item / p.H] : item / p.H, head = item % p.H;
  const int r = lane & 31, h = lane >> 5;
  int Tk_b = p.k_len ? p.k_len[b] : p.Tk;
  Tk_b = Tk_b < KPAD ? Tk_b : KPAD;
  int Tq_b = p.q_len ? p.q_len[b] : p.Tq;
  Tq_b = Tq_b < QPAD ? Tq_b : QPAD;
  const size_t qbase = p.q_off ? (size_t)p.q_off[b] : (size_t)b * p.Tq;
  const size_t kbase = p.k_off ? (size_t)p.k_off[b] : (size_t)b * p.Tk;
  const int kpad_b = (Tk_b + 31) & ~31, qpad_b = (Tq_b + 31) & ~31;
  const int nqt = qpad_b >> 5;
  stage_head<D>(p.v + kbase * p.ldv + head * D, p.ldv, Tk_b, kpad_b, vimg, tid, blockDim.x);
  stage_head<D>(p.q + qbase * p.ldq + head * D, p.ldq, Tq_b, qpad_b, qimg, tid, blockDim.x);

  // ---- phase 1a: this wave's key tile
  const int wt = wave;
  const bool wave_on = wt * 32 < Tk_b;
  int krow = wt * 32 + r;
  if (krow >= Tk_b) krow = Tk_b - 1;
  const bf16_t* kg = p.k + (kbase + krow) * p.ldk + head * D;
  bf16x8 kf[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) kf[ks] = *reinterpret_cast<const bf16x8*>(kg + 16 * ks + 8 * h);
  uint32_t mw[2];
  int qrow[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    int q = 32 * qt + r;
    q = q < Tq_b ? q : Tq_b - 1;
    qrow[qt] = q;
    mw[qt] = p.mask[(size_t)b * p.mask_b_stride + (size_t)q * p.mask_q_stride + wt];
  }
  stage_wait();
  __syncthreads();

  const float c1 = p.scale * LOG2E;
  constexpr float MOFF = -10000.0f * LOG2E;
  f32x16 s[2];
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float mxl = -INFINITY;
    s[qt] = f32x16{};
    if (wave_on && qt < nqt) {
      f32x16 acc = {};
#pragma unroll
      for (int ks = 0; ks < D / 16; ++ks) {
        const bf16x8 qf = read_row_frag<D>(qimg, 32 * qt + r, 2 * ks + h);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf[ks], qf, acc, 0, 0, 0);
      }
      const uint32_t wn = ~(mw[qt] >> (4 * h));               // bit kk set = key kk (+ 4h folded in) of this tile is masked
      const bool partial = 32 * (wt + 1) > Tk_b;               // (wave-uniform) the sequence's last, partial key tile
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const int kk = (e & 3) + 8 * (e >> 2);
        f32x2v madd;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const uint32_t mb = (uint32_t)__builtin_amdgcn_sbfe((int)wn, kk + u, 1) & __builtin_bit_cast(uint32_t, MOFF);
          madd[u] = __uint_as_float(mb);
          if (partial) madd[u] = (32 * wt + kk + u + 4 * h) < Tk_b ? madd[u] : -INFINITY;   // padded keys do not exist
        }
        const f32x2v v = f32x2v{acc[e], acc[e + 1]} * c1 + madd;
        acc[e] = v.x; acc[e + 1] = v.y;
        mxl = fmaxf(mxl, fmaxf(v.x, v.y));
      }
      s[qt] = acc;
      mxl = fmaxf(mxl, __shfl_xor(mxl, 32, 64));
    }
    if (h == 0) mx_w[wave * QPAD + 32 * qt + r] = mxl;
  }
  __syncthreads();

  // ---- phase 1b: the row maximum over all key tiles, p = 2^(v - max), partial row sums, the wave's P tiles
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    float suml = 0.f;
    if (wave_on && qt < nqt) {
      float mx = -INFINITY;
#pragma unroll
      for (int w = 0; w < NW; ++w) mx = fmaxf(mx, mx_w[w * QPAD + 32 * qt + r]);
      f32x2v sum2 = {0.f, 0.f};
#pragma unroll
      for (int e = 0; e < 16; e += 2) {
        const f32x2v a = f32x2v{s[qt][e], s[qt][e + 1]} - mx;
        const float p0 = __builtin_amdgcn_exp2f(a.x), p1 = __builtin_amdgcn_exp2f(a.y);
        s[qt][e] = p0; s[qt][e + 1] = p1;
        sum2 += f32x2v{p0, p1};
      }
      suml = sum2.x + sum2.y;
      suml += __shfl_xor(suml, 32, 64);
      if (p.drop.thr != 0u) {          // the dropout words of attn_fwd_kernel: one hash per two neighbouring keys
        const uint32_t wl = drop_lin(p.drop, drop_wbase(((uint32_t)b * p.H + head) * p.Tq + (uint32_t)qrow[qt], (uint32_t)p.Tk, 0u) + 2u * (uint32_t)h);
#pragma unroll
        for (int e = 0; e < 16; e += 2) {
          const uint32_t w = drop_fin(p.drop, wl + (uint32_t)((32 * wt + key_of_reg(e, 0)) >> 1) * DROP_M1);
          s[qt][e] = drop_keep(p.drop, w, 0u) ? s[qt][e] : 0.0f;
          s[qt][e + 1] = drop_keep(p.drop, w, 1u) ? s[qt][e + 1] : 0.0f;
        }
      }
      char* tile = pt + (wt * 2 + qt) * 2048;                 // [32 q][32 keys] bf16, 64-byte rows, 8-byte chunks swizzled by the row
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4)
        *reinterpret_cast<u32x2*>(tile + r * 64 + (((2 * g4 + h) ^ tile64_swz(r)) << 3)) =
            u32x2{pack2bf(s[qt][4 * g4], s[qt][4 * g4 + 1]), pack2bf(s[qt][4 * g4 + 2], s[qt][4 * g4 + 3])};
    }
    if (h == 0) sm_w[wave * QPAD + 32 * qt + r] = suml;
  }
  __syncthreads();

  // ---- phase 2: wave (dt, qt) owns O^T[32 dt .. 32 dt + 31][query tile qt] over every key tile
  const int dt = wave & 3, qt2 = wave >> 2;
  if (qt2 < nqt) {
    f32x16 o1[1] = {f32x16{}};
    for (int t = 0; t < NW; ++t) {
      if (32 * t >= Tk_b) break;
      const char* tile = pt + (t * 2 + qt2) * 2048;
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const u32x2 lo = *reinterpret_cast<const u32x2*>(tile + r * 64 + (((4 * ss + h) ^ tile64_swz(r)) << 3));
        const u32x2 hi = *reinterpret_cast<const u32x2*>(tile + r * 64 + (((4 * ss + 2 + h) ^ tile64_swz(r)) << 3));
        const bf16x8 pf = __builtin_bit_cast(bf16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        const bf16x8 vf = read_tr_frag<D>(vimg, 32 * t + 16 * ss, 32 * dt, lane);
        o1[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pf, o1[0], 0, 0, 0);
      }
    }
    float mx = -INFINITY, sum = 0.f;
#pragma unroll
    for (int w = 0; w < NW; ++w) {
      mx = fmaxf(mx, mx_w[w * QPAD + 32 * qt2 + r]);
      sum += sm_w[w * QPAD + 32 * qt2 + r];
    }
    const float inv = (p.drop.thr != 0u ? p.drop.scale : 1.0f) / sum;
    const int q = 32 * qt2 + r;
    const bool qvalid = q < Tq_b;
    store_acc_row<32>(p.o + (qbase + (qvalid ? q : 0)) * p.ldo + head * D + 32 * dt, o1, inv, h, qvalid);
    if (dt == 0 && qvalid && h == 0 && p.lse != nullptr) p.lse[((size_t)b * p.H + head) * p.Tq + q] = (mx + __log2f(sum)) * LN2;
  }
}

template <int NQT>
__global__ __launch_bounds__(512, 2) void attn_bwd_fused_kernel(AttnBwdParams p) {
  constexpr int D = 64;
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QPAD = NQT * 32;
  constexpr int AS = D + 4;                                     // accumulator row stride in floats (16-byte accesses of 16 rows cover all banks)
  char* qimg = smem;
  char* doimg = smem + QPAD * 2 * D;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * QPAD * 2 * D);
  float* del_s = lse_s + QPAD;
  uint32_t* mw_s = reinterpret_cast<uint32_t*>(del_s + QPAD);   // [key tile (wave)][query] mask words
  char* scr = reinterpret_cast<char*>(mw_s + 8 * QPAD);         // [wave][32 keys][32 queries] bf16
  float* acc = reinterpret_cast<float*>(scr + 8 * 2048);        // [QPAD q][AS] fp32
  int* turn = reinterpret_cast<int*>(acc + QPAD * AS);          // [NQT]: the wave whose turn it is to add its partial of that query tile
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x;
  const int b = p.order != nullptr ? p.order[item / p.H] : item / p.H, head = item % p.H;
  const int wt = wave;                                          // this wave's 32-key tile
  const int r = lane & 31, h = lane >> 5;

  const int Tq_b = p.q_len ? p.q_len[b] : p.Tq, Tk_b = p.k_len ? p.k_len[b] : p.Tk;
  const size_t qbase = p.q_off ? (size_t)p.q_off[b] : (size_t)b * p.Tq;
  const size_t kbase = p.k_off ? (size_t)p.k_off[b] : (size_t)b * p.Tk;
  const int qpad_b = ((Tq_b + 31) & ~31) < QPAD ? ((Tq_b + 31) & ~31) : QPAD;
  stage_head<D>(p.q + qbase * p.ldq + head * D, p.ldq, Tq_b, qpad_b, qimg, tid, blockDim.x);
  stage_head<D>(p.dout + qbase * p.lddo + head * D, p.lddo, Tq_b, qpad_b, doimg, tid, blockDim.x);
  bwd_stage_row_stats<D>(p, b, head, qbase, Tq_b, Tk_b, qpad_b, lse_s, del_s, mw_s, QPAD, tid, blockDim.x);
  if (tid < NQT) turn[tid] = 0;

  int krow = wt * 32 + r;
  const bool kvalid = krow < Tk_b;
  const bool wave_on = wt * 32 < Tk_b;
  if (!kvalid) krow = Tk_b - 1;
  const size_t grow = kbase + krow;
  const bf16_t* kg = p.k + grow * p.ldk + head * D;
  const bf16_t* vg = p.v + grow * p.ldv + head * D;
  bf16x8 kf[D / 16], vf[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(kg + 16 * ks + 8 * h);
    vf[ks] = *reinterpret_cast<const bf16x8*>(vg + 16 * ks + 8 * h);
  }
  // K^T fragments of the wave's tile for dQ (fragment (dt, ss): element j of lane (r, h) = K[32 wt + 16 ss + 8 (j >> 2) + 4 h +
  // (j & 3)][32 dt + r]): the row fragments just loaded go through a wave-private image in the (still unused) accumulator
  // area and come back as transposed fragments; rows of keys past the end are written as zeros, so whatever those lanes
  // compute as dS never reaches dQ.  No barrier: a wave's LDS accesses execute in order, and every wave is done with this
  // before the barrier below, after which the area becomes the accumulator.
  bf16x8 ktf[2][2];
  {
    char* kt = reinterpret_cast<char*>(acc) + wave * (32 * 2 * D);
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const u32x4 v = kvalid ? __builtin_bit_cast(u32x4, kf[ks]) : u32x4{0u, 0u, 0u, 0u};
      *reinterpret_cast<u32x4*>(kt + r * (2 * D) + (((2 * ks + h) ^ swz<D>(r)) << 4)) = v;
    }
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      ktf[0][ss] = read_tr_frag<D>(kt, 16 * ss, 0, lane);
      ktf[1][ss] = read_tr_frag<D>(kt, 16 * ss, 32, lane);
    }
  }
  stage_wait();
  __syncthreads();

  f32x16 dk[D / 32], dv[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt) { dk[dt] = f32x16{}; dv[dt] = f32x16{}; }
  const uint32_t* mrow = mw_s + wt * QPAD;
  char* ws = scr + wave * 2048;
  const uint32_t hbase = ((uint32_t)b * p.H + head) * (uint32_t)p.Tq;
  const uint32_t halfw = ((uint32_t)p.Tk + 1u) >> 1, halfm = halfw * DROP_M1;
  const uint32_t dlane = drop_lin(p.drop, (4u * (uint32_t)h + ((uint32_t)r & 1u)) * halfw + ((uint32_t)(wt * 32 + r) >> 1));
  const float c1 = p.scale * LOG2E;
  const bool dropping = p.drop.thr != 0u;
  const uint32_t thr16 = dropping ? (p.drop.thr >> 16) : 0u;
  const float dsc = dropping ? p.drop.scale : 1.0f;
  const BwdScoreCtx sctx{dlane, hbase, halfm, thr16, dsc, c1, dropping};

  if (wave_on) {
#pragma unroll 1
  for (int qt = 0; qt < NQT; ++qt) {
    if (32 * qt >= Tq_b) break;
    f32x16 sacc = {}, dpacc = {};
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const bf16x8 qf = read_row_frag<D>(qimg, 32 * qt + r, 2 * ks + h);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sacc, 0, 0, 0);
      const bf16x8 df = read_row_frag<D>(doimg, 32 * qt + r, 2 * ks + h);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[ks], dpacc, 0, 0, 0);
    }
    float pd[16], ds[16];
    bwd_score_tile(p, sctx, sacc, dpacc, lse_s, del_s, mrow, 32 * qt, r, h, pd, ds);
    // dS^T of this (query tile, key tile) -> the wave's LDS tile: row = key (lane), 4 consecutive queries per store
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<u32x2*>(ws + r * 64 + (((2 * g4 + h) ^ tile64_swz(r)) << 3)) =
          u32x2{pack2bf(ds[4 * g4], ds[4 * g4 + 1]), pack2bf(ds[4 * g4 + 2], ds[4 * g4 + 3])};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 pf = pack8(pd + 8 * ss);
      const bf16x8 dsf = pack8(ds + 8 * ss);
#pragma unroll
      for (int dt = 0; dt < D / 32; ++dt) {
        const bf16x8 dotf = read_tr_frag<D>(doimg, 32 * qt + 16 * ss, 32 * dt, lane);
        dv[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pf, dv[dt], 0, 0, 0);
        const bf16x8 qtf = read_tr_frag<D>(qimg, 32 * qt + 16 * ss, 32 * dt, lane);
        dk[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsf, dk[dt], 0, 0, 0);
      }
    }
    // dQ^T partial of the tile over this wave's keys, added into the workgroup's accumulator
    f32x16 dqp[2] = {f32x16{}, f32x16{}};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const bf16x8 dstf = read_tr_tile64s(ws, 16 * ss, lane);
      dqp[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[0][ss], dstf, dqp[0], 0, 0, 0);
      dqp[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf[1][ss], dstf, dqp[1], 0, 0, 0);
    }
    // lane (q = r, h) holds d = 32 dt + 8 g + 4 h + {0..3} in registers 4 g .. 4 g + 3 of dqp[dt]
    float* ap = acc + (32 * qt + r) * AS + 4 * h;
    if (wt == 0) {
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          *reinterpret_cast<f32x4*>(ap + 32 * dt + 8 * g) = f32x4{dqp[dt][4 * g], dqp[dt][4 * g + 1], dqp[dt][4 * g + 2], dqp[dt][4 * g + 3]};
    } else {
      while (__hip_atomic_load(turn + qt, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != wt) __builtin_amdgcn_s_sleep(2);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          f32x4* a4 = reinterpret_cast<f32x4*>(ap + 32 * dt + 8 * g);
          const f32x4 o = *a4;
          *a4 = f32x4{o[0] + dqp[dt][4 * g], o[1] + dqp[dt][4 * g + 1], o[2] + dqp[dt][4 * g + 2], o[3] + dqp[dt][4 * g + 3]};
        }
    }
    if (lane == 0) __hip_atomic_store(turn + qt, wt + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
  }
  }

  if (wave_on) {                                                // (wave-uniform: the lane swap inside needs every lane)
    const bool ok = kvalid;
    store_acc_row<D>(p.dk + (ok ? grow : 0) * p.lddk + head * D, dk, p.scale, h, ok);
    store_acc_row<D>(p.dv + (ok ? grow : 0) * p.lddv + head * D, dv, 1.0f, h, ok);
  }
  __syncthreads();                                              // every wave's dQ partials are in the accumulator
  for (int i = tid; i < qpad_b * 8; i += blockDim.x) {          // 8 lanes = one 128-byte dQ row
    const int q = i >> 3, c = i & 7;
    if (q >= Tq_b) continue;
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = acc[q * AS + 8 * c + j] * p.scale;
    *reinterpret_cast<u32x4*>(p.dq + (qbase + q) * p.lddq + head * D + 8 * c) =
        u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  }
}

// ------------------------------------------------------------------------------------------------
// backward in ONE kernel for the co-attention direction whose QUERIES are the 37 regions (and for the image self-attention):
// D = 128, at most 64 queries x 256 keys.  The orientation of attn_bwd_fused_kernel (key on the lane, a wave = one 32-key tile), in
// phases so that it fits 256 registers: the score phase forms S, dP, P, dS of both query tiles ONCE (the two-kernel path formed
// them twice, in kernels of 235 and 256 + 27 spilled registers), leaves dS^T in the wave's LDS tiles and keeps P^T / dS^T as bf16
// MFMA operands (32 registers); dV and dK of the wave's keys then come out one 32-wide slice of D at a time; after a barrier dQ is
// computed output-stationary (a wave = one 32-wide slice of one query tile, over all key tiles: K^T from the tiles' LDS images).
// LDS: Q / dO images 32 KiB + K images 64 KiB + dS tiles 32 KiB + words 3 KiB = 131 KiB: one per CU.
// ------------------------------------------------------------------------------------------------
template <int NQT>
__global__ __launch_bounds__(512, 2) void attn_bwd_fewq128_kernel(AttnBwdParams p) {
  constexpr int D = 128;
  static_assert(NQT == 2, "the (sequence, head) item's queries are two 32-row tiles: P / dS of both stay in registers for phase 2");
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int QPAD = NQT * 32;
  char* qimg = smem;
  char* doimg = smem + QPAD * 2 * D;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * QPAD * 2 * D);
  float* del_s = lse_s + QPAD;
  uint32_t* mw_s = reinterpret_cast<uint32_t*>(del_s + QPAD);   // [key tile (wave)][query] mask words
  char* scr = reinterpret_cast<char*>(mw_s + 8 * QPAD);         // [wave][query tile][32 keys][32 queries] bf16: the wave's dS^T tiles
  char* kimgs = scr + 8 * NQT * 2048;                           // [wave][32 keys][D] bf16: the wave's K tile, read back transposed for dQ
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x;
  const int b = p.order != nullptr ? p.order[item / p.H] : item / p.H, head = item % p.H;
  const int wt = wave;                                          // this wave's 32-key tile
  const int r = lane & 31, h = lane >> 5;

  const int Tq_b = p.q_len ? p.q_len[b] : p.Tq, Tk_b = p.k_len ? p.k_len[b] : p.Tk;
  const size_t qbase = p.q_off ? (size_t)p.q_off[b] : (size_t)b * p.Tq;
  const size_t kbase = p.k_off ? (size_t)p.k_off[b] : (size_t)b * p.Tk;
  const int qpad_b = ((Tq_b + 31) & ~31) < QPAD ? ((Tq_b + 31) & ~31) : QPAD;
  stage_head<D>(p.q + qbase * p.ldq + head * D, p.ldq, Tq_b, qpad_b, qimg, tid, blockDim.x);
  stage_head<D>(p.dout + qbase * p.lddo + head * D, p.lddo, Tq_b, qpad_b, doimg, tid, blockDim.x);
  bwd_stage_row_stats<D>(p, b, head, qbase, Tq_b, Tk_b, qpad_b, lse_s, del_s, mw_s, QPAD, tid, blockDim.x);

  int krow = wt * 32 + r;
  const bool kvalid = krow < Tk_b;
  const bool wave_on = wt * 32 < Tk_b;
  if (!kvalid) krow = Tk_b - 1;
  const size_t grow = kbase + krow;
  const bf16_t* kg = p.k + grow * p.ldk + head * D;
  const bf16_t* vg = p.v + grow * p.ldv + head * D;
  bf16x8 kf[D / 16], vf[D / 16];
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    kf[ks] = *reinterpret_cast<const bf16x8*>(kg + 16 * ks + 8 * h);
    vf[ks] = *reinterpret_cast<const bf16x8*>(vg + 16 * ks + 8 * h);
  }
  // The wave's K tile also goes to a wave-private LDS image (keys past the end as zeros, so whatever those lanes compute as dS
  // never reaches dQ): dQ reads it back as transposed fragments.  No barrier: a wave's LDS accesses execute in order.
  char* kimg = kimgs + wave * (32 * 2 * D);
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    const u32x4 v = kvalid ? __builtin_bit_cast(u32x4, kf[ks]) : u32x4{0u, 0u, 0u, 0u};
    *reinterpret_cast<u32x4*>(kimg + r * (2 * D) + (((2 * ks + h) ^ swz<D>(r)) << 4)) = v;
  }
  stage_wait();
  __syncthreads();

  bf16x8 pfr[NQT][2], dsfr[NQT][2];                             // P^T / dS^T of (query tile, 16-query half) as MFMA operands: phase 2
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt)
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) { pfr[qt][ss] = bf16x8{}; dsfr[qt][ss] = bf16x8{}; }
  const uint32_t* mrow = mw_s + wt * QPAD;
  const uint32_t hbase = ((uint32_t)b * p.H + head) * (uint32_t)p.Tq;
  const uint32_t halfw = ((uint32_t)p.Tk + 1u) >> 1, halfm = halfw * DROP_M1;
  const uint32_t dlane = drop_lin(p.drop, (4u * (uint32_t)h + ((uint32_t)r & 1u)) * halfw + ((uint32_t)(wt * 32 + r) >> 1));
  const float c1 = p.scale * LOG2E;
  const bool dropping = p.drop.thr != 0u;
  const uint32_t thr16 = dropping ? (p.drop.thr >> 16) : 0u;
  const float dsc = dropping ? p.drop.scale : 1.0f;
  const BwdScoreCtx sctx{dlane, hbase, halfm, thr16, dsc, c1, dropping};

  if (wave_on) {
#pragma unroll
  for (int qt = 0; qt < NQT; ++qt) {
    if (32 * qt < Tq_b) {
    f32x16 sacc = {}, dpacc = {};
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const bf16x8 qf = read_row_frag<D>(qimg, 32 * qt + r, 2 * ks + h);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf, kf[ks], sacc, 0, 0, 0);
      const bf16x8 df = read_row_frag<D>(doimg, 32 * qt + r, 2 * ks + h);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(df, vf[ks], dpacc, 0, 0, 0);
    }
    float pd[16], ds[16];
    bwd_score_tile(p, sctx, sacc, dpacc, lse_s, del_s, mrow, 32 * qt, r, h, pd, ds);
    char* ws = scr + (wave * NQT + qt) * 2048;
    // dS^T of this (query tile, key tile) -> the wave's LDS tile: row = key (lane), 4 consecutive queries per store
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<u32x2*>(ws + r * 64 + (((2 * g4 + h) ^ tile64_swz(r)) << 3)) =
          u32x2{pack2bf(ds[4 * g4], ds[4 * g4 + 1]), pack2bf(ds[4 * g4 + 2], ds[4 * g4 + 3])};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) { pfr[qt][ss] = pack8(pd + 8 * ss); dsfr[qt][ss] = pack8(ds + 8 * ss); }
    }
  }
  }

  // Phase 2: dV = P^T dO and dK = dS^T Q of the wave's 32 keys, one 32-wide slice of D at a time -- 32 accumulator registers
  // alive instead of 2 x 64 through the whole score phase (which is what made this one kernel fit: 2 x 128 + the K / V
  // fragments + a query tile's S, dP, P, dS is 350 registers)
  if (wave_on) {                                                // (wave-uniform: the lane swap inside needs every lane)
    const bool ok = kvalid;
    bf16_t* dkp = p.dk + (ok ? grow : 0) * p.lddk + head * D;
    bf16_t* dvp = p.dv + (ok ? grow : 0) * p.lddv + head * D;
#pragma unroll
    for (int dt = 0; dt < D / 32; ++dt) {
      f32x16 dkc[1] = {f32x16{}}, dvc[1] = {f32x16{}};
#pragma unroll
      for (int qt = 0; qt < NQT; ++qt) {
        if (32 * qt < Tq_b) {
#pragma unroll
          for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 dotf = read_tr_frag<D>(doimg, 32 * qt + 16 * ss, 32 * dt, lane);
            dvc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dotf, pfr[qt][ss], dvc[0], 0, 0, 0);
            const bf16x8 qtf = read_tr_frag<D>(qimg, 32 * qt + 16 * ss, 32 * dt, lane);
            dkc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qtf, dsfr[qt][ss], dkc[0], 0, 0, 0);
          }
        }
      }
      store_acc_row<32>(dkp + 32 * dt, dkc, p.scale, h, ok);
      store_acc_row<32>(dvp + 32 * dt, dvc, 1.0f, h, ok);
    }
  }
  __syncthreads();                                              // every key tile's dS^T tiles and K image are in LDS
  // dQ, output-stationary: wave w owns the 32-wide slice dt = w & 3 of query tile w >> 2 and walks ALL key tiles -- K^T operand
  // from that tile's K image, dS^T from its tile, both already in LDS.  (The first version added each wave's partial dQ into an
  // fp32 LDS accumulator in wave order and converted it in a last cooperative pass: 16 hand-overs, a barrier and 33 KiB more.)
  {
    const int qt = wave >> 2, dt = wave & 3;
    if (32 * qt < Tq_b) {                                       // (wave-uniform)
      const int nkt = (Tk_b + 31) >> 5;
      f32x16 dqa[1] = {f32x16{}};
#pragma unroll
      for (int kt = 0; kt < 8; ++kt) {
        if (kt < nkt) {
#pragma unroll
          for (int ss = 0; ss < 2; ++ss) {
            const bf16x8 dstf = read_tr_tile64s(scr + (kt * NQT + qt) * 2048, 16 * ss, lane);
            const bf16x8 ktf = read_tr_frag<D>(kimgs + kt * (32 * 2 * D), 16 * ss, 32 * dt, lane);
            dqa[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ktf, dstf, dqa[0], 0, 0, 0);
          }
        }
      }
      // lane (q = r, h) holds d = 32 dt + 8 g + 4 h + {0..3} in registers 4 g .. 4 g + 3
      const int q = 32 * qt + r;
      const bool ok = q < Tq_b;
      store_acc_row<32>(p.dq + (qbase + (ok ? q : 0)) * p.lddq + head * D + 32 * dt, dqa, p.scale, h, ok);
    }
  }
}

// ------------------------------------------------------------------------------------------------
// ... and for the mirror direction, text queries attending the 37 regions (D = 128, at most 256 queries x 64 keys): the same
// orientation (key on the lane), but a wave is a 32-QUERY tile here and walks the two key tiles.  Its Q / dO rows stay in
// registers as row fragments, K / V are two 16 KiB images shared by the workgroup; dQ of the wave's queries is complete after the
// two key tiles (registers -> global); dK / dV are sums over ALL queries: every wave leaves its P^T / dS^T as MFMA operands in LDS,
// and after one barrier wave w computes ONE output -- a 32-wide slice of dK or dV for both key tiles -- over all query tiles,
// its Q^T / dO^T operand coming straight from global memory through a wave-private 2 KiB tile that is read back transposed
// (phase 2 in the kernel).  148 KiB of LDS.  Replaces a dQ kernel (98 us per layer at 240 sequences) and a dK/dV kernel
// (110 us, 312 registers) that formed S, dP, P, dS twice: 148 us.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void attn_bwd_fewk128_kernel(AttnBwdParams p) {
  constexpr int D = 128, NKT = 2, KPAD = NKT * 32, QMAX = 256, NQT = QMAX / 32;
  drop_resolve(p.drop);
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* kimg = smem;
  char* vimg = smem + KPAD * 2 * D;
  float* lse_s = reinterpret_cast<float*>(smem + 2 * KPAD * 2 * D);
  float* del_s = lse_s + QMAX;
  uint32_t* mw_s = reinterpret_cast<uint32_t*>(del_s + QMAX);   // [key tile][query] mask words
  char* scr = reinterpret_cast<char*>(mw_s + NKT * QMAX);       // [wave][dS^T tile of key tile 0 | of key tile 1 | slice tile]: 3 x ([32][32] bf16)
  char* bfrag = scr + 8 * 6144;                                 // [key tile][query tile][dS | P][16-query half]: 1 KiB MFMA operands in register layout
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int item = blockIdx.x;
  const int b = p.order != nullptr ? p.order[item / p.H] : item / p.H, head = item % p.H;
  const int wq = wave;                                          // this wave's 32-query tile
  const int r = lane & 31, h = lane >> 5;

  const int Tq_b = p.q_len ? p.q_len[b] : p.Tq, Tk_b = p.k_len ? p.k_len[b] : p.Tk;
  const size_t qbase = p.q_off ? (size_t)p.q_off[b] : (size_t)b * p.Tq;
  const size_t kbase = p.k_off ? (size_t)p.k_off[b] : (size_t)b * p.Tk;
  const int qpad_b = ((Tq_b + 31) & ~31) < QMAX ? ((Tq_b + 31) & ~31) : QMAX;
  const int kpad_b = ((Tk_b + 31) & ~31) < KPAD ? ((Tk_b + 31) & ~31) : KPAD;
  stage_head<D>(p.k + kbase * p.ldk + head * D, p.ldk, Tk_b, kpad_b, kimg, tid, blockDim.x);
  stage_head<D>(p.v + kbase * p.ldv + head * D, p.ldv, Tk_b, kpad_b, vimg, tid, blockDim.x);
  bwd_stage_row_stats<D>(p, b, head, qbase, Tq_b, Tk_b, qpad_b, lse_s, del_s, mw_s, QMAX, tid, blockDim.x);

  int qrow = wq * 32 + r;
  const bool qvalid = qrow < Tq_b;
  const bool wave_on = wq * 32 < Tq_b;
  if (!qvalid) qrow = Tq_b - 1;
  const size_t grow = qbase + qrow;
  const bf16_t* qg = p.q + grow * p.ldq + head * D;
  const bf16_t* dg = p.dout + grow * p.lddo + head * D;
  bf16x8 qf[D / 16], dof[D / 16];                               // row fragments of the wave's queries: lane (r, h) = row r, columns 16 ks + 8 h ..
#pragma unroll
  for (int ks = 0; ks < D / 16; ++ks) {
    qf[ks] = *reinterpret_cast<const bf16x8*>(qg + 16 * ks + 8 * h);
    dof[ks] = *reinterpret_cast<const bf16x8*>(dg + 16 * ks + 8 * h);
  }
  stage_wait();
  __syncthreads();

  char* wsb = scr + wave * 6144;
  const uint32_t hbase = ((uint32_t)b * p.H + head) * (uint32_t)p.Tq;
  const uint32_t halfw = ((uint32_t)p.Tk + 1u) >> 1, halfm = halfw * DROP_M1;
  const float c1 = p.scale * LOG2E;
  const bool dropping = p.drop.thr != 0u;
  const uint32_t thr16 = dropping ? (p.drop.thr >> 16) : 0u;
  const float dsc = dropping ? p.drop.scale : 1.0f;

  if (wave_on) {
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (32 * kt < Tk_b) {
    const uint32_t* mrow = mw_s + kt * QMAX;
    char* ws = wsb + kt * 2048;
    const uint32_t dlane = drop_lin(p.drop, (4u * (uint32_t)h + ((uint32_t)r & 1u)) * halfw + ((uint32_t)(kt * 32 + r) >> 1));
    const BwdScoreCtx sctx{dlane, hbase, halfm, thr16, dsc, c1, dropping};
    f32x16 sacc = {}, dpacc = {};
#pragma unroll
    for (int ks = 0; ks < D / 16; ++ks) {
      const bf16x8 kfr = read_row_frag<D>(kimg, 32 * kt + r, 2 * ks + h);
      sacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qf[ks], kfr, sacc, 0, 0, 0);
      const bf16x8 vfr = read_row_frag<D>(vimg, 32 * kt + r, 2 * ks + h);
      dpacc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(dof[ks], vfr, dpacc, 0, 0, 0);
    }
    float pd[16], ds[16];
    bwd_score_tile(p, sctx, sacc, dpacc, lse_s, del_s, mrow, 32 * wq, r, h, pd, ds);
    // dS^T of this (query tile, key tile) -> the wave's LDS tile: row = key (lane), 4 consecutive queries per store
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4)
      *reinterpret_cast<u32x2*>(ws + r * 64 + (((2 * g4 + h) ^ tile64_swz(r)) << 3)) =
          u32x2{pack2bf(ds[4 * g4], ds[4 * g4 + 1]), pack2bf(ds[4 * g4 + 2], ds[4 * g4 + 3])};
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      // P^T / dS^T of this (key tile, query tile) as MFMA operands, in register layout (16 bytes per lane): phase 2 reads them back
      char* f = bfrag + ((((kt * NQT + wq) * 2) * 2 + ss) << 10) + lane * 16;
      *reinterpret_cast<u32x4*>(f) = __builtin_bit_cast(u32x4, pack8(ds + 8 * ss));
      *reinterpret_cast<u32x4*>(f + 2048) = __builtin_bit_cast(u32x4, pack8(pd + 8 * ss));
    }
    }
  }
  // dQ^T of the wave's queries = K^T dS^T over both key tiles (lane = query, registers = D values): after the score phase, whose
  // S / dP / P / dS registers are dead by now
  f32x16 dqp[D / 32];
#pragma unroll
  for (int dt = 0; dt < D / 32; ++dt) dqp[dt] = f32x16{};
#pragma unroll
  for (int kt = 0; kt < NKT; ++kt) {
    if (32 * kt < Tk_b) {
#pragma unroll
      for (int ss = 0; ss < 2; ++ss) {
        const bf16x8 dstf = read_tr_tile64s(wsb + kt * 2048, 16 * ss, lane);
#pragma unroll
        for (int dt = 0; dt < D / 32; ++dt)
          dqp[dt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(read_tr_frag<D>(kimg, 32 * kt + 16 * ss, 32 * dt, lane), dstf, dqp[dt], 0, 0, 0);
      }
    }
  }
  store_acc_row<D>(p.dq + (qvalid ? grow : 0) * p.lddq + head * D, dqp, p.scale, h, qvalid);

  }
  __syncthreads();                                              // every query tile's P^T / dS^T operands are in LDS

  // Phase 2, output-stationary: wave w owns ONE output, the 32-wide slice dt = w & 3 of dK (w < 4) or dV (w >= 4) for both key
  // tiles, and walks ALL query tiles: A = the slice of Q^T (dO^T) -- 32 rows x 64 bytes straight from global memory (L2: phase 1
  // just read those rows) through the wave's 2 KiB tile, read back transposed -- B = the tile's dS^T (P^T) operand from LDS.  No
  // accumulator is shared between waves (the first version added every wave's partials into two LDS accumulators in wave order:
  // 16 hand-overs per wave with nothing to compute between them, ~1/3 of the item's time), and the waves whose query tile does
  // not exist work here too.  (The phase-1 blocks above are inside `if (wave_on)`; this one is not.)
  {
    const int which = wave >> 2, dt = wave & 3;
    const int nqt = (Tq_b + 31) >> 5;
    const bf16_t* src = (which == 0 ? p.q : p.dout) + head * D + 32 * dt + 16 * (lane & 1);
    const size_t ld = which == 0 ? (size_t)p.ldq : (size_t)p.lddo;
    u32x4 rg[NQT][2];                                           // the slice rows of every query tile: lane = (row = lane >> 1, 32-byte half = lane & 1)
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      int row = 32 * qt + (lane >> 1);
      row = row < Tq_b ? row : Tq_b - 1;                        // (rows past the end: their P / dS are zero, any finite values do)
      const bf16_t* g = src + (qbase + row) * ld;
      if (qt < nqt) {
        rg[qt][0] = *reinterpret_cast<const u32x4*>(g);
        rg[qt][1] = *reinterpret_cast<const u32x4*>(g + 8);
      } else {
        rg[qt][0] = u32x4{0u, 0u, 0u, 0u}; rg[qt][1] = u32x4{0u, 0u, 0u, 0u};
      }
    }
    f32x16 oacc[NKT] = {f32x16{}, f32x16{}};
    char* ws2 = scr + wave * 6144 + 4096;
#pragma unroll
    for (int qt = 0; qt < NQT; ++qt) {
      if (qt < nqt) {
        char* t = ws2 + (lane >> 1) * 64 + (lane & 1) * 32;
        *reinterpret_cast<u32x4*>(t) = rg[qt][0];
        *reinterpret_cast<u32x4*>(t + 16) = rg[qt][1];
        const bf16x8 tf0 = read_tr_tile64(ws2, 0, lane), tf1 = read_tr_tile64(ws2, 16, lane);
#pragma unroll
        for (int kt = 0; kt < NKT; ++kt) {
          if (32 * kt < Tk_b) {
            const char* f = bfrag + ((((kt * NQT + qt) * 2 + which) * 2) << 10) + lane * 16;
            oacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf0, *reinterpret_cast<const bf16x8*>(f), oacc[kt], 0, 0, 0);
            oacc[kt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tf1, *reinterpret_cast<const bf16x8*>(f + 1024), oacc[kt], 0, 0, 0);
          }
        }
      }
    }
    // lane (key = r, h) holds d = 32 dt + 8 g + 4 h + {0..3} in registers 4 g .. 4 g + 3: the 64-byte slice of the key's row
#pragma unroll
    for (int kt = 0; kt < NKT; ++kt) {
      if (32 * kt < Tk_b) {                                       // (workgroup-uniform: the lane swap inside needs every lane)
        const int key = 32 * kt + r;
        const bool ok = key < Tk_b;
        bf16_t* dst = (which == 0 ? p.dk + (kbase + (ok ? key : 0)) * p.lddk : p.dv + (kbase + (ok ? key : 0)) * p.lddv) + head * D + 32 * dt;
        const f32x16 one[1] = {oacc[kt]};
        store_acc_row<32>(dst, one, which == 0 ? p.scale : 1.0f, h, ok);
      }
    }
  }
}

// Workgroups per (sequence, head).  Default: one workgroup (up to 8 waves) per item.  Two 4-wave
// workgroups per item fit two to a CU and overlap each other's staging, but both stage the full K/V
// (or Q/dO) images: measured 589 vs 545 us for the text fwd+bwd trio, so it stays a tuning knob.
constexpr int g_attn_parts = 0;   // compile-time choice: 0 = one workgroup per item (2 = two 4-wave workgroups, 99 = no helper waves)
// Threads per workgroup: one wave per 32-row tile that computes, but never fewer than the staging needs.
// The co-attention direction with 37 region queries (or keys) has 2 compute waves and 2 x 64 KiB of text
// K/V (or Q/dO) to stage; with 128 threads that is 32 dependent load->ds_write rounds per image.  Extra
// waves stage, then leave at the per-wave exit right after the barrier.  (g_attn_parts == 99: off, for A/B.)
inline int block_threads(int tiles_per_part, int staged_rows) {
  int waves = tiles_per_part;
  if (g_attn_parts != 99 && staged_rows >= 128 && waves < 8) waves = 8;
  return waves * 64;
}
inline int parts_for(int tiles, size_t lds) {
  (void)lds;
  if (g_attn_parts > 0 && g_attn_parts != 99) return tiles > 4 ? g_attn_parts : 1;
  return 1;
}

template <typename K>
int set_lds(K kern, size_t lds) {
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
    return UNIMM_E_HIP;
  return UNIMM_OK;
}

template <int D, int NKT>
int launch_bwd_dq(const AttnBwdParams& p, hipStream_t s) {
  const size_t lds = (size_t)2 * NKT * 32 * 2 * D;
  auto kern = attn_bwd_dq_kernel<D, NKT>;
  if (set_lds(kern, lds) != UNIMM_OK) return UNIMM_E_HIP;
  AttnBwdParams q = p;
  const int tiles = (p.Tq + 31) / 32;
  q.parts = parts_for(tiles, lds);
  hipLaunchKernelGGL(kern, dim3(p.B * p.H * q.parts), dim3(block_threads((tiles + q.parts - 1) / q.parts, NKT * 32)), lds, s, q);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

template <int D, int NQT>
int launch_bwd_dkv(const AttnBwdParams& p, hipStream_t s) {
  const size_t lds = (size_t)2 * NQT * 32 * 2 * D + 2 * NQT * 32 * sizeof(float) + (size_t)8 * NQT * 32 * sizeof(uint32_t);
  AttnBwdParams q = p;
  const int tiles = (p.Tk + 31) / 32;
  q.parts = parts_for(tiles, lds);
  if constexpr (D == 128 && NQT == 8) {   // (the 37x37 image self-attention, NQT = 2, lives on many small workgroups per CU)
    if (tiles <= 2 && q.parts == 1 && g_attn_parts != 99) {
      auto k4 = attn_bwd_dkv_kernel<D, NQT, 256>;
      if (set_lds(k4, lds) != UNIMM_OK) return UNIMM_E_HIP;
      hipLaunchKernelGGL(k4, dim3(p.B * p.H), dim3(256), lds, s, q);
      UNIMM_CHECK_LAUNCH();
      return UNIMM_OK;
    }
  }
  auto kern = attn_bwd_dkv_kernel<D, NQT>;
  if (set_lds(kern, lds) != UNIMM_OK) return UNIMM_E_HIP;
  hipLaunchKernelGGL(kern, dim3(p.B * p.H * q.parts), dim3(block_threads((tiles + q.parts - 1) / q.parts, NQT * 32)), lds, s, q);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

template <int NQT>
int launch_bwd_fused(const AttnBwdParams& p, hipStream_t s) {
  constexpr int D = 64, QPAD = NQT * 32;
  const size_t lds = (size_t)2 * QPAD * 2 * D + 2 * QPAD * sizeof(float) + (size_t)8 * QPAD * sizeof(uint32_t) + 8 * 2048 +
                     (size_t)QPAD * (D + 4) * sizeof(float) + NQT * sizeof(int);
  auto kern = attn_bwd_fused_kernel<NQT>;
  if (set_lds(kern, lds) != UNIMM_OK) return UNIMM_E_HIP;
  AttnBwdParams q = p;
  q.parts = 1;
  hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(512), lds, s, q);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

inline int launch_bwd_fewq128(const AttnBwdParams& p, hipStream_t s) {
  constexpr int D = 128, NQT = 2, QPAD = NQT * 32;
  const size_t lds = (size_t)2 * QPAD * 2 * D + 2 * QPAD * sizeof(float) + (size_t)8 * QPAD * sizeof(uint32_t) + 8 * NQT * 2048 +
                     (size_t)8 * 32 * 2 * D;
  auto kern = attn_bwd_fewq128_kernel<NQT>;
  if (set_lds(kern, lds) != UNIMM_OK) return UNIMM_E_HIP;
  AttnBwdParams q = p;
  q.parts = 1;
  hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(512), lds, s, q);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

inline int launch_bwd_fewk128(const AttnBwdParams& p, hipStream_t s) {
  constexpr int D = 128, KPAD = 64, QMAX = 256;
  const size_t lds = (size_t)2 * KPAD * 2 * D + 2 * QMAX * sizeof(float) + (size_t)2 * QMAX * sizeof(uint32_t) + 8 * 6144 +
                     (size_t)2 * (QMAX / 32) * 2 * 2 * 1024;
  auto kern = attn_bwd_fewk128_kernel;
  if (set_lds(kern, lds) != UNIMM_OK) return UNIMM_E_HIP;
  AttnBwdParams q = p;
  q.parts = 1;
  hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(512), lds, s, q);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

inline int launch_fwd_fewq128(const AttnParams& p, hipStream_t s) {
  const size_t lds = (size_t)256 * 2 * 128 + (size_t)64 * 2 * 128 + 8 * 2 * 2048 + 2 * 8 * 64 * sizeof(float);
  auto kern = attn_fwd_fewq128_kernel;
  if (set_lds(kern, lds) != UNIMM_OK) return UNIMM_E_HIP;
  AttnParams q = p;
  q.parts = 1;
  hipLaunchKernelGGL(kern, dim3(p.B * p.H), dim3(512), lds, s, q);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

template <int D, int NKT>
int launch_fwd(const AttnParams& p, hipStream_t s) {
  const int waves = (p.Tq + 31) / 32;
  const size_t lds = (size_t)2 * NKT * 32 * 2 * D;
  auto kern = attn_fwd_kernel<D, NKT>;
  if (set_lds(kern, lds) != UNIMM_OK) return UNIMM_E_HIP;
  AttnParams q = p;
  q.parts = parts_for(waves, lds);
  int threads = block_threads((waves + q.parts - 1) / q.parts, NKT * 32);
  if (D == 64 && threads > 256 && g_attn_parts != 99) threads = 256;   // 4 waves walk the 8 query tiles: two workgroups per CU
  hipLaunchKernelGGL(kern, dim3(p.B * p.H * q.parts), dim3(threads), lds, s, q);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

// ------------------------------------------------------------------------------------------------
// attention probabilities as a tensor (models/vilbert_dialog.py:401-405 returns them; the encoder collects them when
// output_all_attention_masks is set, :855-929).  Not on the hot path: the fused kernels above never materialise P.
// One wave per (sequence, head, query) row of the fixed layout; a lane owns keys lane, lane + 64, ...; fp32
// softmax(q . k * scale + additive mask) from the bf16 operands, then the SAME dropout words as attn_fwd.
// ------------------------------------------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void attn_probs_kernel(AttnParams p, float* __restrict__ probs) {
  drop_resolve(p.drop);
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);           // (b * H + head) * Tq + q
  const long nrows = (long)p.B * p.H * p.Tq;
  if (row >= nrows) return;
  const int q = (int)(row % p.Tq);
  const int bh = (int)(row / p.Tq), head = bh % p.H, b = bh / p.H;
  const bf16_t* qg = p.q + ((size_t)b * p.Tq + q) * p.ldq + head * D;
  float qv[D];
#pragma unroll
  for (int c = 0; c < D / 8; ++c) {
    const u32x4 v = *reinterpret_cast<const u32x4*>(qg + 8 * c);
#pragma unroll
    for (int e = 0; e < 4; ++e) { qv[8 * c + 2 * e] = __uint_as_float(v[e] << 16); qv[8 * c + 2 * e + 1] = __uint_as_float(v[e] & 0xffff0000u); }
  }
  const uint32_t* mrow = p.mask + (size_t)b * p.mask_b_stride + (size_t)q * p.mask_q_stride;
  float sc[4];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = lane + 64 * j;
    sc[j] = -INFINITY;
    if (key < p.Tk) {
      const bf16_t* kg = p.k + ((size_t)b * p.Tk + key) * p.ldk + head * D;
      float acc = 0.f;
#pragma unroll
      for (int c = 0; c < D / 8; ++c) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(kg + 8 * c);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          acc = fmaf(qv[8 * c + 2 * e], __uint_as_float(v[e] << 16), acc);
          acc = fmaf(qv[8 * c + 2 * e + 1], __uint_as_float(v[e] & 0xffff0000u), acc);
        }
      }
      const bool ok = (mrow[key >> 5] >> (key & 31)) & 1u;
      sc[j] = acc * p.scale + (ok ? 0.f : -10000.0f);
      mx = fmaxf(mx, sc[j]);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    sc[j] = (lane + 64 * j) < p.Tk ? __expf(sc[j] - mx) : 0.f;
    sum += sc[j];
  }
  sum = wave_sum(sum);
  const float inv = 1.0f / sum;
  float* out = probs + (size_t)row * p.Tk;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int key = lane + 64 * j;
    if (key >= p.Tk) continue;
    float v = sc[j] * inv;
    if (p.drop.thr != 0u) v = drop_apply(p.drop, (uint32_t)row, (uint32_t)p.Tk, (uint32_t)key, v);
    out[key] = v;
  }
}

}  // namespace

extern "C" int unimm_attn_fwd(const unimm_attn_args* a, void* stream) {
  if (a == nullptr || !a->q || !a->k || !a->v || !a->out || !a->mask) return UNIMM_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->Tq <= 0 || a->Tk <= 0 || a->Tq > 256 || a->Tk > 256) return UNIMM_E_SHAPE;
  if (a->D != 64 && a->D != 128) return UNIMM_E_SHAPE;
  if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->ldo % 8)) return UNIMM_E_ALIGN;   // result rows leave as 16-byte stores (store_acc_row)
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out) & 15) return UNIMM_E_ALIGN;
  AttnParams p;
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v;
  p.o = (bf16_t*)a->out; p.lse = a->lse; p.mask = a->mask;
  p.q_off = a->q_off; p.q_len = a->q_len; p.k_off = a->k_off; p.k_len = a->k_len;
  if ((p.q_off == nullptr) != (p.q_len == nullptr) || (p.k_off == nullptr) != (p.k_len == nullptr)) return UNIMM_E_ARG;
  p.ks_off = a->ks_off; p.ks_len = a->ks_len; p.ks_ins = a->ks_ins;
  if ((p.ks_off == nullptr) != (p.ks_len == nullptr)) return UNIMM_E_ARG;
  if (p.ks_off != nullptr && (p.k_off == nullptr || a->ks_ins < 0 || a->drop_thr != 0u)) return UNIMM_E_ARG;   // variable-length keys, inference
  if (p.ks_off == nullptr) p.ks_ins = 0;
  p.order = a->order;
  p.B = a->B; p.H = a->H; p.Tq = a->Tq; p.Tk = a->Tk;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo;
  p.mask_q_stride = a->mask_q_stride; p.mask_b_stride = a->mask_b_stride;
  p.scale = a->scale;
  p.drop.key = a->drop_key; p.drop.thr = a->drop_thr; p.drop.scale = a->drop_scale; p.drop.salt = a->drop_salt;
  hipStream_t s = (hipStream_t)stream;
  const bool small_k = a->Tk <= 64;
  if (a->D == 64) return small_k ? launch_fwd<64, 2>(p, s) : launch_fwd<64, 8>(p, s);
  if (UNIMM_ATTN_FEWQ_FWD && !small_k && a->Tq <= 64 && p.ks_off == nullptr) return launch_fwd_fewq128(p, s);   // few queries, many keys
  return small_k ? launch_fwd<128, 2>(p, s) : launch_fwd<128, 8>(p, s);
}

extern "C" int unimm_attn_probs(const unimm_attn_args* a, float* probs, void* stream) {
  if (a == nullptr || !a->q || !a->k || !a->mask || !probs) return UNIMM_E_ARG;
  if (a->q_off || a->q_len || a->k_off || a->k_len || a->ks_off || a->ks_len) return UNIMM_E_ARG;   // fixed layout only
  if (a->B <= 0 || a->H <= 0 || a->Tq <= 0 || a->Tk <= 0 || a->Tq > 256 || a->Tk > 256) return UNIMM_E_SHAPE;
  if (a->D != 64 && a->D != 128) return UNIMM_E_SHAPE;
  if ((a->ldq % 8) || (a->ldk % 8) || (((uintptr_t)a->q | (uintptr_t)a->k) & 15)) return UNIMM_E_ALIGN;
  AttnParams p;
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = nullptr; p.o = nullptr; p.lse = nullptr; p.mask = a->mask;
  p.q_off = p.q_len = p.k_off = p.k_len = nullptr;
  p.ks_off = p.ks_len = nullptr; p.ks_ins = 0;
  p.order = nullptr;
  p.B = a->B; p.H = a->H; p.Tq = a->Tq; p.Tk = a->Tk;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = 0; p.ldo = 0;
  p.mask_q_stride = a->mask_q_stride; p.mask_b_stride = a->mask_b_stride; p.parts = 1;
  p.scale = a->scale;
  p.drop.key = a->drop_key; p.drop.thr = a->drop_thr; p.drop.scale = a->drop_scale; p.drop.salt = a->drop_salt; p.drop.key2 = 0u;
  const long rows = (long)a->B * a->H * a->Tq;
  const unsigned blocks = (unsigned)((rows + 3) / 4);
  if (a->D == 64) hipLaunchKernelGGL(attn_probs_kernel<64>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, probs);
  else hipLaunchKernelGGL(attn_probs_kernel<128>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, probs);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_attn_bwd(const unimm_attn_bwd_args* a, void* stream) {
  if (a == nullptr || !a->q || !a->k || !a->v || !a->out || !a->dout || !a->lse || !a->delta || !a->dq || !a->dk ||
      !a->dv || !a->mask)
    return UNIMM_E_ARG;
  if (a->B <= 0 || a->H <= 0 || a->Tq <= 0 || a->Tk <= 0 || a->Tq > 256 || a->Tk > 256) return UNIMM_E_SHAPE;
  if (a->D != 64 && a->D != 128) return UNIMM_E_SHAPE;
  if ((a->ldq % 8) || (a->ldk % 8) || (a->ldv % 8) || (a->ldo % 8) || (a->lddo % 8) || (a->lddq % 8) || (a->lddk % 8) ||
      (a->lddv % 8))
    return UNIMM_E_ALIGN;                                         // dQ / dK / dV rows leave as 16-byte stores (store_acc_row)
  if (((uintptr_t)a->q | (uintptr_t)a->k | (uintptr_t)a->v | (uintptr_t)a->out | (uintptr_t)a->dout | (uintptr_t)a->dq |
       (uintptr_t)a->dk | (uintptr_t)a->dv) & 15)
    return UNIMM_E_ALIGN;
  AttnBwdParams p;
  p.q = (const bf16_t*)a->q; p.k = (const bf16_t*)a->k; p.v = (const bf16_t*)a->v; p.o = (const bf16_t*)a->out;
  p.dout = (const bf16_t*)a->dout; p.lse = a->lse; p.delta = a->delta;
  p.dq = (bf16_t*)a->dq; p.dk = (bf16_t*)a->dk; p.dv = (bf16_t*)a->dv; p.mask = a->mask;
  p.q_off = a->q_off; p.q_len = a->q_len; p.k_off = a->k_off; p.k_len = a->k_len;
  if ((p.q_off == nullptr) != (p.q_len == nullptr) || (p.k_off == nullptr) != (p.k_len == nullptr)) return UNIMM_E_ARG;
  p.order = a->order;
  p.B = a->B; p.H = a->H; p.Tq = a->Tq; p.Tk = a->Tk;
  p.ldq = a->ldq; p.ldk = a->ldk; p.ldv = a->ldv; p.ldo = a->ldo; p.lddo = a->lddo;
  p.lddq = a->lddq; p.lddk = a->lddk; p.lddv = a->lddv;
  p.mask_q_stride = a->mask_q_stride; p.mask_b_stride = a->mask_b_stride;
  p.scale = a->scale;
  p.drop.key = a->drop_key; p.drop.thr = a->drop_thr; p.drop.scale = a->drop_scale; p.drop.salt = a->drop_salt;
  hipStream_t s = (hipStream_t)stream;
  const bool small_k = a->Tk <= 64, small_q = a->Tq <= 64;
  if (a->D == 64 && !small_k && !small_q) return launch_bwd_fused<8>(p, s);   // text self-attention: one kernel
#ifndef UNIMM_ATTN_FEWQ128
#define UNIMM_ATTN_FEWQ128 1     // (A/B builds: 0 = the dQ + dK/dV kernel pair for the 37-query directions too)
#endif
  // regions attend text (37 queries x up to 256 keys): one kernel, 133 against 94 + 134 us per layer at 240 sequences.  (The 37 x 37
  // image self-attention stays on the pair of small workgroups: two active key-tile waves in a one-per-CU workgroup were 95 against 81 us.)
  if (UNIMM_ATTN_FEWQ128 && a->D == 128 && small_q && !small_k) return launch_bwd_fewq128(p, s);
#ifndef UNIMM_ATTN_FEWK128
#define UNIMM_ATTN_FEWK128 1     // (A/B builds: 0 = the kernel pair for the text-attends-regions direction)
#endif
  if (UNIMM_ATTN_FEWK128 && a->D == 128 && small_k && !small_q) return launch_bwd_fewk128(p, s);   // text attends regions: one kernel
  int rc;
  if (a->D == 64) rc = small_k ? launch_bwd_dq<64, 2>(p, s) : launch_bwd_dq<64, 8>(p, s);
  else rc = small_k ? launch_bwd_dq<128, 2>(p, s) : launch_bwd_dq<128, 8>(p, s);
  if (rc != UNIMM_OK) return rc;
  if (a->D == 64) return small_q ? launch_bwd_dkv<64, 2>(p, s) : launch_bwd_dkv<64, 8>(p, s);
  return small_q ? launch_bwd_dkv<128, 2>(p, s) : launch_bwd_dkv<128, 8>(p, s);
}
