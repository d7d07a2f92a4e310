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
#include "common.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;  // 16 KiB per operand tile

struct GemmNtParams {
  const bf16_t* x; const bf16_t* w; const float* bias; const void* aux;  // aux: fp32 for DROP_RESID, bf16 otherwise
  void* out; bf16_t* out2;
  int M, N, K, ldx, ldw, ldaux, ldo;
  DropoutArg drop;
};

// Stage one 128 x 64 bf16 operand tile: 16 wave-instructions of 1 KiB, 4 per wave.
__device__ __forceinline__ void stage_tile(const bf16_t* __restrict__ g, int ld, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r * 32 + wave * 8 + (lane >> 3);
    const int chunk = (lane & 7) ^ ((row >> 1) & 7);
    int grow = row0 + row;
    grow = grow < nrows ? grow : nrows - 1;  // edge rows re-read a valid row; their outputs are never stored
    const bf16_t* src = g + (size_t)grow * ld + k0 + chunk * 8;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + (r * 32 + wave * 8) * 128), 16, 0, 0);
  }
}

__device__ __forceinline__ bf16x8 read_frag(const char* lds_tile, int row, int chunk) {
  return *reinterpret_cast<const bf16x8*>(lds_tile + row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4));
}

template <int EPI, bool OUT_F32>
__global__ __launch_bounds__(256, 2) void gemm_nt_kernel(GemmNtParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nbn = (p.N + BN - 1) / BN;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int tm = lid / nbn, tn = lid - tm * nbn;
  const int m0 = tm * BM, n0 = tn * BN;
  const int wm = wave >> 1, wn = wave & 1;

  f32x4 acc[4][4];  // [n-subtile i][m-subtile j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int nk = p.K / BK;
  // LDS: [stage][W tile | X tile]
  stage_tile(p.w, p.ldw, n0, p.N, 0, smem, wave, lane);
  stage_tile(p.x, p.ldx, m0, p.M, 0, smem + TILE_BYTES, wave, lane);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0)
  __syncthreads();

  for (int t = 0; t < nk; ++t) {
    const int cur = t & 1;
    if (t + 1 < nk) {
      char* nxt = smem + (cur ^ 1) * 2 * TILE_BYTES;
      stage_tile(p.w, p.ldw, n0, p.N, (t + 1) * BK, nxt, wave, lane);
      stage_tile(p.x, p.ldx, m0, p.M, (t + 1) * BK, nxt + TILE_BYTES, wave, lane);
    }
    const char* tw = smem + cur * 2 * TILE_BYTES;
    const char* tx = tw + TILE_BYTES;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fw[4], fx[4];
      const int chunk = ks * 4 + (lane >> 4);
#pragma unroll
      for (int i = 0; i < 4; ++i) fw[i] = read_frag(tw, wn * 64 + i * 16 + (lane & 15), chunk);
#pragma unroll
      for (int j = 0; j < 4; ++j) fx[j] = read_frag(tx, wm * 64 + j * 16 + (lane & 15), chunk);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fw[i], fx[j], acc[i][j], 0, 0, 0);
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): next tile landed
    __syncthreads();
  }

  // ---- epilogue: lane holds rows m = .. + (lane&15), columns n = .. + 4*(lane>>4) + {0..3}
  const int n_l = 4 * (lane >> 4);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + wn * 64 + i * 16 + n_l;
    if (n >= p.N) continue;
    const bool full = (n + 3 < p.N);
    float b[4] = {0.f, 0.f, 0.f, 0.f};
    if (p.bias != nullptr) {
#pragma unroll
      for (int e = 0; e < 4; ++e) b[e] = (n + e < p.N) ? p.bias[n + e] : 0.f;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int m = m0 + wm * 64 + j * 16 + (lane & 15);
      if (m >= p.M) continue;
      float v[4], u[4];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = acc[i][j][e] + b[e];
      if constexpr (EPI == UNIMM_EPI_BIAS_DROP_RESID || EPI == UNIMM_EPI_DGELU || EPI == UNIMM_EPI_ADD) {
        float a[4];
        if constexpr (EPI == UNIMM_EPI_BIAS_DROP_RESID) {   // fp32 residual stream
          const float* ap = reinterpret_cast<const float*>(p.aux) + (size_t)m * p.ldaux + n;
          if (full) {
            const f32x4 raw = *reinterpret_cast<const f32x4*>(ap);
            a[0] = raw[0]; a[1] = raw[1]; a[2] = raw[2]; a[3] = raw[3];
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (n + e < p.N) ? ap[e] : 0.f;
          }
        } else {
          const bf16_t* ap = reinterpret_cast<const bf16_t*>(p.aux) + (size_t)m * p.ldaux + n;
          if (full) {
            const u32x2 raw = *reinterpret_cast<const u32x2*>(ap);
            a[0] = __uint_as_float(raw[0] << 16); a[1] = __uint_as_float(raw[0] & 0xffff0000u);
            a[2] = __uint_as_float(raw[1] << 16); a[3] = __uint_as_float(raw[1] & 0xffff0000u);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) a[e] = (n + e < p.N) ? bf2f(ap[e]) : 0.f;
          }
        }
        if constexpr (EPI == UNIMM_EPI_BIAS_DROP_RESID) {
          if (p.drop.thr != 0u) {
            const uint32_t idx = (uint32_t)m * (uint32_t)p.N + (uint32_t)n;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = drop_apply(p.drop, idx + e, v[e]);
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += a[e];
        } else if constexpr (EPI == UNIMM_EPI_DGELU) {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] *= gelu_erf_grad(a[e]);
        } else {
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] += a[e];
        }
      }
      if constexpr (EPI == UNIMM_EPI_BIAS_GELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) { u[e] = v[e]; v[e] = gelu_erf(v[e]); }
        if (p.out2 != nullptr) {
          bf16_t* up = p.out2 + (size_t)m * p.ldo + n;
          if (full) *reinterpret_cast<u32x2*>(up) = u32x2{pack2bf(u[0], u[1]), pack2bf(u[2], u[3])};
          else
            for (int e = 0; e < 4; ++e) if (n + e < p.N) up[e] = f2bf(u[e]);
        }
      }
      if constexpr (EPI == UNIMM_EPI_BIAS_RELU) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
      if constexpr (OUT_F32) {
        float* op = reinterpret_cast<float*>(p.out) + (size_t)m * p.ldo + n;
        if (full) *reinterpret_cast<f32x4*>(op) = f32x4{v[0], v[1], v[2], v[3]};
        else
          for (int e = 0; e < 4; ++e) if (n + e < p.N) op[e] = v[e];
      } else {
        bf16_t* op = reinterpret_cast<bf16_t*>(p.out) + (size_t)m * p.ldo + n;
        if (full) *reinterpret_cast<u32x2*>(op) = u32x2{pack2bf(v[0], v[1]), pack2bf(v[2], v[3])};
        else
          for (int e = 0; e < 4; ++e) if (n + e < p.N) op[e] = f2bf(v[e]);
      }
    }
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
  int M, N, K, lddy, ldx, lddw, rows_per_split;
};

// [64 m][128 c] bf16 tile, 256-B rows = 16 chunks of 16 B.  A transposed read's 32-lane half touches
// 8 rows {b..b+3, b+8..b+11} x one aligned chunk pair; tn_swz moves each of those rows to its own
// chunk pair, so the half-wave covers all 64 banks exactly once.
__device__ __forceinline__ int tn_swz(int row) { return ((row & 3) | ((row >> 1) & 4)) << 1; }

__device__ __forceinline__ void stage_tile_tn(const bf16_t* __restrict__ g, int ld, int m0, int mend, int c0,
                                              int ncols, char* lds_tile, int wave, int lane) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int row = r * 16 + wave * 4 + (lane >> 4);  // 4 rows of 256 B per wave-instruction
    const int chunk = (lane & 15) ^ tn_swz(row);
    int gm = m0 + row;
    int gc = c0 + chunk * 8;
    // rows past the end re-read the last row (masked to zero in the fragment); columns past
    // round_up(ncols, 8) re-read the last readable chunk (they only feed outputs that are never stored)
    gm = gm < mend ? gm : mend - 1;
    const int cmax = ((ncols + 7) & ~7) - 8;
    gc = gc <= cmax ? gc : cmax;
    const bf16_t* src = g + (size_t)gm * ld + gc;
    __builtin_amdgcn_global_load_lds(GLB_PTR(src), LDS_PTR(lds_tile + (r * 16 + wave * 4) * 256), 16, 0, 0);
  }
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

__global__ __launch_bounds__(256, 2) void gemm_tn_kernel(GemmTnParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
  const int nbn = (p.N + 127) / 128, nbk = (p.K + 127) / 128;
  const int ntile = nbn * nbk;
  const int lid = xcd_remap(blockIdx.x, gridDim.x);
  const int split = lid / ntile;
  const int tile = lid - split * ntile;
  const int tn = tile / nbk, tk = tile - tn * nbk;
  const int n0 = tn * 128, k0 = tk * 128;
  const int mbeg = split * p.rows_per_split;
  int mend = mbeg + p.rows_per_split;
  mend = mend < p.M ? mend : p.M;
  if (mbeg >= mend) return;
  const int wn = wave >> 1, wk = wave & 1;

  f32x4 acc[4][4];  // [n-subtile i][k-subtile j]
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  // bias gradient = column sums of DY: one extra MFMA against an all-ones operand, only in the first
  // k-tile column and only in the waves that own k-subtile 0 (DY is already in LDS for the product)
  const bool do_bias = p.dbias != nullptr && tk == 0 && wk == 0;
  f32x4 accb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) accb[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 ones;
#pragma unroll
  for (int e = 0; e < 8; ++e) ones[e] = (__bf16)1.0f;

  const int nsteps = (mend - mbeg + TK - 1) / TK;

  stage_tile_tn(p.dy, p.lddy, mbeg, mend, n0, p.N, smem, wave, lane);
  stage_tile_tn(p.x, p.ldx, mbeg, mend, k0, p.K, smem + TN_TILE_BYTES, wave, lane);
  __builtin_amdgcn_s_waitcnt(0x0f70);
  __syncthreads();

  for (int t = 0; t < nsteps; ++t) {
    const int cur = t & 1;
    const int mt = mbeg + t * TK;
    if (t + 1 < nsteps) {
      char* nxt = smem + (cur ^ 1) * 2 * TN_TILE_BYTES;
      stage_tile_tn(p.dy, p.lddy, mt + TK, mend, n0, p.N, nxt, wave, lane);
      stage_tile_tn(p.x, p.ldx, mt + TK, mend, k0, p.K, nxt + TN_TILE_BYTES, wave, lane);
    }
    const char* ta = smem + cur * 2 * TN_TILE_BYTES;
    const char* tb = ta + TN_TILE_BYTES;
    const int valid = mend - mt;  // rows of this step that exist (>= 1); others must contribute 0
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      bf16x8 fa[4], fb[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fa[i] = read_frag_tr(ta, wn * 64 + i * 16, ks * 32, lane);
#pragma unroll
      for (int j = 0; j < 4; ++j) fb[j] = read_frag_tr(tb, wk * 64 + j * 16, ks * 32, lane);
      if (valid < TK) {  // ragged tail: zero the A-side elements of rows past the end
        const int rbase = ks * 32 + 8 * (lane >> 4);
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int e = 0; e < 8; ++e)
            if (rbase + e >= valid) fa[i][e] = (__bf16)0.0f;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], fb[j], acc[i][j], 0, 0, 0);
      if (do_bias) {
#pragma unroll
        for (int i = 0; i < 4; ++i) accb[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[i], ones, accb[i], 0, 0, 0);
      }
    }
    __builtin_amdgcn_s_waitcnt(0x0f70);
    __syncthreads();
  }

  if (do_bias && (lane & 15) == 0) {   // every column of accb holds the same row sums
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + e;
        if (n < p.N) atomicAdd(p.dbias + n, accb[i][e]);
      }
  }

  // D[n][k]: lane holds k = .. + (lane&15) (column), n = .. + 4*(lane>>4) + e (rows)
#pragma unroll
  for (int i = 0; i < 4; ++i) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int n = n0 + wn * 64 + i * 16 + 4 * (lane >> 4) + e;
      if (n >= p.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + wk * 64 + j * 16 + (lane & 15);
        if (k < p.K) atomicAdd(p.dw + (size_t)n * p.lddw + k, acc[i][j][e]);
      }
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Launch profiler (bench.py's `roofline` block): when enabled, every GEMM launch is bracketed by HIP
// events on the launch stream; unimm_prof_collect() sums elapsed time and algorithmic FLOPs per kernel
// variant.  Off by default (no events, no overhead).
// ------------------------------------------------------------------------------------------------
struct ProfRec { hipEvent_t a, b; int variant; double flops; };
constexpr int PROF_MAX = 1 << 16;
constexpr int PROF_VARIANTS = 16;   // 0..11: gemm_nt epi*2+out_f32 ; 12: gemm_tn
bool g_prof_on = false;
ProfRec* g_prof = nullptr;
int g_prof_n = 0;

inline ProfRec* prof_begin(int variant, double flops, hipStream_t s) {
  if (!g_prof_on || g_prof_n >= PROF_MAX) return nullptr;
  ProfRec* r = &g_prof[g_prof_n];
  if (r->a == nullptr) {
    if (hipEventCreate(&r->a) != hipSuccess || hipEventCreate(&r->b) != hipSuccess) return nullptr;
  }
  r->variant = variant; r->flops = flops;
  hipEventRecord(r->a, s);
  ++g_prof_n;
  return r;
}
inline void prof_end(ProfRec* r, hipStream_t s) { if (r) hipEventRecord(r->b, s); }

template <int EPI>
int launch_nt(const GemmNtParams& p, bool out_f32, hipStream_t s) {
  const int nwg = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  const size_t lds = 4 * TILE_BYTES;
  ProfRec* pr = prof_begin(EPI * 2 + (out_f32 ? 1 : 0), 2.0 * p.M * (double)p.N * p.K, s);
  if (out_f32) hipLaunchKernelGGL((gemm_nt_kernel<EPI, true>), dim3(nwg), dim3(256), lds, s, p);
  else hipLaunchKernelGGL((gemm_nt_kernel<EPI, false>), dim3(nwg), dim3(256), lds, s, p);
  prof_end(pr, s);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

}  // namespace

extern "C" int unimm_gemm_nt(const unimm_gemm_nt_args* a, void* stream) {
  if (a == nullptr || a->x == nullptr || a->w == nullptr || a->out == nullptr) return UNIMM_E_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0 || (a->K % BK) != 0) return UNIMM_E_SHAPE;
  if ((a->ldx % 8) || (a->ldw % 8) || a->ldx < a->K || a->ldw < a->K || a->ldo < a->N) return UNIMM_E_ALIGN;
  if (((uintptr_t)a->x | (uintptr_t)a->w | (uintptr_t)a->out) & 15) return UNIMM_E_ALIGN;
  if (a->out_f32 ? (a->ldo % 4) : (a->ldo % 4)) return UNIMM_E_ALIGN;
  const bool needs_aux = a->epilogue == UNIMM_EPI_BIAS_DROP_RESID || a->epilogue == UNIMM_EPI_DGELU ||
                         a->epilogue == UNIMM_EPI_ADD;
  if (needs_aux && (a->aux == nullptr || (a->ldaux % 4) || a->ldaux < a->N)) return UNIMM_E_ARG;
  if (a->epilogue == UNIMM_EPI_BIAS_DROP_RESID && !a->out_f32) return UNIMM_E_ARG;  // residual stream is fp32
  GemmNtParams p;
  p.x = (const bf16_t*)a->x; p.w = (const bf16_t*)a->w; p.bias = a->bias; p.aux = a->aux;
  p.out = a->out; p.out2 = (bf16_t*)a->out2;
  p.M = a->M; p.N = a->N; p.K = a->K; p.ldx = a->ldx; p.ldw = a->ldw; p.ldaux = a->ldaux; p.ldo = a->ldo;
  p.drop.key = a->drop_key; p.drop.thr = a->drop_thr; p.drop.scale = a->drop_scale;
  hipStream_t s = (hipStream_t)stream;
  const bool f32 = a->out_f32 != 0;
  switch (a->epilogue) {
    case UNIMM_EPI_BIAS: return launch_nt<UNIMM_EPI_BIAS>(p, f32, s);
    case UNIMM_EPI_BIAS_GELU: return launch_nt<UNIMM_EPI_BIAS_GELU>(p, f32, s);
    case UNIMM_EPI_BIAS_DROP_RESID: return launch_nt<UNIMM_EPI_BIAS_DROP_RESID>(p, f32, s);
    case UNIMM_EPI_BIAS_RELU: return launch_nt<UNIMM_EPI_BIAS_RELU>(p, f32, s);
    case UNIMM_EPI_DGELU: return launch_nt<UNIMM_EPI_DGELU>(p, f32, s);
    case UNIMM_EPI_ADD: return launch_nt<UNIMM_EPI_ADD>(p, f32, s);
    default: return UNIMM_E_ARG;
  }
}

extern "C" int unimm_gemm_tn(const unimm_gemm_tn_args* a, void* stream) {
  if (a == nullptr || a->dy == nullptr || a->x == nullptr || a->dw == nullptr) return UNIMM_E_ARG;
  if (a->M <= 0 || a->N <= 0 || a->K <= 0) return UNIMM_E_SHAPE;
  if ((a->lddy % 8) || (a->ldx % 8) || a->lddy < a->N || a->ldx < a->K || a->lddw < a->K) return UNIMM_E_ALIGN;
  if (((uintptr_t)a->dy | (uintptr_t)a->x) & 15) return UNIMM_E_ALIGN;
  GemmTnParams p;
  p.dy = (const bf16_t*)a->dy; p.x = (const bf16_t*)a->x; p.dw = a->dw; p.dbias = a->dbias;
  p.M = a->M; p.N = a->N; p.K = a->K; p.lddy = a->lddy; p.ldx = a->ldx; p.lddw = a->lddw;
  const int ntile = ((a->N + 127) / 128) * ((a->K + 127) / 128);
  // enough workgroups for ~2 per CU, but keep >= 512 reduction rows per split
  int splits = (512 + ntile - 1) / ntile;
  const int max_splits = (a->M + 511) / 512;
  if (splits > max_splits) splits = max_splits;
  if (splits < 1) splits = 1;
  int rps = (a->M + splits - 1) / splits;
  rps = ((rps + TK - 1) / TK) * TK;
  splits = (a->M + rps - 1) / rps;
  p.rows_per_split = rps;
  ProfRec* pr = prof_begin(12, 2.0 * a->M * (double)a->N * a->K, (hipStream_t)stream);
  hipLaunchKernelGGL(gemm_tn_kernel, dim3(ntile * splits), dim3(256), 4 * TN_TILE_BYTES, (hipStream_t)stream, p);
  prof_end(pr, (hipStream_t)stream);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_prof_enable(int32_t on) {
  if (on && g_prof == nullptr) {
    g_prof = (ProfRec*)calloc(PROF_MAX, sizeof(ProfRec));
    if (g_prof == nullptr) return UNIMM_E_HIP;
  }
  g_prof_on = on != 0;
  g_prof_n = 0;
  return UNIMM_OK;
}

// Synchronises the recorded events (call after the timed region) and fills, per variant:
// ms[v] = summed launch durations, flops[v] = summed algorithmic FLOPs, count[v] = launches.
extern "C" int unimm_prof_collect(double* ms, double* flops, int32_t* count, int32_t nvar) {
  if (!ms || !flops || !count || nvar < PROF_VARIANTS) return UNIMM_E_ARG;
  for (int v = 0; v < nvar; ++v) { ms[v] = 0.0; flops[v] = 0.0; count[v] = 0; }
  for (int i = 0; i < g_prof_n; ++i) {
    ProfRec& r = g_prof[i];
    if (hipEventSynchronize(r.b) != hipSuccess) return UNIMM_E_HIP;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) return UNIMM_E_HIP;
    ms[r.variant] += t; flops[r.variant] += r.flops; count[r.variant] += 1;
  }
  g_prof_n = 0;
  return UNIMM_OK;
}
