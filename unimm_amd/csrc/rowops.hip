// HBM-bound row kernels of the UniMM-UL hot path (gfx950): mask bit-packing, LayerNorm fwd/bwd,
// embedding gather+LayerNorm fwd/bwd, column sums (bias gradients), dtype casts / transposed weight
// copies, region-feature packing.  All are one-wave-per-row or flat streaming kernels
// with 16-byte vector accesses; none is MFMA work, their roofline is HBM bandwidth.
#include "common.h"
#include "rows.h"

namespace {

// ------------------------------------------------------------------------------------------------
// mask pack: any 0/1 mask tensor [rows, T] -> bit words [rows, ceil(T/32)]   (SURVEY K13;
// replaces the fp32 (1-m)*-10000 tensors of models/vilbert_dialog.py:1415-1431)
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void mask_pack_ballot_kernel(const T* __restrict__ m, uint32_t* __restrict__ out, size_t ngroups) {
  const size_t wave = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const size_t nwaves = (size_t)gridDim.x * (blockDim.x >> 6);
  const int lane = threadIdx.x & 63;
  for (size_t g = wave; g < ngroups; g += nwaves) {
    const bool bit = m[g * 64 + lane] != (T)0;
    const unsigned long long bal = __ballot(bit);
    if (lane == 0) {
      out[2 * g] = (uint32_t)bal;
      out[2 * g + 1] = (uint32_t)(bal >> 32);
    }
  }
}

template <typename T>
__global__ void mask_pack_generic_kernel(const T* __restrict__ m, uint32_t* __restrict__ out, size_t rows, int t,
                                         int nw) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * nw) return;
  const size_t row = i / nw;
  const int w = (int)(i % nw);
  uint32_t bits = 0;
  for (int j = 0; j < 32; ++j) {
    const int c = w * 32 + j;
    if (c < t && m[row * t + c] != (T)0) bits |= 1u << j;
  }
  out[i] = bits;
}

// ------------------------------------------------------------------------------------------------
// Packed text / co-attention masks straight from per-sequence descriptors: what
// utils/data_utils.py:199-210 (generative) and :391-396 (discriminative) write as dense [T, T] int64
// matrices, as words.  L = tokens up to and including the answer's [SEP], n = answer length + 1
// (0 in the discriminative regime), c = L - n:
//   generative:  row 0 -> [0, L+n);  rows [1, c) -> [1, c);  rows [c, L) -> [1, row];
//                copy rows r in [L, L+n) -> [1, c + (r - L)) plus r itself;  rows >= L+n -> empty
//   discriminative: rows [0, L) -> [0, L)
//   co-attention (one row per sequence): generative [1, c), discriminative [0, L)
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint32_t range_bits(int lo, int hi, int w) {   // bits of [lo, hi) inside word w
  const int a = lo - 32 * w, b = hi - 32 * w;
  const int x = a < 0 ? 0 : a, y = b > 32 ? 32 : b;
  if (x >= y) return 0u;
  const uint32_t upto_y = y == 32 ? 0xffffffffu : ((1u << y) - 1u);
  return upto_y & ~((1u << x) - 1u);
}

__global__ void mask_synth_kernel(const int* __restrict__ mode, const int* __restrict__ len, const int* __restrict__ nans,
                                  uint32_t* __restrict__ text, uint32_t* __restrict__ co, int B, int T) {
  const int nw = (T + 31) >> 5;
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * T * nw) return;
  const int w = (int)(i % nw), row = (int)((i / nw) % T), b = (int)(i / ((size_t)nw * T));
  const int L = len[b], n = nans[b], c = L - n;
  uint32_t bits;
  if (mode[b] == 0) {
    bits = row < L ? range_bits(0, L, w) : 0u;
  } else {
    if (row == 0) bits = range_bits(0, L + n, w);
    else if (row < c) bits = range_bits(1, c, w);
    else if (row < L) bits = range_bits(1, row + 1, w);
    else if (row < L + n) bits = range_bits(1, c + (row - L), w) | range_bits(row, row + 1, w);
    else bits = 0u;
  }
  bits &= range_bits(0, T, w);
  text[i] = bits;
  if (row == 0) co[(size_t)b * nw + w] = (mode[b] == 0 ? range_bits(0, L, w) : range_bits(1, c, w)) & range_bits(0, T, w);
}

template <typename T>
int mask_pack_impl(const void* m, uint32_t* out, size_t rows, int t, hipStream_t s) {
  const int nw = (t + 31) / 32;
  if (t % 64 == 0) {
    const size_t ngroups = rows * (size_t)(t / 64);
    int blocks = (int)((ngroups + 3) / 4);
    blocks = blocks > 4096 ? 4096 : blocks;
    hipLaunchKernelGGL(mask_pack_ballot_kernel<T>, dim3(blocks), dim3(256), 0, s, (const T*)m, out, ngroups);
  } else {
    const size_t n = rows * nw;
    hipLaunchKernelGGL(mask_pack_generic_kernel<T>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const T*)m,
                       out, rows, t, nw);
  }
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

// ------------------------------------------------------------------------------------------------
// Unpadded-schedule plan on the device.  The engine runs the text stream on the valid prefix of every sequence and
// decodes only the labelled rows; both need counts on the host (allocation sizes, grid sizes), which used to cost ~25
// eager torch kernels over the dense [B,T,T] masks, two device->host round trips at the start of forward and a
// third and fourth in the middle of it (torch.nonzero for the labelled rows and the count of masked regions, each of which
// drained the whole encoder forward before the host could go on).
//   plan_lengths: one workgroup per sequence, lane = token.  header[b] = valid prefix length (a token is valid when it
//     attends something, is attended by a token or a region, carries a label or a weight; >= 1), header[B+b] = number of
//     rows the MLM head decodes (weight != 0 when weights are given, else label != -1), header[2B..2B+1] = the bits of the
//     two NSP class weights when they live on the device, header[2B+2+b] = regions of the sequence that enter the
//     masked-region loss (image_label == 1).  ONE device->host copy of the header follows: the step's only host sync.
//   plan_build: from the header, everything else without the host: offsets, the packed-row -> padded-row map and its
//     inverse, and the decoded rows' positions / packed indices / labels / weights in row order.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void plan_lengths_kernel(const uint32_t* __restrict__ tw, int t_qs, int t_bs,
                                                           const uint32_t* __restrict__ cw, int c_qs, int c_bs, int R,
                                                           const int32_t* __restrict__ labels,
                                                           const int32_t* __restrict__ weights, const float* __restrict__ nspw,
                                                           const int32_t* __restrict__ img_label,
                                                           int B, int T, int nw, int32_t* __restrict__ header) {
  __shared__ uint32_t col[8];
  __shared__ int red_len[4], red_cnt[4], red_img[4];
  const int b = blockIdx.x, t = threadIdx.x;
  if (t < 8) col[t] = 0u;
  __syncthreads();
  bool valid = false;
  bool empty_row = false;        // dense mask: this token's own mask row has no bit set
  if (tw != nullptr && t < T) {
    if (t_qs == 0) {
      valid = (tw[(size_t)b * t_bs + (t >> 5)] >> (t & 31)) & 1u;           // key-padding mask: the bit itself
    } else {
      for (int w = 0; w < nw; ++w) {
        const uint32_t x = tw[(size_t)b * t_bs + (size_t)t * t_qs + w];
        if (x != 0u) { valid = true; atomicOr(&col[w], x); }                // attends something | marks attended keys
      }
      empty_row = !valid;
    }
  }
  if (cw != nullptr) {
    if (c_qs == 0) {
      if (t < nw) atomicOr(&col[t], cw[(size_t)b * c_bs + t]);
    } else if (t < R) {
      for (int w = 0; w < nw; ++w) {
        const uint32_t x = cw[(size_t)b * c_bs + (size_t)t * c_qs + w];
        if (x != 0u) atomicOr(&col[w], x);
      }
    }
  }
  __syncthreads();
  bool sel = false;
  if (t < T) {
    valid = valid || ((col[t >> 5] >> (t & 31)) & 1u);
    const int lab = labels != nullptr ? labels[(size_t)b * T + t] : -1;
    const int wt = weights != nullptr ? weights[(size_t)b * T + t] : 0;
    valid = valid || lab != -1 || wt != 0;
    sel = labels != nullptr && (weights != nullptr ? wt != 0 : lab != -1);
  }
  // A token that is valid (labelled, weighted or attended) although its OWN mask row is empty attends, in the
  // reference, every one of the T keys uniformly (softmax of raw scores - 10000, models/vilbert_dialog.py:1418), the
  // padding rows' K / V included: such a sequence must run at its full length (the reference's own encoders never
  // produce one: utils/data_utils.py:199-210, :354; only user-supplied masks can).
  int len = valid ? (empty_row ? T : t + 1) : 0;
  int cnt = sel ? 1 : 0;
  int nim = (img_label != nullptr && t < R && img_label[(size_t)b * R + t] == 1) ? 1 : 0;   // regions in the masked-region loss
  for (int o = 32; o > 0; o >>= 1) {
    len = max(len, __shfl_xor(len, o, 64)); cnt += __shfl_xor(cnt, o, 64); nim += __shfl_xor(nim, o, 64);
  }
  if ((t & 63) == 0) { red_len[t >> 6] = len; red_cnt[t >> 6] = cnt; red_img[t >> 6] = nim; }
  __syncthreads();
  if (t == 0) {
    header[b] = max(1, max(max(red_len[0], red_len[1]), max(red_len[2], red_len[3])));
    header[B + b] = red_cnt[0] + red_cnt[1] + red_cnt[2] + red_cnt[3];
    header[2 * B + 2 + b] = red_img[0] + red_img[1] + red_img[2] + red_img[3];
  }
  if (b == 0 && t < 2 && nspw != nullptr) header[2 * B + t] = __float_as_int(nspw[t]);
}

__global__ __launch_bounds__(256) void plan_build_kernel(const int32_t* __restrict__ header, const int32_t* __restrict__ labels,
                                                         const int32_t* __restrict__ weights, int B, int T,
                                                         int32_t* __restrict__ off, int32_t* __restrict__ lens,
                                                         int64_t* __restrict__ rows, int64_t* __restrict__ inv,
                                                         int32_t* __restrict__ lm_pos, int32_t* __restrict__ lm_idx,
                                                         int32_t* __restrict__ lm_lab, int32_t* __restrict__ lm_w,
                                                         int rows_cap, int lm_cap, int32_t* __restrict__ dims_i,
                                                         float* __restrict__ dims_f, int32_t* __restrict__ order) {
  __shared__ int s_off, s_lmoff, wave_cnt[4];
  const int b = blockIdx.x, t = threadIdx.x;
  if (b == (int)gridDim.x - 1) {
    // The extra workgroup: totals of the step into device memory (kernels of a replayed launch sequence read their row
    // counts and loss denominators from here, their launch arguments only hold CAPACITIES), and safe values in the unused
    // tail of every index list so that row-independent kernels may run over the whole capacity.
    __shared__ int tot[3][4];
    int a = 0, c = 0, g = 0;
    for (int i = t; i < B; i += 256) { a += header[i]; c += header[B + i]; g += header[2 * B + 2 + i]; }
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); g += __shfl_xor(g, o, 64); }
    if ((t & 63) == 0) { tot[0][t >> 6] = a; tot[1][t >> 6] = c; tot[2][t >> 6] = g; }
    __syncthreads();
    const int mv = tot[0][0] + tot[0][1] + tot[0][2] + tot[0][3];
    const int nlm = tot[1][0] + tot[1][1] + tot[1][2] + tot[1][3];
    const int nimg = tot[2][0] + tot[2][1] + tot[2][2] + tot[2][3];
    // The lists are sized by the CALLER's counts (launch arguments; under graph replay: the header the capture was sized
    // with); the counts above come from the batch itself.  A batch that does not fit -- a stale `plan_header` handed to a
    // replayed step -- must not write past the lists: every index below is bounded by its capacity, the row counts the other
    // kernels read are clamped to it, and the step is marked (dims_i[3] = 1, NaN loss denominators: the losses come out NaN).
    const bool over = (rows != nullptr && mv > rows_cap) || (lm_pos != nullptr && nlm > lm_cap);
    if (t == 0 && dims_i != nullptr) {
      dims_i[0] = rows != nullptr ? min(mv, rows_cap) : mv;
      dims_i[1] = lm_pos != nullptr ? min(nlm, lm_cap) : 0; dims_i[2] = nimg;
      dims_i[3] = over ? 1 : 0;
      dims_f[0] = over ? __builtin_nanf("") : 1.0f / (float)max(nlm, 1);
      dims_f[1] = over ? __builtin_nanf("") : (nimg > 0 ? 1.0f / (float)nimg : __builtin_inff());   // the reference divides by max(n, 0) (:1574)
    }
    if (rows != nullptr) for (int i = mv + t; i < rows_cap; i += 256) rows[i] = 0;
    if (lm_pos != nullptr)
      for (int i = nlm + t; i < lm_cap; i += 256) { lm_pos[i] = 0; lm_idx[i] = 0; lm_lab[i] = -1; lm_w[i] = 0; }
    // the sequences by length, longest first (rank sort; B is a few hundred): the item order of the attention launches
    if (order != nullptr)
      for (int i = t; i < B; i += 256) {
        const int li = header[i];
        int rank = 0;
        for (int j = 0; j < B; ++j) { const int lj = header[j]; rank += (lj > li || (lj == li && j < i)) ? 1 : 0; }
        order[rank] = i;
      }
    return;
  }
  // exclusive prefix sums over the sequences before this one (B is a few hundred: one strided pass)
  int a = 0, c = 0;
  for (int i = t; i < b; i += 256) { a += header[i]; c += header[B + i]; }
  for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); c += __shfl_xor(c, o, 64); }
  __shared__ int pa[4], pc[4];
  if ((t & 63) == 0) { pa[t >> 6] = a; pc[t >> 6] = c; }
  __syncthreads();
  if (t == 0) { s_off = pa[0] + pa[1] + pa[2] + pa[3]; s_lmoff = pc[0] + pc[1] + pc[2] + pc[3]; }
  __syncthreads();
  int len = header[b], o0 = s_off;
  if (rows != nullptr && o0 + len > rows_cap) {     // does not fit the packed layout (see `over` above): stay inside it
    o0 = min(o0, rows_cap - 1);
    len = max(1, min(len, rows_cap - o0));
  }
  if (t == 0) { off[b] = o0; lens[b] = len; }
  if (t < T) {
    const bool in = t < len;
    if (in && rows != nullptr) rows[o0 + t] = (int64_t)b * T + t;
    if (inv != nullptr) inv[(size_t)b * T + t] = in ? (int64_t)(o0 + t) : (int64_t)-1;
  }
  if (labels == nullptr || lm_pos == nullptr) return;
  const int lab = t < T ? labels[(size_t)b * T + t] : -1;
  const int wt = (weights != nullptr && t < T) ? weights[(size_t)b * T + t] : 0;
  const bool sel = t < T && (weights != nullptr ? wt != 0 : lab != -1);
  const unsigned long long bal = __ballot(sel);
  const int lane = t & 63, wv = t >> 6;
  if (lane == 0) wave_cnt[wv] = __popcll(bal);
  __syncthreads();
  if (sel) {
    int r = __popcll(bal & ((1ull << lane) - 1ull));
    for (int w = 0; w < wv; ++w) r += wave_cnt[w];
    const int dst = s_lmoff + r;                    // row order: by sequence, then by position (what torch.nonzero gave)
    if (dst < lm_cap) {
      lm_pos[dst] = b * T + t;
      lm_idx[dst] = rows != nullptr ? min(o0 + t, rows_cap - 1) : o0 + t;   // a decoded row is valid by construction (t < len)
      lm_lab[dst] = lab;
      lm_w[dst] = weights != nullptr ? wt : 1;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// LayerNorm forward (torch.nn.LayerNorm, eps 1e-12; models/vilbert_dialog.py:279,425,468,...)
// optional dropout on the output (embedding dropouts :355, :1491)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y32,
                                                            bf16_t* __restrict__ y,
                                                            float* __restrict__ mean_o, float* __restrict__ rstd_o, int M,
                                                            int H, float eps, DropoutArg drop) {
  drop_resolve(drop);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * (blockDim.x >> 6);
  Row8 g, b;
  load_vec_f32(gamma, H, lane, g);
  load_vec_f32(beta, H, lane, b);
  // software prefetch: the next row of this wave is requested before the current one is reduced, so every wave keeps
  // two rows in flight (the kernel is a pure HBM stream: 6-10 bytes per element, ~2 us of latency to cover)
  Row8 xv;
  if (wave < M) load_vec_f32(x + (size_t)wave * H, H, lane, xv);
  for (int row = wave; row < M; row += nwaves) {
    Row8 nx;
    const int nrow = row + nwaves;
    if (nrow < M) load_vec_f32(x + (size_t)nrow * H, H, lane, nx);
    float mean, rstd;
    row_stats(xv, H, mean, rstd, eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const uint32_t kb = drop.thr != 0u ? drop_bits8(drop, (uint32_t)row, (uint32_t)H, (uint32_t)((lane + 64 * i) * 8)) : 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = (xv.v[i][j] - mean) * rstd * g.v[i][j] + b.v[i][j];
        if (drop.thr != 0u) v = ((kb >> j) & 1u) ? v * drop.scale : 0.f;
        xv.v[i][j] = v;
      }
    }
    if (y32 != nullptr) store_row_f32(y32 + (size_t)row * H, H, lane, xv);
    if (y != nullptr) store_row_bf16(y + (size_t)row * H, H, lane, xv);
    if (lane == 0 && mean_o != nullptr) { mean_o[row] = mean; rstd_o[row] = rstd; }
    if (nrow < M) xv = nx;
  }
}

// ------------------------------------------------------------------------------------------------
// LayerNorm backward.  dx = rstd * (g - mean(g) - xhat * mean(g*xhat)), g = dy*gamma.
// Also emits (a) dx_drop = dropout-masked dx for the dense branch when the forward applied
// dropout before the residual add (:424, :467, ...), and (b) per-block column partials of
// dgamma, dbeta and dbias = colsum(dx_drop) that `colpartials_finish` adds into the gradient arena
// (two-stage, deterministic; no same-address atomics from 1000+ waves).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const bf16_t* __restrict__ dy, const float* __restrict__ x,
                                                            const float* __restrict__ mean_i,
                                                            const float* __restrict__ rstd_i,
                                                            const float* __restrict__ gamma, bf16_t* __restrict__ dx,
                                                            bf16_t* __restrict__ dx_drop, float* __restrict__ partials,
                                                            int M, int H, DropoutArg drop, DropoutArg out_drop,
                                                            const int32_t* __restrict__ m_dev) {
  __shared__ float red[4 * 1024];  // [wave][col], reused for each of the three quantities
  drop_resolve(drop);
  drop_resolve(out_drop);
  if (m_dev != nullptr) M = min(M, m_dev[0]);     // M is a capacity: rows past the real count never enter the column sums
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  const int nwaves = gridDim.x * 4;
  Row8 g;
  load_vec_f32(gamma, H, lane, g);
  Row8 dg, db, dbias;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg.v[i][j] = 0.f; db.v[i][j] = 0.f; dbias.v[i][j] = 0.f; }
  const float invH = 1.0f / (float)H;
  // software prefetch (as the forward): the raw 16-byte pieces of the wave's NEXT row are requested before the current
  // row is reduced; bf16 pieces are unpacked only when the row is consumed (24 extra registers instead of 32)
  u32x4 pdy[MAXC];
  f32x4 pxa[MAXC], pxb[MAXC];
  float pmean = 0.f, prstd = 0.f;
#define UNIMM_LNB_PREFETCH(r_)                                                                                 \
  {                                                                                                            \
    _Pragma("unroll") for (int i = 0; i < MAXC; ++i) {                                                         \
      const int c = lane + 64 * i;                                                                             \
      if (c * 8 < H) {                                                                                         \
        pdy[i] = *reinterpret_cast<const u32x4*>(dy + (size_t)(r_) * H + c * 8);                               \
        pxa[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(r_) * H + c * 8);                                \
        pxb[i] = *reinterpret_cast<const f32x4*>(x + (size_t)(r_) * H + c * 8 + 4);                            \
      }                                                                                                        \
    }                                                                                                          \
    pmean = mean_i[r_]; prstd = rstd_i[r_];                                                                    \
  }
  if (wave < M) UNIMM_LNB_PREFETCH(wave)
  for (int row = wave; row < M; row += nwaves) {
    Row8 dyv, xv;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      if ((lane + 64 * i) * 8 < H) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          dyv.v[i][2 * j] = __uint_as_float(pdy[i][j] << 16);
          dyv.v[i][2 * j + 1] = __uint_as_float(pdy[i][j] & 0xffff0000u);
          xv.v[i][j] = pxa[i][j];
          xv.v[i][4 + j] = pxb[i][j];
        }
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { dyv.v[i][j] = 0.f; xv.v[i][j] = 0.f; }
      }
    }
    const float mean = pmean, rstd = prstd;
    if (row + nwaves < M) UNIMM_LNB_PREFETCH(row + nwaves)
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      // forward applied dropout AFTER this LayerNorm (embeddings)
      const uint32_t kbo = out_drop.thr != 0u ? drop_bits8(out_drop, (uint32_t)row, (uint32_t)H, (uint32_t)((lane + 64 * i) * 8)) : 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = dyv.v[i][j];
        if (out_drop.thr != 0u) d = ((kbo >> j) & 1u) ? d * out_drop.scale : 0.f;
        const float xh = (xv.v[i][j] - mean) * rstd;
        const float gg = d * g.v[i][j];
        dg.v[i][j] += d * xh;
        db.v[i][j] += d;
        s1 += gg;
        s2 += gg * xh;
        dyv.v[i][j] = gg;
        xv.v[i][j] = xh;
      }
    }
    s1 = wave_sum(s1) * invH;
    s2 = wave_sum(s2) * invH;
    Row8 dd;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const uint32_t kb = drop.thr != 0u ? drop_bits8(drop, (uint32_t)row, (uint32_t)H, (uint32_t)((lane + 64 * i) * 8)) : 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rstd * (dyv.v[i][j] - s1 - xv.v[i][j] * s2);
        dyv.v[i][j] = v;
        float vd = v;
        if (drop.thr != 0u) vd = ((kb >> j) & 1u) ? v * drop.scale : 0.f;
        dd.v[i][j] = vd;
        dbias.v[i][j] += vd;
      }
    }
    store_row_bf16(dx + (size_t)row * H, H, lane, dyv);
    if (dx_drop != nullptr) store_row_bf16(dx_drop + (size_t)row * H, H, lane, dd);
  }
#undef UNIMM_LNB_PREFETCH
  // block reduce the three column partials over the 4 waves, write [block][3][H]
  float* redf = red;
  for (int qn = 0; qn < 3; ++qn) {
    const Row8& src = qn == 0 ? dg : (qn == 1 ? db : dbias);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = (lane + 64 * i) * 8 + j;
        if (col < H) redf[wv * 1024 + col] = src.v[i][j];
      }
    __syncthreads();
    for (int col = threadIdx.x; col < H; col += 256)
      partials[((size_t)blockIdx.x * 3 + qn) * H + col] =
          redf[col] + redf[1024 + col] + redf[2048 + col] + redf[3072 + col];
  }
}

// out[q][col] += sum_blocks partials[block][q][col]  (q-th quantity goes to dst[q], NULL = skip).
// 1024 threads = 64 columns x 16 slices of the block range; LDS tree over the slices.
__global__ __launch_bounds__(1024) void colpartials_finish_kernel(const float* __restrict__ partials, int nblocks, int nq,
                                                                  int H, float* d0, float* d1, float* d2, float* d3) {
  __shared__ float red[16][64];
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  const int qn = blockIdx.y;
  float* dst = qn == 0 ? d0 : (qn == 1 ? d1 : (qn == 2 ? d2 : d3));
  float s = 0.f;
  if (col < H && dst != nullptr) {
    // 8 independent loads in flight per thread: the loop was a chain of 32 dependent L2 round trips (~10 us
    // for 4.7 MB); the partial order of the sum stays fixed (deterministic)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int b = sl;
    const size_t st = (size_t)16 * nq * H;
    const float* pp = partials + ((size_t)b * nq + qn) * H + col;
    for (; b + 7 * 16 < nblocks; b += 8 * 16, pp += 8 * st) {
      s0 += pp[0]; s1 += pp[st]; s2 += pp[2 * st]; s3 += pp[3 * st];
      s4 += pp[4 * st]; s5 += pp[5 * st]; s6 += pp[6 * st]; s7 += pp[7 * st];
    }
    for (; b < nblocks; b += 16, pp += st) s0 += pp[0];
    s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));
  }
  red[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && col < H && dst != nullptr) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    dst[col] += t;
  }
}

// The same reduction for up to UNIMM_FINISH_MAX pending row kernels in one launch (blockIdx.z = which): the engine
// defers the column-partials reductions of a block's LayerNorm backward calls to the end of the block, where nothing
// waits for them (94 launches of ~4 us per step otherwise, each between two dependent kernels of the critical path).
struct FinishGroup { unimm_finish_desc d[UNIMM_FINISH_MAX]; };

__global__ __launch_bounds__(1024) void colpartials_finish_grouped_kernel(FinishGroup g) {
  __shared__ float red[16][64];
  const unimm_finish_desc& d = g.d[blockIdx.z];
  const int H = d.H, nq = d.nq, nblocks = d.blocks;
  const int qn = blockIdx.y;
  if (blockIdx.x * 64 >= H || qn >= nq) return;                 // block-uniform
  const int cl = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + cl;
  float* dst = d.dst[qn];
  float s = 0.f;
  if (col < H && dst != nullptr) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f, s4 = 0.f, s5 = 0.f, s6 = 0.f, s7 = 0.f;
    int b = sl;
    const size_t st = (size_t)16 * nq * H;
    const float* pp = d.partials + ((size_t)b * nq + qn) * H + col;
    for (; b + 7 * 16 < nblocks; b += 8 * 16, pp += 8 * st) {
      s0 += pp[0]; s1 += pp[st]; s2 += pp[2 * st]; s3 += pp[3 * st];
      s4 += pp[4 * st]; s5 += pp[5 * st]; s6 += pp[6 * st]; s7 += pp[7 * st];
    }
    for (; b < nblocks; b += 16, pp += st) s0 += pp[0];
    s = ((s0 + s1) + (s2 + s3)) + ((s4 + s5) + (s6 + s7));       // same order as colpartials_finish_kernel
  }
  red[sl][cl] = s;
  __syncthreads();
  if (sl == 0 && col < H && dst != nullptr) {
    float t = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) t += red[k][cl];
    dst[col] += t;
  }
}

// ------------------------------------------------------------------------------------------------
// text embeddings: LN(word[id] + pos[pid] + type) with type ids >= tv routed to the 10-row extension
// table (models/vilbert_dialog.py:326-356), dropout after the LayerNorm.
// ------------------------------------------------------------------------------------------------
struct EmbArgs {
  const int32_t* ids; const int32_t* pos; const int32_t* typ;
  const float* word; const float* post; const float* type; const float* ext;   // fp32 master tables
  const float* gamma; const float* beta;
  int M, H, type_vocab;
  float eps;
  DropoutArg drop;
  const int32_t* m_dev;    // or NULL: rows actually present (the launch's M is then a capacity)
  const int64_t* rows;     // or NULL: packed row -> index into ids / pos / typ (the unpadded schedule's row map)
};

__device__ __forceinline__ void emb_gather(const EmbArgs& a, int row, int lane, Row8& x, int& id, int& pid, int& tt) {
  const size_t src = a.rows != nullptr ? (size_t)a.rows[row] : (size_t)row;
  id = a.ids[src]; pid = a.pos[src]; tt = a.typ[src];
  Row8 t1, t2;
  load_vec_f32(a.word + (size_t)id * a.H, a.H, lane, x);
  load_vec_f32(a.post + (size_t)pid * a.H, a.H, lane, t1);
  if (tt < a.type_vocab) load_vec_f32(a.type + (size_t)tt * a.H, a.H, lane, t2);
  else load_vec_f32(a.ext + (size_t)(tt - a.type_vocab) * a.H, a.H, lane, t2);
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) x.v[i][j] += t1.v[i][j] + t2.v[i][j];
}

__global__ __launch_bounds__(256) void embed_fwd_kernel(EmbArgs a, float* __restrict__ y32, bf16_t* __restrict__ y) {
  drop_resolve(a.drop);
  const int lane = threadIdx.x & 63;
  const int wave = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int nwaves = gridDim.x * 4;
  Row8 g, b;
  load_vec_f32(a.gamma, a.H, lane, g);
  load_vec_f32(a.beta, a.H, lane, b);
  const int M = a.m_dev != nullptr ? min(a.M, a.m_dev[0]) : a.M;
  for (int row = wave; row < M; row += nwaves) {
    Row8 x;
    int id, pid, tt;
    emb_gather(a, row, lane, x, id, pid, tt);
    float mean, rstd;
    row_stats(x, a.H, mean, rstd, a.eps);
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = (x.v[i][j] - mean) * rstd * g.v[i][j] + b.v[i][j];
        if (a.drop.thr != 0u) v = drop_apply(a.drop, (uint32_t)row, (uint32_t)a.H, (uint32_t)((lane + 64 * i) * 8 + j), v);
        x.v[i][j] = v;
      }
    store_row_f32(y32 + (size_t)row * a.H, a.H, lane, x);
    store_row_bf16(y + (size_t)row * a.H, a.H, lane, x);
  }
}

// backward: re-gathers x (nothing was saved), LayerNorm backward, scatter-add into the fp32 tables.
// The two regular type rows receive a colsum from every token -> per-block partials, not atomics.
template <typename DY>      // bf16_t (the bf16 path) or float (the fp32x3 mode's gradient stream)
__global__ __launch_bounds__(256) void embed_bwd_kernel(EmbArgs a, const DY* __restrict__ dy, float* __restrict__ dword,
                                                        float* __restrict__ dpos, float* __restrict__ dext,
                                                        float* __restrict__ partials) {
  __shared__ float red[4 * 1024];
  drop_resolve(a.drop);
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int wave = blockIdx.x * 4 + wv;
  const int nwaves = gridDim.x * 4;
  Row8 g;
  load_vec_f32(a.gamma, a.H, lane, g);
  Row8 dg, db, dt0, dt1;
#pragma unroll
  for (int i = 0; i < MAXC; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) { dg.v[i][j] = 0.f; db.v[i][j] = 0.f; dt0.v[i][j] = 0.f; dt1.v[i][j] = 0.f; }
  const float invH = 1.0f / (float)a.H;
  const int M = a.m_dev != nullptr ? min(a.M, a.m_dev[0]) : a.M;
  for (int row = wave; row < M; row += nwaves) {
    Row8 x, dyv;
    int id, pid, tt;
    emb_gather(a, row, lane, x, id, pid, tt);
    float mean, rstd;
    row_stats(x, a.H, mean, rstd, a.eps);
    if constexpr (sizeof(DY) == 4) load_vec_f32(reinterpret_cast<const float*>(dy) + (size_t)row * a.H, a.H, lane, dyv);
    else load_row_bf16(reinterpret_cast<const bf16_t*>(dy) + (size_t)row * a.H, a.H, lane, dyv);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float d = dyv.v[i][j];
        if (a.drop.thr != 0u) d = drop_apply(a.drop, (uint32_t)row, (uint32_t)a.H, (uint32_t)((lane + 64 * i) * 8 + j), d);
        const float xh = (x.v[i][j] - mean) * rstd;
        const float gg = d * g.v[i][j];
        dg.v[i][j] += d * xh;
        db.v[i][j] += d;
        s1 += gg; s2 += gg * xh;
        dyv.v[i][j] = gg; x.v[i][j] = xh;
      }
    s1 = wave_sum(s1) * invH;
    s2 = wave_sum(s2) * invH;
    float* wrow = dword + (size_t)id * a.H;
    float* prow = dpos + (size_t)pid * a.H;
    float* erow = (tt >= a.type_vocab) ? dext + (size_t)(tt - a.type_vocab) * a.H : nullptr;
    float* mine = red + wv * 1024;   // this wave's row, re-read lane-contiguously for the atomics
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
      const int c = lane + 64 * i;
      if (c * 8 >= a.H) continue;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float v = rstd * (dyv.v[i][j] - s1 - x.v[i][j] * s2);
        mine[c * 8 + j] = v;
        if (erow == nullptr) {
          if (tt == 0) dt0.v[i][j] += v;
          else dt1.v[i][j] += v;   // type_vocab_size is 2 (config/bert_base_6layer_6conect.json:11)
        }
      }
    }
    // one atomic wave-instruction = 256 contiguous bytes of one row (full-rate shape; a lane-strided
    // shape runs ~8x slower, MI355X_MICROARCH.md "Global float atomics")
    for (int col = lane; col < a.H; col += 64) {
      const float v = mine[col];
      atomicAdd(wrow + col, v);
      atomicAdd(prow + col, v);
      if (erow != nullptr) atomicAdd(erow + col, v);
    }
  }
  for (int qn = 0; qn < 4; ++qn) {
    const Row8& src = qn == 0 ? dg : (qn == 1 ? db : (qn == 2 ? dt0 : dt1));
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MAXC; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = (lane + 64 * i) * 8 + j;
        if (col < a.H) red[wv * 1024 + col] = src.v[i][j];
      }
    __syncthreads();
    for (int col = threadIdx.x; col < a.H; col += 256)
      partials[((size_t)blockIdx.x * 4 + qn) * a.H + col] = red[col] + red[1024 + col] + red[2048 + col] + red[3072 + col];
  }
}

// ------------------------------------------------------------------------------------------------
// column sums: db[N] += sum_m dY[m, :]   (bias gradients of the projections whose dY is produced by
// a GEMM / attention kernel rather than by layernorm_bwd)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colsum_kernel(const bf16_t* __restrict__ dy, float* __restrict__ db, int M, int N,
                                                     int ld) {
  __shared__ float red[4][64 * 8];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int col0 = (blockIdx.x * 64 + lane) * 8;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per;
  int r1 = r0 + rows_per;
  r1 = r1 < M ? r1 : M;
  float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (col0 < N) {
    for (int row = r0 + wv; row < r1; row += 4) {
      const u32x4 raw = *reinterpret_cast<const u32x4*>(dy + (size_t)row * ld + col0);
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[2 * j] += __uint_as_float(raw[j] << 16);
        acc[2 * j + 1] += __uint_as_float(raw[j] & 0xffff0000u);
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[wv][lane * 8 + j] = acc[j];
  __syncthreads();
  for (int i = threadIdx.x; i < 512; i += 256) {
    const int col = blockIdx.x * 512 + i;
    if (col < N) atomicAdd(db + col, red[0][i] + red[1][i] + red[2][i] + red[3][i]);
  }
}

// ------------------------------------------------------------------------------------------------
// casts
// ------------------------------------------------------------------------------------------------
__global__ void cast_f32_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst, size_t n) {
  size_t i = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 8;
  const size_t stride = (size_t)gridDim.x * blockDim.x * 8;
  for (; i + 8 <= n; i += stride) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(src + i);
    const f32x4 b = *reinterpret_cast<const f32x4*>(src + i + 4);
    *reinterpret_cast<u32x4*>(dst + i) =
        u32x4{pack2bf(a[0], a[1]), pack2bf(a[2], a[3]), pack2bf(b[0], b[1]), pack2bf(b[2], b[3])};
  }
  if (i < n)
    for (size_t j = i; j < n && j < i + 8; ++j) dst[j] = f2bf(src[j]);
}

// dst[c][r] (bf16, row stride ldd, columns >= R zero up to ldd) = src[r][c] (fp32 [R, C])
__global__ __launch_bounds__(256) void transpose_cast_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                                             int R, int C, int ldd) {
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  const int r0 = blockIdx.y * 32, c0 = blockIdx.x * 32;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < R && c < C) ? src[(size_t)r * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < C && r < ldd) dst[(size_t)c * ldd + r] = f2bf(tile[tx][ty + 8 * k]);
  }
}

// dst[c][r] = src[r][c], both bf16: src [R, C] with row stride lds, dst [C, ldd] (columns R..ldd-1 zero-filled).
// 64 x 64 tiles through LDS, 16-byte global accesses on both sides.  (The gradient of the decoder logits, [rows, vocab],
// is needed reduction-major by the weight-gradient kernel when it computes a few hundred rows x 768 over 30,522
// reduction steps: see Engine._decoder_dx.)
__global__ __launch_bounds__(256) void transpose_bf16_kernel(const bf16_t* __restrict__ src, bf16_t* __restrict__ dst,
                                                             int R, int C, int lds, int ldd) {
  __shared__ bf16_t tile[64][72];
  const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
  const int t = threadIdx.x;
#pragma unroll
  for (int k = 0; k < 2; ++k) {                     // 64 rows x 8 chunks of 8 columns
    const int id = t + 256 * k, r = id >> 3, cc = (id & 7) * 8;
    u32x4 v = {0u, 0u, 0u, 0u};
    if (r0 + r < R) {
      const bf16_t* sp = src + (size_t)(r0 + r) * lds + c0 + cc;
      if (c0 + cc + 7 < C) {
        v = *reinterpret_cast<const u32x4*>(sp);
      } else {
        bf16_t e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (c0 + cc + i < C) ? sp[i] : (bf16_t)0;
        v = *reinterpret_cast<const u32x4*>(e);
      }
    }
    *reinterpret_cast<u32x4*>(&tile[r][cc]) = v;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 2; ++k) {                     // 64 output rows (source columns) x 8 chunks of 8 source rows
    const int id = t + 256 * k, c = id >> 3, rr = (id & 7) * 8;
    if (c0 + c >= C || r0 + rr >= ldd) continue;
    bf16_t e[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) e[i] = tile[rr + i][c];
    *reinterpret_cast<u32x4*>(dst + (size_t)(c0 + c) * ldd + r0 + rr) = *reinterpret_cast<const u32x4*>(e);
  }
}

// out[i] (bf16) = slabs[0][i] + slabs[1][i] + ... in that order (fp32), i < n: the fixed-order sum of the per-chunk partial
// results of Engine._decoder_dx (a sum whose order does not depend on which workgroup finished first).
__global__ __launch_bounds__(256) void sum_slabs_bf16_kernel(const float* __restrict__ slabs, int S, size_t stride,
                                                             bf16_t* __restrict__ out, size_t n8) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n8; i += (size_t)gridDim.x * 256) {
    f32x4 a0 = {0.f, 0.f, 0.f, 0.f}, a1 = a0;
    for (int s = 0; s < S; ++s) {
      const f32x4* p = reinterpret_cast<const f32x4*>(slabs + (size_t)s * stride) + 2 * i;
      const f32x4 v0 = p[0], v1 = p[1];
      a0 += v0; a1 += v1;
    }
    u32x4 o = {pack2bf(a0[0], a0[1]), pack2bf(a0[2], a0[3]), pack2bf(a1[0], a1[1]), pack2bf(a1[2], a1[3])};
    *reinterpret_cast<u32x4*>(out + 8 * i) = o;
  }
}

// the same for a table of matrices in ONE launch (all transposed weight copies after an optimizer step:
// ~190 launches of a few microseconds each otherwise).  Block b belongs to the last entry with tile0 <= b.
__global__ __launch_bounds__(256) void transpose_cast_grouped_kernel(const unimm_transpose_desc* __restrict__ tab, int count) {
  __shared__ float tile[32][33];
  int lo = 0, hi = count - 1;                 // wave-uniform binary search over the prefix sums
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].tile0 <= (int)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const unimm_transpose_desc d = tab[lo];
  const int local = blockIdx.x - d.tile0, nbx = (d.C + 31) / 32;
  const int r0 = (local / nbx) * 32, c0 = (local % nbx) * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
  bf16_t* dst = reinterpret_cast<bf16_t*>(d.dst);
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = r0 + ty + 8 * k, c = c0 + tx;
    tile[ty + 8 * k][tx] = (r < d.R && c < d.C) ? d.src[(size_t)r * d.C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = c0 + ty + 8 * k, r = r0 + tx;
    if (c < d.C && r < d.ldd) dst[(size_t)c * d.ldd + r] = f2bf(tile[tx][ty + 8 * k]);
  }
}

// region features: fp32 [rows, F] + fp32 loc [rows, 5] -> bf16 [rows, ld] = [feat | loc | 0...]
// (operand of the single image-embedding GEMM; models/vilbert_dialog.py:1488-1489)
__global__ void pack_image_kernel(const float* __restrict__ feat, const float* __restrict__ loc, bf16_t* __restrict__ out,
                                  int rows, int F, int ld) {
  const int chunks = ld / 8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)rows * chunks;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int row = (int)(i / chunks), c = (int)(i % chunks) * 8;
    float v[8];
    if (c + 8 <= F) {
      const f32x4 a = *reinterpret_cast<const f32x4*>(feat + (size_t)row * F + c);
      const f32x4 b = *reinterpret_cast<const f32x4*>(feat + (size_t)row * F + c + 4);
#pragma unroll
      for (int j = 0; j < 4; ++j) { v[j] = a[j]; v[4 + j] = b[j]; }
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int col = c + j;
        v[j] = col < F ? feat[(size_t)row * F + col] : (col < F + 5 ? loc[(size_t)row * 5 + (col - F)] : 0.f);
      }
    }
    *reinterpret_cast<u32x4*>(out + (size_t)row * ld + c) =
        u32x4{pack2bf(v[0], v[1]), pack2bf(v[2], v[3]), pack2bf(v[4], v[5]), pack2bf(v[6], v[7])};
  }
}

// out = dropout(a * b)  fp32 [n]   (pooled_t * pooled_v, models/vilbert_dialog.py:1065)
template <bool SUM>      // fusion_method 'mul' (default) / 'sum' (models/vilbert_dialog.py:1062-1065)
__global__ void mul_dropout_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ out,
                                   size_t n, DropoutArg drop) {
  drop_resolve(drop);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float v = SUM ? a[i] + b[i] : a[i] * b[i];
  if (drop.thr != 0u) v = drop_apply(drop, 0u, (uint32_t)n, (uint32_t)i, v);
  out[i] = v;
}

// backward of the above given dfused: da = drop(dfused) * b, db = drop(dfused) * a
template <bool SUM>
__global__ void mul_dropout_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                       const float* __restrict__ dout, float* __restrict__ da, float* __restrict__ db,
                                       size_t n, DropoutArg drop) {
  drop_resolve(drop);
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  float d = dout[i];
  if (drop.thr != 0u) d = drop_apply(drop, 0u, (uint32_t)n, (uint32_t)i, d);
  // ReLU of the poolers (:951,:966) is folded in: a, b are post-ReLU, gradient is zero where they are
  const float av = a[i], bv = b[i];
  da[i] = av > 0.f ? (SUM ? d : d * bv) : 0.f;
  db[i] = bv > 0.f ? (SUM ? d : d * av) : 0.f;
}

// du = dt * GELU'(u)  (backward of the erf-GELU that sits between a dense and a LayerNorm in the two
// prediction-head transforms, models/vilbert_dialog.py:983-985, :1002-1004)
__global__ void gelu_bwd_kernel(const bf16_t* __restrict__ dt, const bf16_t* __restrict__ u, bf16_t* __restrict__ du,
                                size_t n8) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n8; i += stride) {
    const u32x4 a = *reinterpret_cast<const u32x4*>(dt + i * 8);
    const u32x4 b = *reinterpret_cast<const u32x4*>(u + i * 8);
    u32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float d0 = __uint_as_float(a[j] << 16), d1 = __uint_as_float(a[j] & 0xffff0000u);
      const float u0 = __uint_as_float(b[j] << 16), u1 = __uint_as_float(b[j] & 0xffff0000u);
      o[j] = pack2bf(d0 * gelu_erf_grad(u0), d1 * gelu_erf_grad(u1));
    }
    *reinterpret_cast<u32x4*>(du + i * 8) = o;
  }
}

// dst[i, :] = src[idx[i], :]  /  dst[idx[i], :] = src[i, :]   (bf16 rows of H elements, H % 8 == 0)
__global__ void gather_rows_kernel(const bf16_t* __restrict__ src, const int32_t* __restrict__ idx, bf16_t* __restrict__ dst,
                                   int n, int H, int scatter, const int32_t* __restrict__ n_dev) {
  if (n_dev != nullptr) n = min(n, n_dev[0]);
  const int cpr = H / 8;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)n * cpr;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < total; i += stride) {
    const int r = (int)(i / cpr), c = (int)(i % cpr) * 8;
    const int g = idx[r];
    if (scatter) *reinterpret_cast<u32x4*>(dst + (size_t)g * H + c) = *reinterpret_cast<const u32x4*>(src + (size_t)r * H + c);
    else *reinterpret_cast<u32x4*>(dst + (size_t)r * H + c) = *reinterpret_cast<const u32x4*>(src + (size_t)g * H + c);
  }
}

inline DropoutArg mk_drop(uint32_t key, uint32_t thr, float scale, const uint32_t* salt = nullptr) {
  DropoutArg d; d.key = key; d.thr = thr; d.scale = scale; d.salt = salt; d.key2 = 0u; return d;
}

}  // namespace

extern "C" int unimm_mask_pack(const void* mask, int dtype, uint32_t* out, int64_t rows, int32_t t, void* stream) {
  if (!mask || !out || rows <= 0 || t <= 0) return UNIMM_E_ARG;
  hipStream_t s = (hipStream_t)stream;
  switch (dtype) {
    case UNIMM_DT_U8: return mask_pack_impl<uint8_t>(mask, out, (size_t)rows, t, s);
    case UNIMM_DT_I32: return mask_pack_impl<int32_t>(mask, out, (size_t)rows, t, s);
    case UNIMM_DT_I64: return mask_pack_impl<int64_t>(mask, out, (size_t)rows, t, s);
    case UNIMM_DT_F32: return mask_pack_impl<float>(mask, out, (size_t)rows, t, s);
    default: return UNIMM_E_ARG;
  }
}

extern "C" int unimm_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y32, void* y16, float* mean,
                                   float* rstd, int32_t M, int32_t H, float eps, uint32_t drop_key, uint32_t drop_thr,
                                   float drop_scale, const uint32_t* drop_salt, void* stream) {
  if (!x || !gamma || !beta || (!y32 && !y16)) return UNIMM_E_ARG;
  if (M <= 0 || H <= 0 || H > MAXC * 512 || (H % 8)) return UNIMM_E_SHAPE;
  int blocks = (M + 3) / 4;
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, gamma, beta, y32,
                     (bf16_t*)y16, mean, rstd, M, H, eps, mk_drop(drop_key, drop_thr, drop_scale, drop_salt));
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int64_t unimm_colpartials_bytes(int32_t H) { return (int64_t)RED_BLOCKS * 4 * H * sizeof(float); }

extern "C" int unimm_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                   void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dbias, float* partials,
                                   int32_t M, int32_t H, uint32_t drop_key, uint32_t drop_thr, float drop_scale,
                                   uint32_t odrop_key, uint32_t odrop_thr, float odrop_scale, const uint32_t* drop_salt,
                                   void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !partials) return UNIMM_E_ARG;
  if (M <= 0 || H <= 0 || H > MAXC * 512 || (H % 8)) return UNIMM_E_SHAPE;
  int blocks = (M + 3) / 4;
  blocks = blocks > RED_BLOCKS ? RED_BLOCKS : blocks;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, s, (const bf16_t*)dy, x, mean, rstd,
                     gamma, (bf16_t*)dx, (bf16_t*)dx_drop, partials, M, H, mk_drop(drop_key, drop_thr, drop_scale, drop_salt),
                     mk_drop(odrop_key, odrop_thr, odrop_scale, drop_salt), (const int32_t*)nullptr);
  UNIMM_CHECK_LAUNCH();
  hipLaunchKernelGGL(colpartials_finish_kernel, dim3((H + 63) / 64, 3), dim3(1024), 0, s, partials, blocks, 3, H, dgamma,
                     dbeta, dbias, (float*)nullptr);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_layernorm_bwd_partials(const void* dy, const float* x, const float* mean, const float* rstd,
                                            const float* gamma, void* dx, void* dx_drop, float* partials, int32_t M, int32_t H,
                                            uint32_t drop_key, uint32_t drop_thr, float drop_scale, uint32_t odrop_key,
                                            uint32_t odrop_thr, float odrop_scale, int32_t* blocks_out, const int32_t* m_dev,
                                            const uint32_t* drop_salt, void* stream) {
  if (!dy || !x || !mean || !rstd || !gamma || !dx || !partials || !blocks_out) return UNIMM_E_ARG;
  if (M <= 0 || H <= 0 || H > MAXC * 512 || (H % 8)) return UNIMM_E_SHAPE;
  int blocks = (M + 3) / 4;
  blocks = blocks > RED_BLOCKS ? RED_BLOCKS : blocks;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, x, mean, rstd,
                     gamma, (bf16_t*)dx, (bf16_t*)dx_drop, partials, M, H, mk_drop(drop_key, drop_thr, drop_scale, drop_salt),
                     mk_drop(odrop_key, odrop_thr, odrop_scale, drop_salt), m_dev);
  UNIMM_CHECK_LAUNCH();
  *blocks_out = blocks;
  return UNIMM_OK;
}

extern "C" int unimm_colpartials_finish_grouped(const unimm_finish_desc* descs, int32_t count, void* stream) {
  if (descs == nullptr || count < 1) return UNIMM_E_ARG;
  for (int base = 0; base < count; base += UNIMM_FINISH_MAX) {
    const int n = count - base < UNIMM_FINISH_MAX ? count - base : UNIMM_FINISH_MAX;
    FinishGroup g;
    int hmax = 0, qmax = 0;
    for (int i = 0; i < n; ++i) {
      const unimm_finish_desc& d = descs[base + i];
      if (d.partials == nullptr || d.blocks < 1 || d.nq < 1 || d.nq > 4 || d.H < 1) return UNIMM_E_ARG;
      g.d[i] = d;
      hmax = d.H > hmax ? d.H : hmax;
      qmax = d.nq > qmax ? d.nq : qmax;
    }
    hipLaunchKernelGGL(colpartials_finish_grouped_kernel, dim3((hmax + 63) / 64, qmax, n), dim3(1024), 0, (hipStream_t)stream, g);
    UNIMM_CHECK_LAUNCH();
  }
  return UNIMM_OK;
}

extern "C" int unimm_embed_fwd(const unimm_embed_args* a, float* y32, void* y, void* stream) {
  if (!a || !a->ids || !a->pos || !a->typ || !a->word || !a->post || !a->type || !a->ext || !a->gamma || !a->beta || !y ||
      !y32)
    return UNIMM_E_ARG;
  if (a->M <= 0 || a->H <= 0 || a->H > MAXC * 512 || (a->H % 8)) return UNIMM_E_SHAPE;
  EmbArgs e;
  e.ids = a->ids; e.pos = a->pos; e.typ = a->typ;
  e.word = a->word; e.post = a->post; e.type = a->type;
  e.ext = a->ext; e.gamma = a->gamma; e.beta = a->beta;
  e.M = a->M; e.H = a->H; e.type_vocab = a->type_vocab; e.eps = a->eps;
  e.drop = mk_drop(a->drop_key, a->drop_thr, a->drop_scale, a->drop_salt);
  e.m_dev = a->m_dev; e.rows = a->rows;
  int blocks = (a->M + 3) / 4;
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(embed_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, e, y32, (bf16_t*)y);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

namespace {
int embed_bwd_any(const unimm_embed_args* a, const void* dy, bool dy_f32, float* dword, float* dpos, float* dtype,
                  float* dext, float* dgamma, float* dbeta, float* partials, void* stream);
}
extern "C" int unimm_embed_bwd(const unimm_embed_args* a, const void* dy, float* dword, float* dpos, float* dtype,
                               float* dext, float* dgamma, float* dbeta, float* partials, void* stream) {
  return embed_bwd_any(a, dy, false, dword, dpos, dtype, dext, dgamma, dbeta, partials, stream);
}
extern "C" int unimm_embed_bwd_f32(const unimm_embed_args* a, const float* dy, float* dword, float* dpos, float* dtype,
                                   float* dext, float* dgamma, float* dbeta, float* partials, void* stream) {
  return embed_bwd_any(a, dy, true, dword, dpos, dtype, dext, dgamma, dbeta, partials, stream);
}
namespace {
int embed_bwd_any(const unimm_embed_args* a, const void* dy, bool dy_f32, float* dword, float* dpos, float* dtype,
                  float* dext, float* dgamma, float* dbeta, float* partials, void* stream) {
  if (!a || !dy || !dword || !dpos || !dtype || !dext || !dgamma || !dbeta || !partials) return UNIMM_E_ARG;
  if (a->M <= 0 || a->H <= 0 || a->H > MAXC * 512 || (a->H % 8) || a->type_vocab != 2) return UNIMM_E_SHAPE;
  EmbArgs e;
  e.ids = a->ids; e.pos = a->pos; e.typ = a->typ;
  e.word = a->word; e.post = a->post; e.type = a->type;
  e.ext = a->ext; e.gamma = a->gamma; e.beta = a->beta;
  e.M = a->M; e.H = a->H; e.type_vocab = a->type_vocab; e.eps = a->eps;
  e.drop = mk_drop(a->drop_key, a->drop_thr, a->drop_scale, a->drop_salt);
  e.m_dev = a->m_dev; e.rows = a->rows;
  int blocks = (a->M + 3) / 4;
  blocks = blocks > RED_BLOCKS ? RED_BLOCKS : blocks;
  hipStream_t s = (hipStream_t)stream;
  if (dy_f32) hipLaunchKernelGGL(embed_bwd_kernel<float>, dim3(blocks), dim3(256), 0, s, e, (const float*)dy, dword, dpos, dext, partials);
  else hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, e, (const bf16_t*)dy, dword, dpos, dext, partials);
  UNIMM_CHECK_LAUNCH();
  hipLaunchKernelGGL(colpartials_finish_kernel, dim3((a->H + 63) / 64, 4), dim3(1024), 0, s, partials, blocks, 4, a->H,
                     dgamma, dbeta, dtype, dtype + a->H);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}
}  // namespace

extern "C" int unimm_plan_lengths(const uint32_t* text_words, int32_t t_q_stride, int32_t t_b_stride, const uint32_t* co_words,
                                  int32_t c_q_stride, int32_t c_b_stride, int32_t R, const int32_t* labels, const int32_t* weights,
                                  const float* nsp_weight, const int32_t* image_label, int32_t B, int32_t T, int32_t* header,
                                  void* stream) {
  if (header == nullptr || (text_words == nullptr && co_words == nullptr && labels == nullptr)) return UNIMM_E_ARG;
  if (B <= 0 || T <= 0 || T > 256 || R < 0 || R > 256) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(plan_lengths_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, text_words, t_q_stride, t_b_stride,
                     co_words, c_q_stride, c_b_stride, R, labels, weights, nsp_weight, image_label, B, T, (T + 31) / 32, header);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_plan_build(const int32_t* header, const int32_t* labels, const int32_t* weights, int32_t B, int32_t T,
                                int32_t* off, int32_t* lens, int64_t* rows, int64_t* inv, int32_t* lm_pos, int32_t* lm_idx,
                                int32_t* lm_label, int32_t* lm_weight, int32_t rows_cap, int32_t lm_cap, int32_t* dims_i,
                                float* dims_f, int32_t* order, void* stream) {
  if ((dims_i == nullptr) != (dims_f == nullptr)) return UNIMM_E_ARG;
  if (header == nullptr || off == nullptr || lens == nullptr) return UNIMM_E_ARG;
  if (lm_pos != nullptr && (lm_idx == nullptr || lm_label == nullptr || lm_weight == nullptr || labels == nullptr)) return UNIMM_E_ARG;
  if (B <= 0 || T <= 0 || T > 256) return UNIMM_E_SHAPE;
  hipLaunchKernelGGL(plan_build_kernel, dim3(B + 1), dim3(256), 0, (hipStream_t)stream, header, labels, weights, B, T, off, lens,
                     rows, inv, lm_pos, lm_idx, lm_label, lm_weight, rows_cap, lm_cap, dims_i, dims_f, order);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_colsum(const void* dy, float* db, int32_t M, int32_t N, int32_t ld, void* stream) {
  if (!dy || !db) return UNIMM_E_ARG;
  if (M <= 0 || N <= 0 || (ld % 8) || ld < ((N + 7) & ~7)) return UNIMM_E_ALIGN;
  int ysplit = (M + 255) / 256;
  ysplit = ysplit > 64 ? 64 : ysplit;
  hipLaunchKernelGGL(colsum_kernel, dim3((N + 511) / 512, ysplit), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dy, db,
                     M, N, ld);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream) {
  if (!src || !dst || n <= 0) return UNIMM_E_ARG;
  if (((uintptr_t)src | (uintptr_t)dst) & 15) return UNIMM_E_ALIGN;
  int64_t blocks = (n / 8 + 255) / 256;
  blocks = blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks);
  hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, src, (bf16_t*)dst,
                     (size_t)n);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_transpose_cast(const float* src, void* dst, int32_t R, int32_t C, int32_t ldd, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || ldd < R) return UNIMM_E_ARG;
  hipLaunchKernelGGL(transpose_cast_kernel, dim3((C + 31) / 32, (ldd + 31) / 32), dim3(256), 0, (hipStream_t)stream, src,
                     (bf16_t*)dst, R, C, ldd);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int32_t lds, int32_t ldd, void* stream) {
  if (!src || !dst || R <= 0 || C <= 0 || lds < C || ldd < R) return UNIMM_E_ARG;
  if ((lds % 8) || (ldd % 8) || (((uintptr_t)src | (uintptr_t)dst) & 15)) return UNIMM_E_ALIGN;
  hipLaunchKernelGGL(transpose_bf16_kernel, dim3((C + 63) / 64, (ldd + 63) / 64), dim3(256), 0, (hipStream_t)stream,
                     (const bf16_t*)src, (bf16_t*)dst, R, C, lds, ldd);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_sum_slabs_bf16(const float* slabs, int32_t count, int64_t stride, void* out, int64_t n, void* stream) {
  if (!slabs || !out || count <= 0 || n <= 0 || stride < n) return UNIMM_E_ARG;
  if ((n % 8) || (stride % 4) || (((uintptr_t)slabs | (uintptr_t)out) & 15)) return UNIMM_E_ALIGN;
  int64_t blocks = (n / 8 + 255) / 256;
  blocks = blocks > 2048 ? 2048 : blocks;
  hipLaunchKernelGGL(sum_slabs_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, slabs, count,
                     (size_t)stride, (bf16_t*)out, (size_t)(n / 8));
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_mask_synth(const int32_t* mode, const int32_t* len, const int32_t* nans, uint32_t* text_words,
                                uint32_t* co_words, int32_t B, int32_t T, void* stream) {
  if (!mode || !len || !nans || !text_words || !co_words || B <= 0 || T <= 0) return UNIMM_E_ARG;
  const size_t total = (size_t)B * T * ((T + 31) / 32);
  hipLaunchKernelGGL(mask_synth_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, mode, len,
                     nans, text_words, co_words, B, T);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_transpose_cast_grouped(const unimm_transpose_desc* table, int32_t count, int32_t total_tiles, void* stream) {
  if (!table || count <= 0 || total_tiles <= 0) return UNIMM_E_ARG;
  hipLaunchKernelGGL(transpose_cast_grouped_kernel, dim3(total_tiles), dim3(256), 0, (hipStream_t)stream, table, count);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_pack_image(const float* feat, const float* loc, void* out, int32_t rows, int32_t F, int32_t ld,
                                void* stream) {
  if (!feat || !loc || !out || rows <= 0 || F <= 0 || (F % 8) || (ld % 8) || ld < F + 5) return UNIMM_E_ARG;
  if (((uintptr_t)feat | (uintptr_t)out) & 15) return UNIMM_E_ALIGN;
  const size_t total = (size_t)rows * (ld / 8);
  size_t blocks = (total + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  hipLaunchKernelGGL(pack_image_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, feat, loc, (bf16_t*)out,
                     rows, F, ld);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_mul_dropout(const float* a, const float* b, float* out, int64_t n, uint32_t drop_key, uint32_t drop_thr,
                                 float drop_scale, const uint32_t* drop_salt, void* stream) {
  if (!a || !b || !out || n <= 0) return UNIMM_E_ARG;
  hipLaunchKernelGGL(mul_dropout_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, b, out, (size_t)n, mk_drop(drop_key, drop_thr, drop_scale, drop_salt));
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_mul_dropout_bwd(const float* a, const float* b, const float* dout, float* da, float* db, int64_t n,
                                     uint32_t drop_key, uint32_t drop_thr, float drop_scale, const uint32_t* drop_salt,
                                     void* stream) {
  if (!a || !b || !dout || !da || !db || n <= 0) return UNIMM_E_ARG;
  hipLaunchKernelGGL(mul_dropout_bwd_kernel<false>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, b, dout, da, db, (size_t)n,
                     mk_drop(drop_key, drop_thr, drop_scale, drop_salt));
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_sum_dropout(const float* a, const float* b, float* out, int64_t n, uint32_t drop_key, uint32_t drop_thr,
                                 float drop_scale, const uint32_t* drop_salt, void* stream) {
  if (!a || !b || !out || n <= 0) return UNIMM_E_ARG;
  hipLaunchKernelGGL(mul_dropout_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, b, out, (size_t)n, mk_drop(drop_key, drop_thr, drop_scale, drop_salt));
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_sum_dropout_bwd(const float* a, const float* b, const float* dout, float* da, float* db, int64_t n,
                                     uint32_t drop_key, uint32_t drop_thr, float drop_scale, const uint32_t* drop_salt,
                                     void* stream) {
  if (!a || !b || !dout || !da || !db || n <= 0) return UNIMM_E_ARG;
  hipLaunchKernelGGL(mul_dropout_bwd_kernel<true>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, b, dout, da, db, (size_t)n,
                     mk_drop(drop_key, drop_thr, drop_scale, drop_salt));
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_gelu_bwd(const void* dt, const void* u, void* du, int64_t n, void* stream) {
  if (!dt || !u || !du || n <= 0 || (n % 8)) return UNIMM_E_ARG;
  int64_t blocks = (n / 8 + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dt,
                     (const bf16_t*)u, (bf16_t*)du, (size_t)(n / 8));
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}

extern "C" int unimm_gather_rows(const void* src, const int32_t* idx, void* dst, int32_t n, int32_t H, int32_t scatter,
                                 const int32_t* n_dev, void* stream) {
  if (!src || !idx || !dst || n <= 0 || H <= 0 || (H % 8)) return UNIMM_E_ARG;
  size_t blocks = ((size_t)n * (H / 8) + 255) / 256;
  blocks = blocks > 8192 ? 8192 : blocks;
  hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)src, idx,
                     (bf16_t*)dst, n, H, scatter, n_dev);
  UNIMM_CHECK_LAUNCH();
  return UNIMM_OK;
}
