/* unimm_hip.h -- C ABI of libunimm_hip.so: hand-written gfx950 (MI355X) kernels for the UniMM-UL
 * forward/backward hot path.
 *
 * The reference (ZihaoW123/UniMM) is pure Python/PyTorch and has no FFI of its own (SURVEY.md 2:
 * no .cu/.cpp, no custom op).  The "interface each entry point replaces" is therefore a torch-op
 * chain inside models/vilbert_dialog.py; every declaration below cites it (file:line relative to
 * the reference root).  INTEGRATION.md shows the ctypes stub a maintainer would add.
 *
 * Conventions: extern "C"; every pointer is a DEVICE pointer on the current HIP device; `stream`
 * is a hipStream_t passed as void*; bf16 tensors are uint16 bit patterns; row strides ("ld") are in
 * elements.  Calls only enqueue work (no allocation, no synchronisation) and return UNIMM_OK or a
 * negative UNIMM_E_* code; they never throw.
 */
#ifndef UNIMM_HIP_H
#define UNIMM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNIMM_OK 0
#define UNIMM_E_ARG (-1)   /* null pointer / unknown enum value */
#define UNIMM_E_SHAPE (-2) /* unsupported size */
#define UNIMM_E_ALIGN (-3) /* pointer or stride alignment */
#define UNIMM_E_HIP (-4)   /* launch failed */

int unimm_version(void);          /* ABI version, bumped on any signature change */
const char* unimm_arch(void);     /* "gfx950" */

/* ---------------------------------------------------------------------------------------------
 * GEMM, "NT": OUT[M,N] = epilogue( X[M,K] . W[N,K]^T ), bf16 operands, fp32 accumulate.
 * Replaces every nn.Linear forward on the path (models/vilbert_dialog.py:386-388, 423, 453, 466,
 * 515-517, 552, 582, 595, 659-661, 670-672, 745-748, 950, 965, 983, 1002, 1025, 1070, 1087,
 * 1488-1489) with the elementwise op that follows it fused in, and -- called with the transposed
 * bf16 weight copy -- the input-gradient half of their autograd backward.
 * K % 64 == 0; ldx, ldw % 8 == 0; ldo, ldaux % 4 == 0; x, w, out 16-byte aligned.
 * ------------------------------------------------------------------------------------------- */
enum {
  UNIMM_EPI_BIAS = 0,            /* out = acc + bias                                            */
  UNIMM_EPI_BIAS_GELU = 1,       /* u = acc + bias; out = erf-GELU(u) (:115-121); out2 = u      */
  UNIMM_EPI_BIAS_DROP_RESID = 2, /* out(fp32) = dropout(acc + bias) + aux(fp32 residual stream) (:423-425, :466-468, ...) */
  UNIMM_EPI_BIAS_RELU = 3,       /* poolers (:950-951, :965-966)                                */
  UNIMM_EPI_DGELU = 4,           /* out = acc * GELU'(aux)           (backward of :453-454)     */
  UNIMM_EPI_ADD = 5,             /* out = acc + aux                  (residual gradient join)   */
  UNIMM_EPI_MUL = 6,             /* out = acc * aux                  (backward of GELU with aux = GELU'(u)) */
  UNIMM_EPI_BIAS_GELU_DG = 7     /* u = acc + bias; out = erf-GELU(u); out2 = GELU'(u) (training FFN) */
};

typedef struct {
  const void* x;     /* [M, K] bf16 */
  const void* w;     /* [N, K] bf16 */
  const float* bias; /* [N] fp32 or NULL */
  const void* aux;   /* [M, N] epilogue operand or NULL: fp32 residual (DROP_RESID), bf16 otherwise */
  void* out;         /* [M, N] bf16, or fp32 when out_f32 != 0 */
  void* out2;        /* [M, N] bf16 second output of UNIMM_EPI_BIAS_GELU (row stride ldo) or NULL */
  int32_t M, N, K;
  int32_t ldx, ldw, ldaux, ldo;
  int32_t epilogue;
  int32_t out_f32;
  uint32_t drop_key; /* dropout (UNIMM_EPI_BIAS_DROP_RESID): element (m, n) is kept iff the 16-bit field (n & 1) of
                      * drop_word(key, m * ceil(N / 2) + n / 2) is >= thr -- one counter-based hash per two neighbouring
                      * columns (csrc/common.h: drop_word; unimm_amd/dropout.py is the bit-exact host mirror)  */
  uint32_t drop_thr; /* p * 2^16; 0 disables                                                    */
  float drop_scale;  /* 1 / (1 - p)                                                             */
  /* UNIMM_EPI_BIAS_DROP_RESID only, all four or none: the residual operand is LayerNorm(aux) evaluated on
   * the fly, (aux[m,n] - aux_mean[m]) * aux_rstd[m] * aux_gamma[n] + aux_beta[n] -- the fp32 output of the
   * previous LayerNorm (models/vilbert_dialog.py:425, :468, ...) is then never written or read */
  const float* aux_mean;
  const float* aux_rstd;
  const float* aux_gamma;
  const float* aux_beta;
  /* Per-call tuning (tests and A/B measurements; 0 = everything automatic): tile = gn * 1000 + p * 100 + cfg.
   *   cfg: 0 automatic (a cost model over {256x256, 192x256, 128x128}; 64x128 for grids that do not fill the chip),
   *        1 = 128x128 / 4 waves / 2-slot ring of BK 64 / 2 workgroups per CU,   3 = 256x256 / 8 waves / lock-step ring,
   *        6 = 192x256 / 8 waves,   7 = 64x128 / 4 waves of 32x64 / 3 workgroups per CU,   8 = 256x256 / ping-pong loop,
   *        9 = 64x128 with a 3-slot ring (two K steps in flight, 2 workgroups per CU),   10 = 128x128 with a 3-slot ring (1 per CU),
   *        12 / 14 / 15 = 192x256 / 128x128 / 64x128 with the X operand alone on a three-slot ring (W stays on two: X(t+2) is in
   *        flight while step t computes -- in a training step X was just written by the previous kernel and streams from HBM;
   *        12 is what the automatic choice uses for its 192x256 class);  anything else is rejected with UNIMM_E_ARG
   *   p:   0 automatic, 1 persistent workgroups (one per CU slot walks several tiles), 2 one workgroup per tile
   *   gn:  n-tiles per column group of the tile order (0 = default 4) */
  int32_t tile;
  /* or NULL: a device word XORed into drop_key at kernel entry.  The argument then carries the per-(seed, site) part of the
   * key and memory the per-step part, so that a captured / replayed launch (unimm_amd/graphs.py) draws a fresh mask every
   * step; unimm_amd/dropout.py: make_key(seed, step, site) = site_key(seed, site) ^ step_salt(seed, step).  Every entry
   * point that takes a dropout triple takes such a word. */
  const uint32_t* drop_salt;
  /* Split-K for grids that leave the chip under-filled (the ~4-8k rows of a 30-60-sequence share of a split batch against
   * K = 2304 / 3072: a few hundred tiles, each a 36-48 step dependent chain).  splitk: 0 / 1 = off; 2 .. 8 = at most that many
   * workgroups per output tile, each reducing a slice of K; -1 = the library's choice (as many as stay resident at once, <= 4,
   * >= 8 K-steps each).  Partial tiles meet in the caller's workspace and the last arriver of a tile runs the epilogue, so the
   * fused epilogues work unchanged.  splitk_ws: 256-byte aligned device memory, ZERO-FILLED ONCE by the caller and then private
   * to launches of ONE stream (the kernel leaves its counters zero); 16 KiB + tiles * splits * tile bytes (a 64 x 128 tile is
   * 32 KiB); too small for a launch = that launch runs unsplit.  Ring-loop tiles only (64x128, 128x128). */
  void* splitk_ws;
  int64_t splitk_ws_bytes;
  int32_t splitk;
} unimm_gemm_nt_args;

int unimm_gemm_nt(const unimm_gemm_nt_args* args, void* stream);

/* GEMM, "TN": DW[N,K] += DY[M,N]^T . X[M,K] (fp32 atomics; caller zeroes DW once per step) and,
 * when dbias != NULL, dbias[N] += column sums of DY (the bias gradient, one extra MFMA per tile).
 * Weight-gradient half of every nn.Linear backward on the path.  Rows of DY / X must be readable
 * up to round_up(N, 8) / round_up(K, 8) columns; lddy, ldx % 8 == 0. */
typedef struct {
  const void* dy; /* [M, N] bf16 */
  const void* x;  /* [M, K] bf16 */
  float* dw;      /* [N, K] fp32, row stride lddw */
  float* dbias;   /* [N] fp32 or NULL */
  int32_t M, N, K;
  int32_t lddy, ldx, lddw;
  const int32_t* m_dev; /* or NULL: device word holding the number of reduction rows actually present; M is then the
                         * CAPACITY the launch is sized for (replayed launch sequences: see unimm_plan_build) */
  int32_t overwrite;    /* != 0: DW = DY^T X instead of +=.  The caller asserts that dw is all zero and that no other problem,
                         * launch or stream adds to it concurrently; a tile reduced by one workgroup is then written with plain
                         * stores (no memory-side atomics).  Ignored (+= as usual) where the reduction is split.  dbias always +=.
                         * ABI 16. */
} unimm_gemm_tn_args;

int unimm_gemm_tn(const unimm_gemm_tn_args* args, void* stream);
/* `count` independent TN problems (host array) in as few grids as possible: the weight gradients of one
 * encoder block (models/vilbert_dialog.py:386-388, 423-425, 453-454, 466-468 and twins) in one launch.
 * Same result as `count` calls of unimm_gemm_tn; the problems may alias each other's dw / dbias when they are
 * meant to accumulate (every path, with or without a workspace, ends in fp32 atomics). */
int unimm_gemm_tn_grouped(const unimm_gemm_tn_args* args, int32_t count, void* stream);
/* The same with (a) the "launches share the chip with another stream's kernels" hint of the split heuristic
 * (shared_chip != 0: fewer, longer workgroups and fewer partial tiles), and (b) a caller-owned WORKSPACE: when the
 * reduction is split, every split stores its fp32 partial tile to a slab of the workspace (plain coalesced stores) and
 * the split that arrives last at the tile's counter sums the slabs and adds the tile into dw once, instead of every
 * split adding 256 KiB with memory-side atomics (8x the algorithmic write traffic, a ~43 us drain per launch).
 * ws: 256-byte aligned device memory, ZERO-FILLED ONCE by the caller before its first use and then private to launches
 * of ONE stream (the kernels leave the counters zero); ws_bytes >= 16 KiB + tiles * splits * tile bytes
 * (512 MiB covers every launch of the full config at 240 sequences); too small or NULL = the atomic path.
 * The result equals the atomic path's up to the order of the fp32 sums (the last arriver adds the other splits'
 * partials to its own in index order). */
int unimm_gemm_tn_grouped_ws(const unimm_gemm_tn_args* args, int32_t count, int32_t shared_chip, void* ws, int64_t ws_bytes,
                             void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused attention core: out = dropout(softmax(Q K^T * scale + additive(mask))) V per (sequence,
 * head); additive(mask) = (1 - bit) * -10000 exactly as models/vilbert_dialog.py:1415-1431.
 * Replaces :390-410 (text, Tq=Tk=256, D=64, dense mask), :519-539 (image, 37x37, D=128, key mask)
 * and both directions of :681-721 (256x37 with the image key mask; 37x256 with the co-attention
 * mask).  Q/K/V are strided views into the fused projection output: row (b*T + t), columns
 * head*D .. head*D+D-1 from the given base pointer.  Tq, Tk <= 256; D in {64, 128}.
 * mask: bit-packed words [B][rows][ceil(Tk/32)] (unimm_mask_pack); mask_q_stride = 0 broadcasts
 * one row over all queries (key-padding mask).  lse (fp32 [B,H,Tq], log-sum-exp of the masked,
 * scaled scores) is what the backward kernels need; may be NULL for inference.
 * Variable-length ("unpadded") mode: when q_off/q_len (int32 [B], device) are given, sequence b owns
 * rows [q_off[b], q_off[b]+q_len[b]) of the packed q / out matrices (same for k_off/k_len and k, v);
 * Tq / Tk remain the PADDED lengths that index mask, lse and the dropout counters.  Padding rows
 * (fully masked queries that no valid row attends, models/vilbert_dialog.py:1418) are then never
 * computed at all.
 * Alignment: q, k, v, out (and dq, dk, dv, dout of the backward) 16-byte aligned, every row stride a multiple of 8
 * elements: operand fragments are 16-byte loads and result rows leave as 16-byte stores; UNIMM_E_ALIGN otherwise.
 * ------------------------------------------------------------------------------------------- */
typedef struct {
  const void* q; const void* k; const void* v; /* bf16 */
  void* out;                                   /* bf16 [rows, ldo] */
  float* lse;
  const uint32_t* mask;
  const int32_t* q_off; const int32_t* q_len; const int32_t* k_off; const int32_t* k_len; /* or all NULL */
  int32_t B, H, Tq, Tk, D;
  int32_t ldq, ldk, ldv, ldo;
  int32_t mask_q_stride, mask_b_stride; /* in 32-bit words */
  float scale;
  uint32_t drop_key, drop_thr; float drop_scale;
  const uint32_t* drop_salt; /* or NULL (see unimm_gemm_nt_args.drop_salt) */ /* element index = ((b*H+h)*Tq+q)*Tk+k */
  /* unimm_attn_fwd only (NULL elsewhere), with k_off / k_len given and dropout off: a SHARED key/value segment -- rows
   * [ks_off[b], ks_off[b] + ks_len[b]) of the same k / v matrices -- spliced into sequence b's keys after its first ks_ins
   * private rows.  Key position j of sequence b is private row j (j < ks_ins), shared row j - ks_ins (j < ks_ins + ks_len[b]),
   * else private row j - ks_len[b]; the sequence has k_len[b] + ks_len[b] <= 256 keys and the mask words index key positions.
   * Generative scoring (val_lm.py:52-121): the 100 candidate answers of a dialog round attend the round's context rows,
   * which are computed once (utils/data_utils.py:199-210: context rows never see the answer).  ABI 16. */
  const int32_t* ks_off; const int32_t* ks_len;
  int32_t ks_ins;
  /* or NULL: int32 [B], a permutation of 0..B-1 -- the order in which the launch's workgroups take the sequences (every head of
   * order[0] first).  Results do not depend on it; with variable lengths, longest first (unimm_plan_build writes that) keeps the
   * launch's tail short: +1 % on the 240-sequence step.  ABI 17.  (The fp32-class entry points honour it in their matrix-instruction kernels.) */
  const int32_t* order;
} unimm_attn_args;

int unimm_attn_fwd(const unimm_attn_args* args, void* stream);

/* The attention probabilities themselves, fp32 [B, H, Tq, Tk] (after dropout, as models/vilbert_dialog.py:401-405 /
 * :690-717 return them and BertEncoder collects them under output_all_attention_masks, :855-929).  Fixed layout only
 * (q_off .. k_len NULL); v / out / lse of the argument struct are ignored.  A diagnostic output, not part of the hot path:
 * unimm_attn_fwd never materialises them. */
int unimm_attn_probs(const unimm_attn_args* args, float* probs, void* stream);

/* Backward of unimm_attn_fwd (autograd of models/vilbert_dialog.py:390-410 / 519-539 / 681-721):
 * ONE launch for the text self-attention shape (D = 64, Tq and Tk above 64: dQ comes out of the dK / dV walk, `delta` is
 * then formed inside the kernel and the argument is left untouched); two launches on `stream` for the other shapes --
 * dQ (+ delta = rowsum(dO o O), fp32 [B,H,Tq] scratch) then dK/dV.
 * P is recomputed from Q, K and lse; the dropout mask is re-generated from (key, thr).
 * dq/dk/dv are bf16 strided views like q/k/v (typically into one [rows, 3*H*D] buffer). */
typedef struct {
  const void* q; const void* k; const void* v; const void* out; const void* dout; /* bf16 */
  const float* lse; float* delta;
  void* dq; void* dk; void* dv; /* bf16 */
  const uint32_t* mask;
  const int32_t* q_off; const int32_t* q_len; const int32_t* k_off; const int32_t* k_len; /* or all NULL */
  int32_t B, H, Tq, Tk, D;
  int32_t ldq, ldk, ldv, ldo, lddo, lddq, lddk, lddv;
  int32_t mask_q_stride, mask_b_stride;
  float scale;
  uint32_t drop_key, drop_thr; float drop_scale;
  const uint32_t* drop_salt;
  const int32_t* order;      /* as unimm_attn_args.order (ABI 17) */
} unimm_attn_bwd_args;

int unimm_attn_bwd(const unimm_attn_bwd_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row kernels (HBM-bound).
 * ------------------------------------------------------------------------------------------- */
enum { UNIMM_DT_U8 = 0, UNIMM_DT_I32 = 1, UNIMM_DT_I64 = 2, UNIMM_DT_F32 = 3 };

/* Bit-pack a 0/1 mask [rows, t] (bool/uint8, int32, int64 or fp32; nonzero = attend) into
 * ceil(t/32) words per row.  Replaces the fp32 (1-m)*-10000 mask tensors of
 * models/vilbert_dialog.py:1415-1431 (the -10000 is applied inside the attention kernels). */
int unimm_mask_pack(const void* mask, int dtype, uint32_t* out, int64_t rows, int32_t t, void* stream);
/* The same words computed on the HOST (ABI 18; no device work, no stream): `mask` and `out` are host pointers.  The reference's
 * callers pass CPU tensors to forward() (train.py:113-129; utils/data_parallel.py:123-124 scatters them) -- int64 [B, 256, 256]
 * masks, 512 KiB per sequence; packed on the host side of the copy, 8 KiB per sequence cross PCIe.  threads: 0 = up to 16. */
int unimm_host_mask_pack(const void* mask, int dtype, uint32_t* out, int64_t rows, int32_t t, int32_t threads);
/* memcpy between two HOST buffers on `threads` threads (0 = up to 8): the caller's pageable tensors -> the pinned staging ring the
 * asynchronous host->device copies read (one thread moves the 130 MB of region features / targets of a 240-sequence batch in
 * ~25 ms: most of a step; torch's copy_ depends on the size of torch's intra-op pool).  ABI 18. */
int unimm_host_memcpy(void* dst, const void* src, int64_t bytes, int32_t threads);
/* The same words without the dense mask: the text mask [B, T, ceil(T/32)] and the
 * co-attention key mask [B, ceil(T/32)] (one row per sequence, mask_q_stride = 0) of B sequences from three
 * int32 device arrays: mode (0 = discriminative, utils/data_utils.py:391-396; 1 = generative, :199-210),
 * len = L (tokens up to and including the answer's [SEP]; L <= T) and nans = n (answer length + 1, the size
 * of the [MASK]-copy block; ignored when mode = 0).  Removes the 512 KiB per sequence of int64 mask the
 * reference ships to the device. */
int unimm_mask_synth(const int32_t* mode, const int32_t* len, const int32_t* nans, uint32_t* text_words, uint32_t* co_words,
                     int32_t B, int32_t T, void* stream);

/* Plan of the unpadded schedule, on the device (replaces ~25 eager kernels over the dense masks and two of the three
 * device->host round trips of a step).  unimm_plan_lengths: header[b] = valid prefix length of sequence b (>= 1; a token
 * is valid when it attends something, is attended by a token or a region, or carries a label / weight), header[B+b] =
 * rows the MLM head decodes (weight != 0 when weights are given, else label != -1), header[2B..2B+1] = bit patterns of
 * the two NSP class weights when nsp_weight != NULL, header[2B+2+b] = regions of sequence b with image_label == 1
 * (int32 [B, R] or NULL: the divisor of the masked-region loss); header has 3B+2 entries.  Masks are the packed words of
 * unimm_mask_pack / unimm_mask_synth with their (query, batch) word strides (query stride 0 = one key-padding row per
 * sequence); text_words or co_words may be NULL, labels / weights int32 [B, T] or NULL.
 * unimm_plan_build (after the host has read the header and allocated): off[B] / lens[B], rows[sum len] = padded row of
 * every packed row, inv[B*T] = packed row or -1, and for the decoded rows, in row order, lm_pos (padded row), lm_idx
 * (packed row), lm_label, lm_weight (1 when weights == NULL); rows / inv / the lm_* group may be NULL. */
int unimm_plan_lengths(const uint32_t* text_words, int32_t t_q_stride, int32_t t_b_stride, const uint32_t* co_words,
                       int32_t c_q_stride, int32_t c_b_stride, int32_t R, const int32_t* labels, const int32_t* weights,
                       const float* nsp_weight, const int32_t* image_label, int32_t B, int32_t T, int32_t* header,
                       void* stream);
/* Replayed launch sequences (unimm_amd/graphs.py: the step as two hipGraphs): kernel ARGUMENTS are frozen at capture, so a
 * launch is sized for a CAPACITY (the step's row counts rounded up to a bucket) and every kernel whose result would change
 * with the surplus rows reads the REAL count from device memory (the `*_dev` arguments below and in the structs above;
 * NULL = the argument is exact).  unimm_plan_build writes those words: dims_i = {valid text rows, decoded rows, regions in
 * the masked-region loss}, dims_f = {1 / decoded rows, 1 / regions (inf when none, as the reference's division)}; it also
 * fills rows[real .. rows_cap) and the lm_* lists [real .. lm_cap) with safe values (index 0, label -1, weight 0) so that
 * row-independent kernels (GEMM, LayerNorm forward, gathers) may run over the whole capacity.  dims_i (>= 4 words) / dims_f
 * (>= 2) both or neither; capacities 0 = lists are exact.  A batch whose real counts EXCEED a capacity (a replayed step sized
 * from a stale header) never writes past the lists: indices and the counts in dims_i are clamped to the capacities,
 * dims_i[3] = 1 and both dims_f words are NaN (the step's losses come out NaN instead of memory being corrupted). */
int unimm_plan_build(const int32_t* header, const int32_t* labels, const int32_t* weights, int32_t B, int32_t T,
                     int32_t* off, int32_t* lens, int64_t* rows, int64_t* inv, int32_t* lm_pos, int32_t* lm_idx,
                     int32_t* lm_label, int32_t* lm_weight, int32_t rows_cap, int32_t lm_cap, int32_t* dims_i, float* dims_f,
                     int32_t* order, void* stream);
/* order (ABI 17; or NULL): int32 [B] = the sequences by valid length, longest first (ties in batch order): what
 * unimm_attn_args.order takes. */

/* y = LayerNorm(x) (eps inside sqrt, torch.nn.LayerNorm; models/vilbert_dialog.py:279) with optional
 * dropout on y.  The residual stream is fp32 (as under the reference's autocast, where layer_norm and
 * the residual add run in fp32): x fp32 [M, H]; y32 (fp32, next residual) and / or y16 (bf16, next GEMM
 * operand) are written; mean/rstd fp32 [M] saved for backward (may be NULL). */
int unimm_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y32, void* y16, float* mean,
                        float* rstd, int32_t M, int32_t H, float eps, uint32_t drop_key, uint32_t drop_thr,
                        float drop_scale, const uint32_t* drop_salt, void* stream);

/* bytes of the `partials` scratch the two backward row kernels need for hidden size H */
int64_t unimm_colpartials_bytes(int32_t H);

/* LayerNorm backward (dy bf16, x = the fp32 pre-LayerNorm sum).  dx (bf16) is the gradient w.r.t. the pre-LayerNorm sum (= the residual
 * branch gradient); dx_drop (optional) = dropout-masked dx for the dense branch (drop_*: the mask the
 * forward GEMM epilogue applied); odrop_*: dropout applied to y in the forward (embeddings), 0 = none.
 * dgamma/dbeta/dbias (fp32 [H], any may be NULL) are ACCUMULATED (+=); dbias = colsum(dx_drop). */
int unimm_layernorm_bwd(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                        void* dx, void* dx_drop, float* dgamma, float* dbeta, float* dbias, float* partials,
                        int32_t M, int32_t H, uint32_t drop_key, uint32_t drop_thr, float drop_scale,
                        uint32_t odrop_key, uint32_t odrop_thr, float odrop_scale, const uint32_t* drop_salt, void* stream);
/* The same row kernel without the reduction of its column partials: `partials` (unimm_colpartials_bytes(H), private to
 * this call until it is reduced) holds [blocks][3][H] = per-block sums for dgamma, dbeta, dbias; *blocks_out (host)
 * receives the block count.  unimm_colpartials_finish_grouped then adds the column sums of up to many pending calls
 * into their destinations in one launch (dst[q] == NULL skips quantity q): the engine reduces a block's LayerNorm
 * partials once at the end of the block instead of between two dependent kernels each time. */
int unimm_layernorm_bwd_partials(const void* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                 void* dx, void* dx_drop, float* partials, int32_t M, int32_t H, uint32_t drop_key,
                                 uint32_t drop_thr, float drop_scale, uint32_t odrop_key, uint32_t odrop_thr,
                                 float odrop_scale, int32_t* blocks_out, const int32_t* m_dev, const uint32_t* drop_salt,
                                 void* stream);
#define UNIMM_FINISH_MAX 8
typedef struct {
  const float* partials;
  float* dst[4];
  int32_t blocks, nq, H, pad_;
} unimm_finish_desc;
int unimm_colpartials_finish_grouped(const unimm_finish_desc* descs, int32_t count, void* stream);

/* Text embeddings: y = dropout(LN(word[ids] + pos[position] + type)) with token-type ids >= type_vocab
 * routed to the 10-row extension table (BertEmbeddingsDialog.forward, models/vilbert_dialog.py:326-356).
 * Tables are the fp32 master weights; ids int32 [M]; y32 / y16 as for unimm_layernorm_fwd. */
typedef struct {
  const int32_t* ids; const int32_t* pos; const int32_t* typ;
  const float* word; const float* post; const float* type; const float* ext; /* fp32 tables, row length H */
  const float* gamma; const float* beta;
  int32_t M, H, type_vocab;
  float eps;
  uint32_t drop_key, drop_thr; float drop_scale;
  const int32_t* m_dev; /* or NULL: device word with the rows actually present (M = capacity) */
  const int64_t* rows;  /* or NULL: row r takes ids / pos / typ at index rows[r] (the unpadded schedule's row map) */
  const uint32_t* drop_salt; /* or NULL (see unimm_gemm_nt_args.drop_salt) */
} unimm_embed_args;

int unimm_embed_fwd(const unimm_embed_args* args, float* y32, void* y16, void* stream);
/* backward: re-gathers the rows, LayerNorm backward, scatter-add (fp32 atomics) into the word /
 * position / extension-type gradient tables; dtype [2, H], dgamma, dbeta accumulated via partials. */
int unimm_embed_bwd(const unimm_embed_args* args, const void* dy, float* dword, float* dpos, float* dtype,
                    float* dext, float* dgamma, float* dbeta, float* partials, void* stream);

/* db[N] += column sums of dy (bf16 [M, N], row stride ld): bias gradients. */
int unimm_colsum(const void* dy, float* db, int32_t M, int32_t N, int32_t ld, void* stream);

/* fp32 -> bf16 weight copies: flat cast, and dst[c][r] = src[r][c] (dst row stride ldd >= R, columns
 * R..ldd-1 zero-filled) for the transposed copies the input-gradient GEMMs read. */
int unimm_cast_f32_bf16(const float* src, void* dst, int64_t n, void* stream);
int unimm_transpose_cast(const float* src, void* dst, int32_t R, int32_t C, int32_t ldd, void* stream);
/* bf16 -> bf16 transpose: dst[c][r] = src[r][c]; src [R, C] row stride lds, dst [C, ldd] with columns R..ldd-1
 * zero-filled; lds, ldd multiples of 8, both pointers 16-byte aligned.  Puts the decoder-logit gradient of
 * models/vilbert_dialog.py:1023-1026 reduction-major so that its input gradient (a few hundred rows x 768 over the
 * 30,522-long vocabulary axis) can run as a split reduction on unimm_gemm_tn_grouped. */
int unimm_transpose_bf16(const void* src, void* dst, int32_t R, int32_t C, int32_t lds, int32_t ldd, void* stream);
/* out[i] (bf16) = slabs[0][i] + slabs[1][i] + ... + slabs[count-1][i] (fp32, in that order), i < n; slab s starts at
 * slabs + s * stride.  n a multiple of 8, stride of 4, 16-byte aligned pointers.  The fixed-order reduction of the
 * per-chunk partial sums of the split decoder input gradient (see unimm_transpose_bf16). */
int unimm_sum_slabs_bf16(const float* slabs, int32_t count, int64_t stride, void* out, int64_t n, void* stream);
/* The same for `count` matrices in one launch.  `table` is a DEVICE array; entry i owns blocks
 * [tile0_i, tile0_{i+1}) with ceil(C/32) * ceil(ldd/32) blocks each (tile0 ascending, tile0_0 = 0);
 * total_tiles = their sum.  Used after the optimizer step to rebuild every transposed weight copy. */
typedef struct {
  const float* src;  /* fp32 [R, C] */
  void* dst;         /* bf16 [C, ldd] */
  int32_t R, C, ldd, tile0;
} unimm_transpose_desc;
int unimm_transpose_cast_grouped(const unimm_transpose_desc* table, int32_t count, int32_t total_tiles, void* stream);

/* Region features fp32 [rows, F] + box geometry fp32 [rows, 5] -> bf16 [rows, ld] = [feat | loc | 0]:
 * the operand of the single image-embedding GEMM (models/vilbert_dialog.py:1488-1489). */
int unimm_pack_image(const float* feat, const float* loc, void* out, int32_t rows, int32_t F, int32_t ld, void* stream);

/* out = dropout(a * b) (pooled_t * pooled_v, models/vilbert_dialog.py:1065), fp32 flat [n], and its backward, which
 * also folds in the pooler ReLU gradient (:951, :966): da = [a>0] drop(dout) b, db = [b>0] drop(dout) a. */
int unimm_mul_dropout(const float* a, const float* b, float* out, int64_t n, uint32_t drop_key, uint32_t drop_thr,
                      float drop_scale, const uint32_t* drop_salt, void* stream);
int unimm_mul_dropout_bwd(const float* a, const float* b, const float* dout, float* da, float* db, int64_t n,
                          uint32_t drop_key, uint32_t drop_thr, float drop_scale, const uint32_t* drop_salt, void* stream);

/* The same for fusion_method = 'sum' (models/vilbert_dialog.py:1062-1063): out = dropout(a + b); da = [a>0] drop(dout),
 * db = [b>0] drop(dout). */
int unimm_sum_dropout(const float* a, const float* b, float* out, int64_t n, uint32_t drop_key, uint32_t drop_thr,
                      float drop_scale, const uint32_t* drop_salt, void* stream);
int unimm_sum_dropout_bwd(const float* a, const float* b, const float* dout, float* da, float* db, int64_t n,
                          uint32_t drop_key, uint32_t drop_thr, float drop_scale, const uint32_t* drop_salt, void* stream);

/* fp32 linear algebra of the heads on top of the network (the two poolers :946-967, the NSP head :1070) and their
 * backward, on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32, straight from the fp32 master weights:
 *   OUT[m, n] (+)= act( sum_k A(m, k) * B(k, n) + bias[n] ),  A(m, k) = a[m * sa_m + k * sa_k],  B(k, n) = b[k * sb_k + n * sb_n]
 * (strides in elements), act = ReLU when relu != 0; accumulate != 0 adds into OUT with fp32 atomics (gradients
 * accumulate); rowsum (or NULL): rowsum[m] += sum_k A(m, k), the bias gradient when A = dY^T.  One kernel gives
 *   y = x W^T + b   (a = x: sa_m = ldx, sa_k = 1;  b = W [N, K]: sb_k = 1, sb_n = ldw),
 *   dx = dy W        (a = dy: sa_m = lddy, sa_k = 1;  b = W: sb_k = ldw, sb_n = 1),
 *   dW += dy^T x     (a = dy: sa_m = 1, sa_k = lddy;  b = x: sb_k = ldx, sb_n = 1;  M = out features, K = batch rows). */
typedef struct {
  const float* a; const float* b; const float* bias; float* out; float* rowsum;
  int32_t M, N, K;
  int64_t sa_m, sa_k, sb_k, sb_n, ldo;
  int32_t relu, accumulate;
} unimm_linear_f32_args;
int unimm_linear_f32(const unimm_linear_f32_args* args, void* stream);
/* dst[idx[r], :] += src[r, :]: bf16 rows of H elements, fp32 addend, idx unique (the pooler input gradient joins the
 * gradient of the first-token rows, models/vilbert_dialog.py:949, :964). */
int unimm_rows_add_f32(void* dst, const int32_t* idx, const float* src, int32_t n, int32_t H, void* stream);

/* du = dt * GELU'(u), bf16 flat [n], n % 8 == 0 (prediction-head transforms, :983-985, :1002-1004) */
int unimm_gelu_bwd(const void* dt, const void* u, void* du, int64_t n, void* stream);
/* scatter == 0: dst[i,:] = src[idx[i],:]; scatter != 0: dst[idx[i],:] = src[i,:]  (bf16 rows of H, idx unique).
 * Selects the labelled token rows the decoder runs on (the reference decodes all 256 rows and then
 * boolean-gathers, models/vilbert_dialog.py:1583-1584). */
int unimm_gather_rows(const void* src, const int32_t* idx, void* dst, int32_t n, int32_t H, int32_t scatter,
                      const int32_t* n_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Losses.  Row kernels (one workgroup per row) keep logits, log-sum-exp and 1-p in fp32.
 * `g` is a DEVICE pointer to the upstream gradient of the (shape-[1]) loss; inv_denom a host float.
 * ------------------------------------------------------------------------------------------- */
/* Token-level likelihood / unlikelihood (models/vilbert_dialog.py:1577-1595): per decoded row with
 * label y (-1 = ignore) and integer weight w: w>0 -> -w*log p_y; w==-1 -> -log(clamp(1-p_y, 1e-6)).
 * rowloss is the un-normalised contribution, rownll = -log p_y (generative scoring, val_lm.py:131-136),
 * lse the row's log-sum-exp.  The :1600-1604 CrossEntropy fallback is this with w = [y != -1]. */
int unimm_lm_loss_fwd(const float* logits, const int32_t* labels, const int32_t* weights, float* rowloss,
                      float* rownll, float* lse, int32_t n, int32_t V, int32_t ld, const int32_t* n_dev, void* stream);
/* dlogits (bf16 [n, ldd], columns >= V zero-filled) = g * inv_denom * d(rowloss)/d(logits); inv_dev (or NULL): the
 * denominator's reciprocal as a device word instead of the host float */
int unimm_lm_loss_bwd(const float* logits, const int32_t* labels, const int32_t* weights, const float* lse,
                      const float* g, float inv_denom, void* dlogits, int32_t n, int32_t V, int32_t ld,
                      int32_t ldd, const int32_t* n_dev, const float* inv_dev, void* stream);
/* Masked-region KL (models/vilbert_dialog.py:1569-1574): rowloss = [label==1] * sum_j t_j (log t_j - logp_j) */
int unimm_kl_loss_fwd(const float* pred, const float* target, const int32_t* label, float* rowloss, float* lse,
                      int32_t rows, int32_t C, int32_t ld, void* stream);
int unimm_kl_loss_bwd(const float* pred, const float* target, const int32_t* label, const float* lse,
                      const float* g, float inv_denom, void* dpred, int32_t rows, int32_t C, int32_t ld,
                      int32_t ldd, const float* inv_dev, void* stream);
/* Masked-region MSE, the predict_feature = True branch (models/vilbert_dialog.py:1562-1566): rowloss = [label==1] *
 * sum_j (pred_j - target_j)^2 / C (the reference divides the sum over selected ELEMENTS by their count; the caller divides the
 * sum of rowloss by max(#selected rows, 1)).  Backward: dpred = [label==1] g inv_denom 2 (pred - target) / C as bf16 [rows, ldd]
 * (out_split == 0, columns >= C zero) or as an x-type split operand [rows, 3 ldd] (out_split != 0, the fp32-accuracy mode). */
int unimm_mse_loss_fwd(const float* pred, const float* target, const int32_t* label, float* rowloss, int32_t rows, int32_t C,
                       int32_t ld, void* stream);
int unimm_mse_loss_bwd(const float* pred, const float* target, const int32_t* label, const float* g, float inv_denom,
                       void* dpred, int32_t rows, int32_t C, int32_t ld, int32_t ldd, int32_t out_split, void* stream);
/* Weighted 2-way cross-entropy, reduction 'mean' = sum w_y l / sum w_y (models/vilbert_dialog.py:1617-1621);
 * w0, w1 already divided by w0 (:1608). */
int unimm_nsp_loss_fwd(const float* logits, const int32_t* labels, float w0, float w1, float* loss, int32_t B,
                       int32_t ld, void* stream);
/* dlogits: fp32 [B, ldd] (columns 2.. zeroed); extra (or NULL): fp32 [B, 2] added to it -- a gradient that arrives
 * through the returned NSP scores (the ranking loss of dense_annotation_finetuning.py:263-293) */
int unimm_nsp_loss_bwd(const float* logits, const int32_t* labels, float w0, float w1, const float* g, const float* extra,
                       float* dlogits, int32_t B, int32_t ld, int32_t ldd, void* stream);
/* dst[0] = scale * sum(src) (fixed order, deterministic); dst[seg[i]] += sign * src[i] */
int unimm_reduce_sum(const float* src, int64_t n, float* dst, float scale, const int32_t* n_dev, const float* scale_dev,
                     void* stream);
int unimm_segment_sum(const float* src, const int32_t* seg, float* dst, int64_t n, float sign, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused AdamW over the flat arenas.  Replaces the optimizer of train.py:322-347 /
 * :458 (`pytorch_transformers.AdamW`, one parameter group per tensor: lr in {lr, image_lr} by
 * config/language_weights.json, weight_decay in {0.01, 0}); the lr values are whatever the caller's
 * scheduler (utils/optim_utils.py:8-26) set for this step.  Per element, in fp32:
 *   g *= grad_scale;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;
 *   p -= lr_g * [sqrt(1-b2^step)/(1-b1^step) if correct_bias] * m / (sqrt(v) + eps);  p -= lr_g * wd_g * p
 * group[i] (uint8, device) is the (lr, wd) group of elements [64 i, 64 i + 64); values >=
 * UNIMM_ADAMW_MAX_GROUPS mark parameters that never receive a gradient (skipped entirely, as the
 * reference skips `p.grad is None`).  w16 (bf16, may be NULL) receives the updated parameters: the
 * GEMM-operand copy, so no separate cast pass.  zero_grad != 0 clears g in the same pass. */
#define UNIMM_ADAMW_MAX_GROUPS 8
typedef struct {
  float* p;              /* [n] fp32 parameters (updated in place) */
  const float* g;        /* [n] fp32 gradients */
  float* m;              /* [n] fp32 exp_avg */
  float* v;              /* [n] fp32 exp_avg_sq */
  void* w16;             /* [n] bf16 copy of p or NULL */
  const uint8_t* group;  /* [n / 64] group id per 64-element chunk */
  int64_t n;             /* multiple of 64 */
  int32_t n_groups;      /* <= UNIMM_ADAMW_MAX_GROUPS */
  int32_t step;          /* 1-based step count t */
  double lr[UNIMM_ADAMW_MAX_GROUPS];           /* doubles: the scalar factors (1 - beta, lr * correction,    */
  double weight_decay[UNIMM_ADAMW_MAX_GROUPS]; /* lr * wd) are formed in double and rounded to fp32 once,    */
  double beta1, beta2, eps;                    /* as the reference's Python-float arithmetic does            */
  float grad_scale;      /* 1 / loss scale (or 1 / batch_multiply); 1 = none */
  int32_t correct_bias;
  int32_t zero_grad;
} unimm_adamw_args;

int unimm_adamw_step(const unimm_adamw_args* args, void* stream);

/* NeuralNDCG-transposed of `slates` independent slates of `n` answer options: value and gradient with
 * respect to the predictions in one launch.  Replaces the deterministic branch of
 * neuralNDCG_transposed (utils/rank_loss.py:518-581: deterministic_neural_sort :79-112, sinkhorn_scaling
 * :55-78, dcg :18-54) as dense_annotation_finetuning.py:286-288 calls it, and its autograd backward.
 * pred / truth fp32 [slates, n]; truth == pad_label marks a padded option.  Outputs per slate:
 * ndcg (0 when the slate has no relevant option), alive (1 / 0: ideal DCG != 0), iters (Sinkhorn sweeps
 * run), and dpred [slates, n] = d ndcg / d pred.  The caller forms loss = -sum(ndcg) / sum(alive). */
#define UNIMM_NDCG_MAX_OPTIONS 128
#define UNIMM_NDCG_MAX_ITER 64
typedef struct {
  const float* pred;
  const float* truth;
  float* ndcg;
  float* alive;
  float* dpred;
  int32_t* iters;
  int32_t slates, n;
  int32_t k;                    /* truncation rank; <= 0 = n */
  int32_t powered_relevancies;  /* gain 2^y - 1 (1) or y (0); the ideal DCG always uses 2^y - 1, as the reference does */
  int32_t max_iter;             /* 1 .. UNIMM_NDCG_MAX_ITER (reference default 50) */
  float pad_label, temperature, tol;
} unimm_ndcg_args;

int unimm_neural_ndcg(const unimm_ndcg_args* args, void* stream);

/* ---------------------------------------------------------------------------------------------
 * fp32-accuracy compute mode ("fp32x3", csrc/x3ops.hip).  The reference's dense-annotation fine-tune calls the model
 * WITHOUT autocast (dense_annotation_finetuning.py:253: fp32 end to end; north_star gates fp32 results at 1e-3).  gfx950
 * has no fast fp32 matrix path, so this mode runs every nn.Linear of the path on the SAME bf16 MFMA kernels
 * (unimm_gemm_nt / unimm_gemm_tn_grouped) over SPLIT operands, x = hi + lo, hi = bf16(x), lo = bf16(x - hi):
 *   activation operand ("x-type")  X3[M, 3 Kp] = [ hi(X) | lo(X) | hi(X) ]
 *   weight operand     ("w-type")  W3[N, 3 Kp] = [ hi(W) | hi(W) | lo(W) ]        Kp = K rounded up to 64, padding zero
 *   unimm_gemm_nt(x = X3, w = W3, K = 3 Kp, out fp32)  =  hi hi + lo hi + hi lo   (fp32 accumulate; lo lo ~ 2^-16 dropped)
 * and a weight gradient as three problems of unimm_gemm_tn_grouped over column planes of two x-type buffers
 * (dY.hi, X.hi), (dY.lo, X.hi), (dY.hi, X.lo), all accumulating into the same fp32 gradient.  Between two GEMMs
 * everything is fp32; the entry points below produce the next split operand from fp32 results, run LayerNorm / loss
 * backward on fp32 gradients, and compute the attention cores (models/vilbert_dialog.py:390-410, :519-539, :681-721)
 * in fp32 on the vector ALUs.
 * ------------------------------------------------------------------------------------------- */
enum { UNIMM_X3_COPY = 0, UNIMM_X3_ADD = 1, UNIMM_X3_GELU = 2, UNIMM_X3_MUL_DGELU = 3 };
typedef struct {
  const float* a;  /* [rows, cols] fp32, row stride lda */
  const float* b;  /* second operand of ADD / MUL_DGELU (row stride ldb) or NULL */
  float* out32;    /* or NULL: y as fp32 [rows, cols], row stride ld32 */
  void* out3;      /* or NULL: y as a split operand, bf16 [rows, 3 cp] (columns cols .. cp-1 of every plane zero) */
  int64_t rows;
  int32_t cols, cp;          /* cp % 64 == 0, cp >= cols */
  int32_t lda, ldb, ld32;
  int32_t op;                /* y = a | a + b | erf-GELU(a) (:115-121) | a * GELU'(b) */
  int32_t wtype;             /* 0: x-type planes (hi, lo, hi); 1: w-type planes (hi, hi, lo) */
} unimm_x3_split_args;
int unimm_x3_split(const unimm_x3_split_args* args, void* stream);
/* transposed w-type split of a weight: src fp32 [R, C] (row stride lds) -> dst bf16 [C, 3 Rp], dst[c][p Rp + r]; columns
 * r in [R, Rp) of every plane are not written (zero-fill dst once).  The operand of the input-gradient GEMMs. */
int unimm_x3_split_wt(const float* src, void* dst, int32_t R, int32_t C, int32_t lds, int32_t Rp, void* stream);
/* unimm_layernorm_bwd_partials on fp32 gradients: dy fp32 [M, H]; dx32 (or NULL) = fp32 gradient w.r.t. the pre-LayerNorm
 * sum; dxd3 (or NULL) = its dropout-masked copy as an x-type split operand [M, 3 H]; partials / blocks_out as there
 * (reduced by unimm_colpartials_finish_grouped).  H % 64 == 0. */
int unimm_x3_layernorm_bwd_partials(const float* dy, const float* x, const float* mean, const float* rstd, const float* gamma,
                                    float* dx32, void* dxd3, float* partials, int32_t M, int32_t H, uint32_t drop_key,
                                    uint32_t drop_thr, float drop_scale, uint32_t odrop_key, uint32_t odrop_thr,
                                    float odrop_scale, int32_t* blocks_out, const int32_t* m_dev, const uint32_t* drop_salt,
                                    void* stream);
/* unimm_layernorm_fwd that also writes the normalised rows as an x-type split operand y3 [M, 3 H] (H % 64 == 0); y32 (or NULL),
 * mean / rstd (both or neither), dropout as there. */
int unimm_x3_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y32, void* y3, float* mean, float* rstd,
                           int32_t M, int32_t H, float eps, uint32_t drop_key, uint32_t drop_thr, float drop_scale,
                           const uint32_t* drop_salt, void* stream);
/* unimm_embed_bwd with an fp32 upstream gradient */
int unimm_embed_bwd_f32(const unimm_embed_args* args, const float* dy, float* dword, float* dpos, float* dtype, float* dext,
                        float* dgamma, float* dbeta, float* partials, void* stream);
/* unimm_lm_loss_bwd / unimm_kl_loss_bwd writing x-type split operands [rows, 3 cp] (cp % 64 == 0, cp >= V / C) */
int unimm_x3_lm_loss_bwd(const float* logits, const int32_t* labels, const int32_t* weights, const float* lse, const float* g,
                         float inv_denom, void* out3, int32_t n, int32_t V, int32_t ld, int32_t cp, const int32_t* n_dev,
                         const float* inv_dev, void* stream);
int unimm_x3_kl_loss_bwd(const float* pred, const float* target, const int32_t* label, const float* lse, const float* g,
                         float inv_denom, void* out3, int32_t rows, int32_t C, int32_t ld, int32_t cp, const float* inv_dev,
                         void* stream);
/* dst[idx[r], :] += src[r, :], both fp32 (unimm_rows_add_f32 for an fp32 gradient stream) */
int unimm_x3_rows_add(float* dst, const int32_t* idx, const float* src, int32_t n, int32_t H, int32_t ldd, void* stream);
/* unimm_attn_fwd / unimm_attn_bwd with fp32 q / k / v / out / dout / dq / dk / dv (same argument structs, row strides in
 * fp32 elements, multiples of 4; every other field -- masks, lse, variable-length offsets, dropout -- as there). */
typedef struct {
  void* out3;                      /* forward: the context ALSO as an x-type split operand [rows, 3 cp3] (bf16), or NULL */
  void* dq3; void* dk3; void* dv3; /* backward: dQ / dK / dV as x-type split operands INSTEAD of fp32 (all three or none; the args'
                                    * dq / dk / dv may then be NULL); each points at its first column of plane 0 */
  int32_t ld3;                     /* row stride of the split buffers, bf16 elements (>= 3 cp3) */
  int32_t cp3;                     /* plane stride = columns per plane (multiple of 8, >= H * D); buffers 16-byte aligned, ld3 % 8 == 0 */
} unimm_x3_attn_planes;
/* planes (or NULL): what the next GEMM reads, written by the attention kernel itself instead of a unimm_x3_split pass over its
 * fp32 result.  Matrix-instruction kernels only (UNIMM_E_ARG under unimm_x3_attn_set_impl(0)). */
int unimm_x3_attn_fwd(const unimm_attn_args* args, const unimm_x3_attn_planes* planes, void* stream);
int unimm_x3_attn_bwd(const unimm_attn_bwd_args* args, const unimm_x3_attn_planes* planes, void* stream);
/* Which kernels the two entry points above launch: 1 (default) = the fp32 matrix-instruction kernels
 * (v_mfma_f32_16x16x4_f32, exact fp32 operands), 0 = the vector-ALU kernels of the first version (kept for A/B runs). */
int unimm_x3_attn_set_impl(int32_t impl);

/* Launch profiler for bench.py's `roofline` block: HIP events around every GEMM launch on its own
 * stream while enabled.  Variant index: 0..11 = unimm_gemm_nt (epilogue * 2 + out_f32), 12 = unimm_gemm_tn.
 * unimm_prof_collect synchronises the events and returns per-variant summed milliseconds, algorithmic
 * FLOPs (2*M*N*K per launch) and launch counts; arrays of >= 20 entries.  on = 1: every GEMM launch;
 * on = 2: unimm_gemm_tn launches only (the two event records per launch are host time, which a
 * launch-rate-bound step should not pay for all ~340 GEMM launches); 0 = off. */
int unimm_prof_enable(int32_t on);
int unimm_prof_collect(double* ms, double* flops, int32_t* count, int32_t nvar);
/* Caller-side tag (0 .. 7, default 0) carried by the unimm_gemm_nt launches that follow; unimm_prof_tagged returns, per tag, the
 * summed milliseconds / FLOPs / launch counts of the records the LAST unimm_prof_collect consumed.  The engine tags the launches
 * it issues from inside a BertConnectionLayer (models/vilbert_dialog.py:655-783) with 1: bench.py's
 * `roofline.coattention_gemms`, the figure north_star's ">= 40 % MFMA utilisation on the co-attention GEMMs" is judged by. */
int unimm_prof_tag(int32_t tag);
/* union_ms (or NULL): per tag > 0, the length of the UNION of the launches' execution intervals on one time axis -- launches
 * of two streams that ran side by side count once: the wall time during which at least one tagged GEMM was executing. */
int unimm_prof_tagged(double* ms, double* flops, int32_t* count, double* union_ms, int32_t ntags);

#ifdef __cplusplus
}
#endif
#endif /* UNIMM_HIP_H */
