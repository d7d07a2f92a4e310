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
  UNIMM_EPI_BIAS_DROP_RESID = 2, /* out = dropout(acc + bias) + aux  (:423-425, :466-468, ...)  */
  UNIMM_EPI_BIAS_RELU = 3,       /* poolers (:950-951, :965-966)                                */
  UNIMM_EPI_DGELU = 4,           /* out = acc * GELU'(aux)           (backward of :453-454)     */
  UNIMM_EPI_ADD = 5              /* out = acc + aux                  (residual gradient join)   */
};

typedef struct {
  const void* x;     /* [M, K] bf16 */
  const void* w;     /* [N, K] bf16 */
  const float* bias; /* [N] fp32 or NULL */
  const void* aux;   /* [M, N] bf16, epilogue operand (residual / pre-activation) or NULL */
  void* out;         /* [M, N] bf16, or fp32 when out_f32 != 0 */
  void* out2;        /* [M, N] bf16 second output of UNIMM_EPI_BIAS_GELU (row stride ldo) or NULL */
  int32_t M, N, K;
  int32_t ldx, ldw, ldaux, ldo;
  int32_t epilogue;
  int32_t out_f32;
  uint32_t drop_key; /* dropout (UNIMM_EPI_BIAS_DROP_RESID): keep iff mix32(idx ^ key) >= thr   */
  uint32_t drop_thr; /* p * 2^32; 0 disables                                                    */
  float drop_scale;  /* 1 / (1 - p)                                                             */
} unimm_gemm_nt_args;

int unimm_gemm_nt(const unimm_gemm_nt_args* args, void* stream);

/* GEMM, "TN": DW[N,K] += DY[M,N]^T . X[M,K] (fp32 atomics; caller zeroes DW once per step).
 * Weight-gradient half of every nn.Linear backward on the path.  Rows of DY / X must be readable
 * up to round_up(N, 8) / round_up(K, 8) columns; lddy, ldx % 8 == 0. */
typedef struct {
  const void* dy; /* [M, N] bf16 */
  const void* x;  /* [M, K] bf16 */
  float* dw;      /* [N, K] fp32, row stride lddw */
  int32_t M, N, K;
  int32_t lddy, ldx, lddw;
} unimm_gemm_tn_args;

int unimm_gemm_tn(const unimm_gemm_tn_args* args, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNIMM_HIP_H */
