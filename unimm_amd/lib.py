"""ctypes binding of libunimm_hip.so (the C ABI in include/unimm_hip.h).

There is no CPU fallback: if the library is missing or a kernel call fails, this raises.
PyTorch is used only for device memory and streams; every compute call below goes straight to a
hand-written HIP kernel on torch's current stream."""
from __future__ import annotations

import ctypes as C
import os
import threading

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libunimm_hip.so")
ABI_VERSION = 18

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_DROP_RESID, EPI_BIAS_RELU, EPI_DGELU, EPI_ADD, EPI_MUL, EPI_BIAS_GELU_DG = range(8)

_ERR = {-1: "UNIMM_E_ARG", -2: "UNIMM_E_SHAPE", -3: "UNIMM_E_ALIGN", -4: "UNIMM_E_HIP"}


class UnimmHipError(RuntimeError):
    pass


class GemmNtArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("aux", C.c_void_p),
                ("out", C.c_void_p), ("out2", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("ldx", C.c_int32), ("ldw", C.c_int32), ("ldaux", C.c_int32), ("ldo", C.c_int32),
                ("epilogue", C.c_int32), ("out_f32", C.c_int32),
                ("drop_key", C.c_uint32), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float),
                ("aux_mean", C.c_void_p), ("aux_rstd", C.c_void_p), ("aux_gamma", C.c_void_p), ("aux_beta", C.c_void_p),
                ("tile", C.c_int32), ("drop_salt", C.c_void_p),
                ("splitk_ws", C.c_void_p), ("splitk_ws_bytes", C.c_int64), ("splitk", C.c_int32)]


class GemmTnArgs(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p), ("dbias", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lddy", C.c_int32), ("ldx", C.c_int32), ("lddw", C.c_int32), ("m_dev", C.c_void_p), ("overwrite", C.c_int32)]


_lib = None


def lib():
    """Load the shared library once; fail loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("UNIMM_HIP_LIB", LIB_PATH)      # A/B runs of two builds in one process tree (tools only)
    if not os.path.exists(path):
        raise UnimmHipError(
            f"{path} is missing: build it with `python -m unimm_amd.build` (hipcc, gfx950). "
            "unimm_amd has no CPU or PyTorch fallback for its kernels.")
    L = C.CDLL(path)
    L.unimm_version.restype = C.c_int
    L.unimm_arch.restype = C.c_char_p
    if L.unimm_version() != ABI_VERSION:
        raise UnimmHipError(f"libunimm_hip.so ABI {L.unimm_version()} != expected {ABI_VERSION}: rebuild")
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name == "unimm_colpartials_bytes":
            fn.restype = C.c_int64
        elif name != "unimm_arch":
            fn.restype = C.c_int
    VP, I32, U32, F32, I64 = C.c_void_p, C.c_int32, C.c_uint32, C.c_float, C.c_int64
    L.unimm_gemm_nt.argtypes = [VP, VP]
    L.unimm_attn_fwd.argtypes = [VP, VP]
    L.unimm_attn_bwd.argtypes = [VP, VP]
    L.unimm_x3_split.argtypes = [VP, VP]
    L.unimm_x3_attn_fwd.argtypes = [VP, VP, VP]
    L.unimm_x3_attn_bwd.argtypes = [VP, VP, VP]
    L.unimm_attn_probs.argtypes = [VP, VP, VP]
    L.unimm_gemm_tn_grouped.argtypes = [VP, I32, VP]
    L.unimm_gemm_tn_grouped_ws.argtypes = [VP, I32, I32, VP, I64, VP]
    L.unimm_colpartials_finish_grouped.argtypes = [VP, I32, VP]
    L.unimm_layernorm_fwd.argtypes = [VP] * 7 + [I32, I32, F32, U32, U32, F32, VP, VP]
    L.unimm_x3_layernorm_fwd.argtypes = [VP] * 7 + [I32, I32, F32, U32, U32, F32, VP, VP]
    L.unimm_layernorm_bwd_partials.argtypes = [VP] * 8 + [I32, I32, U32, U32, F32, U32, U32, F32, VP, VP, VP, VP]
    L.unimm_layernorm_bwd.argtypes = [VP] * 11 + [I32, I32, U32, U32, F32, U32, U32, F32, VP, VP]
    _lib = L
    return L


# every symbol include/unimm_hip.h declares (tests check the .so exports each of them)
SYMBOLS = ["unimm_version", "unimm_arch", "unimm_gemm_nt", "unimm_gemm_tn", "unimm_gemm_tn_grouped", "unimm_attn_fwd", "unimm_attn_bwd",
           "unimm_mask_pack", "unimm_layernorm_fwd", "unimm_colpartials_bytes", "unimm_layernorm_bwd",
           "unimm_embed_fwd", "unimm_embed_bwd", "unimm_colsum", "unimm_cast_f32_bf16", "unimm_transpose_cast",
           "unimm_pack_image", "unimm_mul_dropout", "unimm_mul_dropout_bwd", "unimm_lm_loss_fwd",
           "unimm_lm_loss_bwd", "unimm_kl_loss_fwd", "unimm_kl_loss_bwd", "unimm_nsp_loss_fwd",
           "unimm_nsp_loss_bwd", "unimm_reduce_sum", "unimm_segment_sum", "unimm_gelu_bwd", "unimm_gather_rows", "unimm_prof_enable", "unimm_prof_collect", "unimm_adamw_step", "unimm_transpose_cast_grouped", "unimm_mask_synth", "unimm_neural_ndcg", "unimm_plan_lengths", "unimm_plan_build", "unimm_layernorm_bwd_partials",
           "unimm_colpartials_finish_grouped", "unimm_gemm_tn_grouped_ws", "unimm_linear_f32", "unimm_rows_add_f32", "unimm_transpose_bf16", "unimm_sum_slabs_bf16", "unimm_attn_probs",
           # the fp32-accuracy mode (csrc/x3ops.hip)
           "unimm_x3_split", "unimm_x3_split_wt", "unimm_x3_layernorm_bwd_partials", "unimm_embed_bwd_f32", "unimm_x3_lm_loss_bwd",
           "unimm_x3_kl_loss_bwd", "unimm_x3_rows_add", "unimm_x3_attn_fwd", "unimm_x3_attn_bwd", "unimm_x3_attn_set_impl", "unimm_x3_layernorm_fwd", "unimm_prof_tag", "unimm_prof_tagged",
           "unimm_sum_dropout", "unimm_sum_dropout_bwd", "unimm_mse_loss_fwd", "unimm_mse_loss_bwd", "unimm_host_mask_pack", "unimm_host_memcpy"]


def _check(rc, what):
    if rc != 0:
        raise UnimmHipError(f"{what} failed: {_ERR.get(rc, rc)}")


_tls = threading.local()      # per thread: argument structs of the hot wrappers (the C side reads them before it returns)
                              # and the raw stream of the innermost stream_scope() (None = ask torch on every call)


def scoped_stream():
    """The raw stream object the calling thread's innermost `stream_scope` installed, or None."""
    return getattr(_tls, "stream", None)


def _stream():
    s = getattr(_tls, "stream", None)
    if s is not None:
        return s
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class stream_scope:
    """`with stream_scope(stream):` pins the stream every launch inside the block goes to (the block must also run
    under `torch.cuda.stream(stream)` / on that current stream for torch's own ops).  The engine brackets forward and
    backward with it: `torch.cuda.current_stream()` costs ~4 us and was asked ~240 times per step."""

    def __init__(self, stream):
        self.ptr = C.c_void_p(stream.cuda_stream)

    def __enter__(self):
        self.old, _tls.stream = getattr(_tls, "stream", None), self.ptr
        return self

    def __exit__(self, *exc):
        _tls.stream = self.old
        return False


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _P(t):
    """raw device address for an entry point with declared argtypes / a c_void_p struct field (None = NULL)"""
    return t.data_ptr() if t is not None else None


def _salt(*drops):
    """Device address of the salt word a dropout triple may carry as a 4th element: (key, thr, scale[, salt tensor])."""
    for d in drops:
        if d is not None and len(d) > 3 and d[3] is not None:
            return d[3].data_ptr()
    return None


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise UnimmHipError("unimm_amd kernels need device tensors (got a CPU tensor)")


def gemm_nt(x, w, out, bias=None, epilogue=EPI_BIAS, aux=None, out2=None, drop=None, M=None, N=None, K=None, aux_ln=None, tile=0,
            splitk=0, splitk_ws=None):
    """out[M,N] = epi(x[M,K] @ w[N,K]^T).  x/w bf16 2-D (row stride = stride(0)); out bf16 or fp32.
    aux_ln = (mean[M], rstd[M], gamma[N], beta[N]): the DROP_RESID residual is LayerNorm(aux) computed on the fly.
    tile: per-call tuning code (include/unimm_hip.h: unimm_gemm_nt_args.tile); 0 = automatic."""
    _dev(x, w, out, bias, aux, out2)
    st = getattr(_tls, "nt", None)
    if st is None:
        a = GemmNtArgs()
        st = _tls.nt = (a, C.addressof(a), lib().unimm_gemm_nt)
    a, addr, fn = st
    a.x, a.w, a.out = x.data_ptr(), w.data_ptr(), out.data_ptr()
    a.bias = bias.data_ptr() if bias is not None else None
    a.out2 = out2.data_ptr() if out2 is not None else None
    if aux is not None:
        a.aux, a.ldaux = aux.data_ptr(), aux.stride(0)
    else:
        a.aux, a.ldaux = None, 0
    a.M = x.shape[0] if M is None else M
    a.N = w.shape[0] if N is None else N
    a.K = x.shape[1] if K is None else K
    a.ldx, a.ldw, a.ldo = x.stride(0), w.stride(0), out.stride(0)
    a.epilogue = epilogue
    a.tile = tile
    a.out_f32 = 1 if out.dtype == torch.float32 else 0
    a.drop_key, a.drop_thr, a.drop_scale = drop[:3] if drop is not None else (0, 0, 0.0)
    a.drop_salt = _salt(drop)
    if splitk_ws is not None and splitk not in (0, 1):          # (see include/unimm_hip.h: unimm_gemm_nt_args.splitk)
        a.splitk, a.splitk_ws, a.splitk_ws_bytes = splitk, splitk_ws.data_ptr(), splitk_ws.numel() * splitk_ws.element_size()
    else:
        a.splitk, a.splitk_ws, a.splitk_ws_bytes = 0, None, 0
    if aux_ln is not None:
        _dev(*aux_ln)
        a.aux_mean, a.aux_rstd, a.aux_gamma, a.aux_beta = (t.data_ptr() for t in aux_ln)
    else:
        a.aux_mean = a.aux_rstd = a.aux_gamma = a.aux_beta = None
    rc = fn(addr, _stream())
    if rc != 0:
        _check(rc, "unimm_gemm_nt")
    return out


def gemm_tn(dy, x, dw, M=None, N=None, K=None, dbias=None, m_dev=None):
    """dw[N,K] += dy[M,N]^T @ x[M,K] (fp32 atomics); dbias[N] += colsum(dy) when given.
    m_dev: int32 device word with the rows actually present (M is then a capacity)."""
    _dev(dy, x, dw, dbias, m_dev)
    a = GemmTnArgs()
    a.dy, a.x, a.dw, a.dbias = _ptr(dy), _ptr(x), _ptr(dw), _ptr(dbias)
    a.m_dev = _P(m_dev)
    a.M = dy.shape[0] if M is None else M
    a.N = dy.shape[1] if N is None else N
    a.K = x.shape[1] if K is None else K
    a.lddy, a.ldx, a.lddw = dy.stride(0), x.stride(0), dw.stride(0)
    _check(lib().unimm_gemm_tn(C.byref(a), _stream()), "unimm_gemm_tn")
    return dw


def gemm_tn_grouped(problems, shared=None, ws=None):
    """problems: list of (dy, x, dw, M, N, K, dbias[, m_dev[, overwrite]]) -- every dw[N,K] += dy[:M,:N]^T @ x[:M,:K] in as few
    launches as possible (one per <= 12 problems of the same tile class).  overwrite: dw is known to be zero and has no other
    contributor -> plain stores instead of atomics (unimm_gemm_tn_args.overwrite).
    shared: the launches run beside another stream's kernels (split heuristic hint; None = the process-wide default).
    ws: zero-initialised uint8 device tensor private to the launch stream (partial-tile slabs + arrival counters);
    None = every split adds its partial tile with fp32 atomics."""
    n = len(problems)
    if n == 0:
        return
    arr = (GemmTnArgs * n)()
    for a, (dy, x, dw, M, N, K, dbias, *rest) in zip(arr, problems):
        _dev(dy, x, dw, dbias)
        a.dy, a.x, a.dw, a.dbias = dy.data_ptr(), x.data_ptr(), dw.data_ptr(), (dbias.data_ptr() if dbias is not None else None)
        a.m_dev = rest[0].data_ptr() if rest and rest[0] is not None else None
        a.overwrite = 1 if (len(rest) > 1 and rest[1]) else 0
        a.M = dy.shape[0] if M is None else M
        a.N = dy.shape[1] if N is None else N
        a.K = x.shape[1] if K is None else K
        a.lddy, a.ldx, a.lddw = dy.stride(0), x.stride(0), dw.stride(0)
    if shared is None and ws is None:
        _check(lib().unimm_gemm_tn_grouped(C.addressof(arr), n, _stream()), "unimm_gemm_tn_grouped")
        return
    _check(lib().unimm_gemm_tn_grouped_ws(C.addressof(arr), n, int(shared or 0), _P(ws),
                                          ws.numel() if ws is not None else 0, _stream()), "unimm_gemm_tn_grouped_ws")


# ---------------------------------------------------------------------------------------------
# attention
# ---------------------------------------------------------------------------------------------
class AttnArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p),
                ("lse", C.c_void_p), ("mask", C.c_void_p),
                ("q_off", C.c_void_p), ("q_len", C.c_void_p), ("k_off", C.c_void_p), ("k_len", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32), ("D", C.c_int32),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32),
                ("mask_q_stride", C.c_int32), ("mask_b_stride", C.c_int32), ("scale", C.c_float),
                ("drop_key", C.c_uint32), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float), ("drop_salt", C.c_void_p),
                ("ks_off", C.c_void_p), ("ks_len", C.c_void_p), ("ks_ins", C.c_int32), ("order", C.c_void_p)]


class AttnBwdArgs(C.Structure):
    _fields_ = [("q", C.c_void_p), ("k", C.c_void_p), ("v", C.c_void_p), ("out", C.c_void_p), ("dout", C.c_void_p),
                ("lse", C.c_void_p), ("delta", C.c_void_p),
                ("dq", C.c_void_p), ("dk", C.c_void_p), ("dv", C.c_void_p), ("mask", C.c_void_p),
                ("q_off", C.c_void_p), ("q_len", C.c_void_p), ("k_off", C.c_void_p), ("k_len", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("Tq", C.c_int32), ("Tk", C.c_int32), ("D", C.c_int32),
                ("ldq", C.c_int32), ("ldk", C.c_int32), ("ldv", C.c_int32), ("ldo", C.c_int32), ("lddo", C.c_int32),
                ("lddq", C.c_int32), ("lddk", C.c_int32), ("lddv", C.c_int32),
                ("mask_q_stride", C.c_int32), ("mask_b_stride", C.c_int32), ("scale", C.c_float),
                ("drop_key", C.c_uint32), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float), ("drop_salt", C.c_void_p),
                ("order", C.c_void_p)]


def _item_order(qvar, kvar):
    """The item order of an attention launch (unimm_attn_args.order): element 3 of a variable-length descriptor, if it has one."""
    for v in (qvar, kvar):
        if v is not None and len(v) > 3 and v[3] is not None:
            return v[3].data_ptr()
    return None


NO_DROP = (0, 0, 1.0)


def attn_fwd(q, k, v, out, lse, mask, B, H, Tq, Tk, D, scale, mask_q_stride, mask_b_stride, drop=NO_DROP,
             qvar=None, kvar=None, kshared=None):
    """q/k/v/out: 2-D bf16 views [rows, >=H*D] (row stride = stride(0)); mask: packed uint32 words.
    qvar / kvar: (offsets, lengths) int32 [B] tensors for the variable-length layout, or None.
    kshared: (offsets, lengths, ins) -- a shared key/value segment spliced into every sequence's keys after its first `ins`
    private rows (unimm_attn_args.ks_*; needs kvar, inference only)."""
    _dev(q, k, v, out, lse, mask)
    st = getattr(_tls, "af", None)
    if st is None:
        a = AttnArgs()
        st = _tls.af = (a, C.addressof(a), lib().unimm_attn_fwd)
    a, addr, fn = st
    a.q, a.k, a.v, a.out, a.lse, a.mask = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _P(lse), mask.data_ptr()
    a.q_off, a.q_len = (qvar[0].data_ptr(), qvar[1].data_ptr()) if qvar is not None else (None, None)
    a.k_off, a.k_len = (kvar[0].data_ptr(), kvar[1].data_ptr()) if kvar is not None else (None, None)
    if kshared is not None:
        a.ks_off, a.ks_len, a.ks_ins = kshared[0].data_ptr(), kshared[1].data_ptr(), int(kshared[2])
    else:
        a.ks_off, a.ks_len, a.ks_ins = None, None, 0
    a.order = _item_order(qvar, kvar)
    a.B, a.H, a.Tq, a.Tk, a.D = B, H, Tq, Tk, D
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.mask_q_stride, a.mask_b_stride, a.scale = mask_q_stride, mask_b_stride, scale
    a.drop_key, a.drop_thr, a.drop_scale = drop[:3]
    a.drop_salt = _salt(drop)
    rc = fn(addr, _stream())
    if rc != 0:
        _check(rc, "unimm_attn_fwd")


def attn_probs(q, k, probs, mask, B, H, Tq, Tk, D, scale, mask_q_stride, mask_b_stride, drop=NO_DROP):
    """probs: fp32 [B, H, Tq, Tk] = dropout(softmax(q k^T scale + additive mask)); fixed row layout (diagnostic output)."""
    _dev(q, k, probs, mask)
    a = AttnArgs()
    a.q, a.k, a.v, a.out, a.lse, a.mask = q.data_ptr(), k.data_ptr(), None, None, None, mask.data_ptr()
    a.q_off = a.q_len = a.k_off = a.k_len = None
    a.ks_off = a.ks_len = None
    a.ks_ins = 0
    a.order = None
    a.B, a.H, a.Tq, a.Tk, a.D = B, H, Tq, Tk, D
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), 0, 0
    a.mask_q_stride, a.mask_b_stride, a.scale = mask_q_stride, mask_b_stride, scale
    a.drop_key, a.drop_thr, a.drop_scale = drop[:3]
    a.drop_salt = _salt(drop)
    _check(lib().unimm_attn_probs(C.byref(a), _ptr(probs), _stream()), "unimm_attn_probs")


def attn_bwd(q, k, v, out, dout, lse, delta, dq, dk, dv, mask, B, H, Tq, Tk, D, scale, mask_q_stride,
             mask_b_stride, drop=NO_DROP, qvar=None, kvar=None):
    _dev(q, k, v, out, dout, lse, delta, dq, dk, dv, mask)
    st = getattr(_tls, "ab", None)
    if st is None:
        a = AttnBwdArgs()
        st = _tls.ab = (a, C.addressof(a), lib().unimm_attn_bwd)
    a, addr, fn = st
    a.q_off, a.q_len = (qvar[0].data_ptr(), qvar[1].data_ptr()) if qvar is not None else (None, None)
    a.k_off, a.k_len = (kvar[0].data_ptr(), kvar[1].data_ptr()) if kvar is not None else (None, None)
    a.q, a.k, a.v, a.out, a.dout = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr()
    a.lse, a.delta, a.dq, a.dk, a.dv, a.mask = (lse.data_ptr(), delta.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                                                mask.data_ptr())
    a.B, a.H, a.Tq, a.Tk, a.D = B, H, Tq, Tk, D
    a.ldq, a.ldk, a.ldv, a.ldo, a.lddo = q.stride(0), k.stride(0), v.stride(0), out.stride(0), dout.stride(0)
    a.lddq, a.lddk, a.lddv = dq.stride(0), dk.stride(0), dv.stride(0)
    a.mask_q_stride, a.mask_b_stride, a.scale = mask_q_stride, mask_b_stride, scale
    a.drop_key, a.drop_thr, a.drop_scale = drop[:3]
    a.drop_salt = _salt(drop)
    a.order = _item_order(qvar, kvar)
    rc = fn(addr, _stream())
    if rc != 0:
        _check(rc, "unimm_attn_bwd")


# ---------------------------------------------------------------------------------------------
# row kernels
# ---------------------------------------------------------------------------------------------
_DT = {torch.bool: 0, torch.uint8: 0, torch.int32: 1, torch.int64: 2, torch.float32: 3}


def mask_pack(mask, out=None):
    """0/1 mask [..., T] -> uint32 words [..., ceil(T/32)] (stored in an int32 tensor)."""
    _dev(mask)
    if mask.dtype not in _DT:
        raise UnimmHipError(f"mask dtype {mask.dtype} not supported (bool, uint8, int32, int64, float32)")
    mask = mask.contiguous()
    t = mask.shape[-1]
    rows = mask.numel() // t
    nw = (t + 31) // 32
    if out is None:
        out = torch.empty(mask.shape[:-1] + (nw,), dtype=torch.int32, device=mask.device)
    _check(lib().unimm_mask_pack(_ptr(mask), _DT[mask.dtype], _ptr(out), C.c_int64(rows), C.c_int32(t), _stream()),
           "unimm_mask_pack")
    return out


def host_mask_pack(mask, out=None, threads=0):
    """The words of `mask_pack` for a mask in HOST memory, computed on the host (unimm_host_mask_pack): 0/1 mask [..., T] ->
    int32 words [..., ceil(T/32)] in `out` (e.g. a pinned staging buffer) or a new CPU tensor."""
    if mask.is_cuda:
        raise UnimmHipError("host_mask_pack takes a CPU tensor (device masks: mask_pack)")
    if mask.dtype not in _DT:
        raise UnimmHipError(f"mask dtype {mask.dtype} not supported (bool, uint8, int32, int64, float32)")
    mask = mask.contiguous()
    t = mask.shape[-1]
    rows = mask.numel() // t
    nw = (t + 31) // 32
    if out is None:
        out = torch.empty(mask.shape[:-1] + (nw,), dtype=torch.int32)
    if out.is_cuda or out.dtype != torch.int32 or out.numel() != rows * nw or not out.is_contiguous():
        raise UnimmHipError("host_mask_pack: out must be a contiguous int32 CPU tensor of rows x ceil(T/32) words")
    _check(lib().unimm_host_mask_pack(C.c_void_p(mask.data_ptr()), _DT[mask.dtype], C.c_void_p(out.data_ptr()), C.c_int64(rows),
                                      C.c_int32(t), C.c_int32(threads)), "unimm_host_mask_pack")
    return out


def host_copy(dst, src, threads=0):
    """dst.copy_(src) for two CPU tensors of one shape and dtype, both contiguous, on several threads (unimm_host_memcpy);
    anything else falls back to torch's copy_."""
    if (dst.is_cuda or src.is_cuda or dst.dtype != src.dtype or dst.shape != src.shape or not dst.is_contiguous()
            or not src.is_contiguous() or src.numel() * src.element_size() < (8 << 20)):
        dst.copy_(src)
        return dst
    _check(lib().unimm_host_memcpy(C.c_void_p(dst.data_ptr()), C.c_void_p(src.data_ptr()), C.c_int64(src.numel() * src.element_size()),
                                   C.c_int32(threads)), "unimm_host_memcpy")
    return dst


def colpartials_bytes(H):
    return int(lib().unimm_colpartials_bytes(C.c_int32(H)))


class FinishDesc(C.Structure):
    _fields_ = [("partials", C.c_void_p), ("dst", C.c_void_p * 4), ("blocks", C.c_int32), ("nq", C.c_int32), ("H", C.c_int32),
                ("pad_", C.c_int32)]


def layernorm_bwd_partials(dy, x, mean, rstd, gamma, dx, dx_drop, partials, M, H, drop=None, out_drop=None, m_dev=None):
    """LayerNorm backward row kernel only; returns the number of partial blocks (see unimm_layernorm_bwd_partials)."""
    drop = drop or NO_DROP
    out_drop = out_drop or NO_DROP
    _dev(dy, x, mean, rstd, gamma, dx, dx_drop, partials)
    blocks = C.c_int32(0)
    rc = lib().unimm_layernorm_bwd_partials(dy.data_ptr(), x.data_ptr(), mean.data_ptr(), rstd.data_ptr(), gamma.data_ptr(),
                                            dx.data_ptr(), _P(dx_drop), partials.data_ptr(), M, H, drop[0], drop[1], drop[2],
                                            out_drop[0], out_drop[1], out_drop[2], C.addressof(blocks), _P(m_dev),
                                            _salt(drop, out_drop), _stream())
    if rc != 0:
        _check(rc, "unimm_layernorm_bwd_partials")
    return blocks.value


def colpartials_finish_grouped(pending):
    """pending: list of (partials, blocks, H, [dst tensors or None, up to 4]) -> their column sums added in one launch."""
    n = len(pending)
    if n == 0:
        return
    arr = (FinishDesc * n)()
    for d, (part, blocks, H, dsts) in zip(arr, pending):
        d.partials, d.blocks, d.nq, d.H = part.data_ptr(), blocks, len(dsts), H
        for q, t in enumerate(dsts):
            d.dst[q] = t.data_ptr() if t is not None else None
    _check(lib().unimm_colpartials_finish_grouped(C.addressof(arr), n, _stream()), "unimm_colpartials_finish_grouped")


def layernorm_fwd(x, gamma, beta, y32, y16, mean, rstd, M, H, eps=1e-12, drop=NO_DROP):
    _dev(x, gamma, beta, y32, y16, mean, rstd)
    rc = lib().unimm_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _P(y32), _P(y16), _P(mean), _P(rstd), M, H, eps,
                                   drop[0], drop[1], drop[2], _salt(drop), _stream())
    if rc != 0:
        _check(rc, "unimm_layernorm_fwd")


def layernorm_bwd(dy, x, mean, rstd, gamma, dx, dx_drop, dgamma, dbeta, dbias, partials, M, H, drop=NO_DROP,
                  out_drop=NO_DROP):
    _dev(dy, x, mean, rstd, gamma, dx, dx_drop, dgamma, dbeta, dbias, partials)
    _check(lib().unimm_layernorm_bwd(_P(dy), _P(x), _P(mean), _P(rstd), _P(gamma), _P(dx), _P(dx_drop),
                                     _P(dgamma), _P(dbeta), _P(dbias), _P(partials), M, H, drop[0], drop[1], drop[2],
                                     out_drop[0], out_drop[1], out_drop[2], _salt(drop, out_drop), _stream()), "unimm_layernorm_bwd")


class EmbedArgs(C.Structure):
    _fields_ = [("ids", C.c_void_p), ("pos", C.c_void_p), ("typ", C.c_void_p),
                ("word", C.c_void_p), ("post", C.c_void_p), ("type", C.c_void_p), ("ext", C.c_void_p),
                ("gamma", C.c_void_p), ("beta", C.c_void_p),
                ("M", C.c_int32), ("H", C.c_int32), ("type_vocab", C.c_int32), ("eps", C.c_float),
                ("drop_key", C.c_uint32), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float),
                ("m_dev", C.c_void_p), ("rows", C.c_void_p), ("drop_salt", C.c_void_p)]


def _embed_args(ids, pos, typ, word, post, type_, ext, gamma, beta, M, H, type_vocab, eps, drop, m_dev=None, rows=None):
    _dev(ids, pos, typ, word, post, type_, ext, gamma, beta, m_dev, rows)
    a = EmbedArgs()
    a.ids, a.pos, a.typ = _ptr(ids), _ptr(pos), _ptr(typ)
    a.word, a.post, a.type, a.ext = _ptr(word), _ptr(post), _ptr(type_), _ptr(ext)
    a.gamma, a.beta = _ptr(gamma), _ptr(beta)
    a.M, a.H, a.type_vocab, a.eps = M, H, type_vocab, eps
    a.drop_key, a.drop_thr, a.drop_scale = drop[:3]
    a.drop_salt = _salt(drop)
    a.m_dev, a.rows = _P(m_dev), _P(rows)
    return a


def embed_fwd(ids, pos, typ, word, post, type_, ext, gamma, beta, y32, y16, M, H, type_vocab=2, eps=1e-12, drop=NO_DROP,
              m_dev=None, rows=None):
    """rows (int64 [M]): row r embeds ids / pos / typ at index rows[r]; m_dev: int32 device word, rows actually present."""
    a = _embed_args(ids, pos, typ, word, post, type_, ext, gamma, beta, M, H, type_vocab, eps, drop, m_dev, rows)
    _dev(y32, y16)
    _check(lib().unimm_embed_fwd(C.byref(a), _ptr(y32), _ptr(y16), _stream()), "unimm_embed_fwd")


def embed_bwd(ids, pos, typ, word, post, type_, ext, gamma, beta, dy, dword, dpos, dtype, dext, dgamma, dbeta,
              partials, M, H, type_vocab=2, eps=1e-12, drop=NO_DROP, m_dev=None, rows=None):
    a = _embed_args(ids, pos, typ, word, post, type_, ext, gamma, beta, M, H, type_vocab, eps, drop, m_dev, rows)
    _dev(dy, dword, dpos, dtype, dext, dgamma, dbeta, partials)
    _check(lib().unimm_embed_bwd(C.byref(a), _ptr(dy), _ptr(dword), _ptr(dpos), _ptr(dtype), _ptr(dext), _ptr(dgamma),
                                 _ptr(dbeta), _ptr(partials), _stream()), "unimm_embed_bwd")


def colsum(dy, db, M, N):
    _dev(dy, db)
    _check(lib().unimm_colsum(_ptr(dy), _ptr(db), C.c_int32(M), C.c_int32(N), C.c_int32(dy.stride(0)), _stream()),
           "unimm_colsum")


def cast_f32_bf16(src, dst, n=None):
    _dev(src, dst)
    _check(lib().unimm_cast_f32_bf16(_ptr(src), _ptr(dst), C.c_int64(src.numel() if n is None else n), _stream()),
           "unimm_cast_f32_bf16")


def mask_synth(mode, length, nans, T):
    """Packed text mask [B, T, ceil(T/32)] and co-attention key mask [B, ceil(T/32)] from int32 descriptors."""
    _dev(mode, length, nans)
    B, nw = mode.numel(), (T + 31) // 32
    text = torch.empty((B, T, nw), dtype=torch.int32, device=mode.device)
    co = torch.empty((B, nw), dtype=torch.int32, device=mode.device)
    _check(lib().unimm_mask_synth(_ptr(mode), _ptr(length), _ptr(nans), _ptr(text), _ptr(co), C.c_int32(B), C.c_int32(T),
                                  _stream()), "unimm_mask_synth")
    return text, co


def plan_lengths(text_mask, co_mask, R, labels, weights, nsp_weight, B, T, image_label=None):
    """-> int32 header [3B+2] on the device (see include/unimm_hip.h: unimm_plan_lengths).  text_mask / co_mask are
    (words, q_stride, b_stride) tuples or None."""
    tw, tq, tb = text_mask if text_mask is not None else (None, 0, 0)
    cw, cq, cb = co_mask if co_mask is not None else (None, 0, 0)
    ref = tw if tw is not None else (cw if cw is not None else labels)
    header = torch.zeros(3 * B + 2, dtype=torch.int32, device=ref.device)
    _dev(tw, cw, labels, weights, nsp_weight, image_label)
    _check(lib().unimm_plan_lengths(_ptr(tw), C.c_int32(tq), C.c_int32(tb), _ptr(cw), C.c_int32(cq), C.c_int32(cb), C.c_int32(R),
                                    _ptr(labels), _ptr(weights), _ptr(nsp_weight), _ptr(image_label), C.c_int32(B), C.c_int32(T),
                                    _ptr(header),
                                    _stream()), "unimm_plan_lengths")
    return header


def plan_build(header, labels, weights, B, T, Mv, n_lm, want_rows=True, dims=None):
    """-> dict(off, lens, rows, inv, lm_pos, lm_idx, lm_label, lm_weight) built on the device from the header.
    Mv / n_lm are CAPACITIES (>= the header's totals): the lists' unused tails get safe values.  dims = (int32 [>=4],
    fp32 [>=2]) device tensors that receive the step's real counts and loss denominators (unimm_hip.h; int32 needs >= 4
    words: [3] = 1 when the batch did not fit the capacities)."""
    dev = header.device
    i32 = lambda n: torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    off, lens = i32(B), i32(B)
    rows = torch.empty(Mv, dtype=torch.int64, device=dev) if want_rows else None
    inv = torch.empty(B * T, dtype=torch.int64, device=dev) if want_rows else None
    lm = [i32(n_lm) for _ in range(4)] if (n_lm > 0 and labels is not None) else [None] * 4
    di, df = dims if dims is not None else (None, None)
    _dev(di, df)
    order = i32(B)                     # the sequences by length, longest first: the item order of the attention launches
    _check(lib().unimm_plan_build(_ptr(header), _ptr(labels), _ptr(weights), C.c_int32(B), C.c_int32(T), _ptr(off), _ptr(lens),
                                  _ptr(rows), _ptr(inv), *[_ptr(t) for t in lm], C.c_int32(Mv if want_rows else 0),
                                  C.c_int32(n_lm if lm[0] is not None else 0), _ptr(di), _ptr(df), _ptr(order), _stream()),
           "unimm_plan_build")
    return dict(off=off, lens=lens, rows=rows, inv=inv, lm_pos=lm[0], lm_idx=lm[1], lm_label=lm[2], lm_weight=lm[3], order=order)


def transpose_cast(src, dst, R, C_, ldd):
    _dev(src, dst)
    _check(lib().unimm_transpose_cast(_ptr(src), _ptr(dst), C.c_int32(R), C.c_int32(C_), C.c_int32(ldd), _stream()),
           "unimm_transpose_cast")


def transpose_bf16(src, dst, R, C_):
    """dst[c, r] = src[r, c] (bf16 [R, C] -> bf16 [C, ldd >= R], surplus columns zero)."""
    _dev(src, dst)
    _check(lib().unimm_transpose_bf16(_ptr(src), _ptr(dst), C.c_int32(R), C.c_int32(C_), C.c_int32(src.stride(0)),
                                      C.c_int32(dst.stride(0)), _stream()), "unimm_transpose_bf16")


def sum_slabs_bf16(slabs, out, n):
    """out[:n] (bf16) = slabs.sum(0) in slab order; slabs: fp32 [S, ...] contiguous, n a multiple of 8."""
    _dev(slabs, out)
    _check(lib().unimm_sum_slabs_bf16(_ptr(slabs), C.c_int32(slabs.shape[0]), C.c_int64(slabs.stride(0)), _ptr(out),
                                      C.c_int64(n), _stream()), "unimm_sum_slabs_bf16")


class TransposeDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("dst", C.c_void_p), ("R", C.c_int32), ("C", C.c_int32), ("ldd", C.c_int32),
                ("tile0", C.c_int32)]


def transpose_table(entries, device):
    """entries: list of (src fp32 [R, C] tensor, dst bf16 [C, ldd] tensor) -> (device table tensor, count, tiles)."""
    arr = (TransposeDesc * len(entries))()
    tiles = 0
    for d, (src, dst) in zip(arr, entries):
        _dev(src, dst)
        R, Cc = src.shape
        d.src, d.dst, d.R, d.C, d.ldd, d.tile0 = src.data_ptr(), dst.data_ptr(), R, Cc, dst.shape[1], tiles
        tiles += ((Cc + 31) // 32) * ((dst.shape[1] + 31) // 32)
    raw = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(device)
    return raw, len(entries), tiles


def transpose_cast_grouped(table, count, tiles):
    _check(lib().unimm_transpose_cast_grouped(_ptr(table), C.c_int32(count), C.c_int32(tiles), _stream()),
           "unimm_transpose_cast_grouped")


def pack_image(feat, loc, out, rows, F, ld):
    _dev(feat, loc, out)
    _check(lib().unimm_pack_image(_ptr(feat), _ptr(loc), _ptr(out), C.c_int32(rows), C.c_int32(F), C.c_int32(ld),
                                  _stream()), "unimm_pack_image")


def mul_dropout(a, b, out, n, drop=NO_DROP, fusion_sum=False):
    """out = dropout(a * b) (fusion_method 'mul') or dropout(a + b) ('sum')."""
    _dev(a, b, out)
    fn = lib().unimm_sum_dropout if fusion_sum else lib().unimm_mul_dropout
    _check(fn(_ptr(a), _ptr(b), _ptr(out), C.c_int64(n), C.c_uint32(drop[0]), C.c_uint32(drop[1]),
                                   C.c_float(drop[2]), C.c_void_p(_salt(drop)), _stream()), "unimm_mul_dropout")


def mul_dropout_bwd(a, b, dout, da, db, n, drop=NO_DROP, fusion_sum=False):
    _dev(a, b, dout, da, db)
    fn = lib().unimm_sum_dropout_bwd if fusion_sum else lib().unimm_mul_dropout_bwd
    _check(fn(_ptr(a), _ptr(b), _ptr(dout), _ptr(da), _ptr(db), C.c_int64(n),
                                       C.c_uint32(drop[0]), C.c_uint32(drop[1]), C.c_float(drop[2]), C.c_void_p(_salt(drop)),
                                       _stream()), "unimm_mul_dropout_bwd")


# ---------------------------------------------------------------------------------------------
# losses
# ---------------------------------------------------------------------------------------------
def lm_loss_fwd(logits, labels, weights, rowloss, rownll, lse, n, V, n_dev=None):
    _dev(logits, labels, weights, rowloss, rownll, lse, n_dev)
    _check(lib().unimm_lm_loss_fwd(_ptr(logits), _ptr(labels), _ptr(weights), _ptr(rowloss), _ptr(rownll), _ptr(lse),
                                   C.c_int32(n), C.c_int32(V), C.c_int32(logits.stride(0)), _ptr(n_dev), _stream()),
           "unimm_lm_loss_fwd")


def lm_loss_bwd(logits, labels, weights, lse, g, inv_denom, dlogits, n, V, n_dev=None, inv_dev=None):
    _dev(logits, labels, weights, lse, g, dlogits, n_dev, inv_dev)
    _check(lib().unimm_lm_loss_bwd(_ptr(logits), _ptr(labels), _ptr(weights), _ptr(lse), _ptr(g), C.c_float(inv_denom),
                                   _ptr(dlogits), C.c_int32(n), C.c_int32(V), C.c_int32(logits.stride(0)),
                                   C.c_int32(dlogits.stride(0)), _ptr(n_dev), _ptr(inv_dev), _stream()), "unimm_lm_loss_bwd")


def kl_loss_fwd(pred, target, label, rowloss, lse, rows, Cn):
    _dev(pred, target, label, rowloss, lse)
    _check(lib().unimm_kl_loss_fwd(_ptr(pred), _ptr(target), _ptr(label), _ptr(rowloss), _ptr(lse), C.c_int32(rows),
                                   C.c_int32(Cn), C.c_int32(pred.stride(0)), _stream()), "unimm_kl_loss_fwd")


def kl_loss_bwd(pred, target, label, lse, g, inv_denom, dpred, rows, Cn, inv_dev=None):
    _dev(pred, target, label, lse, g, dpred, inv_dev)
    _check(lib().unimm_kl_loss_bwd(_ptr(pred), _ptr(target), _ptr(label), _ptr(lse), _ptr(g), C.c_float(inv_denom),
                                   _ptr(dpred), C.c_int32(rows), C.c_int32(Cn), C.c_int32(pred.stride(0)),
                                   C.c_int32(dpred.stride(0)), _ptr(inv_dev), _stream()), "unimm_kl_loss_bwd")


def mse_loss_fwd(pred, target, label, rowloss, rows, Cn):
    _dev(pred, target, label, rowloss)
    _check(lib().unimm_mse_loss_fwd(_ptr(pred), _ptr(target), _ptr(label), _ptr(rowloss), C.c_int32(rows), C.c_int32(Cn),
                                    C.c_int32(pred.stride(0)), _stream()), "unimm_mse_loss_fwd")


def mse_loss_bwd(pred, target, label, g, inv_denom, dpred, rows, Cn, split=False):
    """dpred: bf16 [rows, ldd], or (split=True) an x-type split operand [rows, 3 ldd]."""
    _dev(pred, target, label, g, dpred)
    ldd = dpred.shape[1] // 3 if split else dpred.stride(0)
    _check(lib().unimm_mse_loss_bwd(_ptr(pred), _ptr(target), _ptr(label), _ptr(g), C.c_float(inv_denom), _ptr(dpred),
                                    C.c_int32(rows), C.c_int32(Cn), C.c_int32(pred.stride(0)), C.c_int32(ldd),
                                    C.c_int32(1 if split else 0), _stream()), "unimm_mse_loss_bwd")


def nsp_loss_fwd(logits, labels, w0, w1, loss, B):
    _dev(logits, labels, loss)
    _check(lib().unimm_nsp_loss_fwd(_ptr(logits), _ptr(labels), C.c_float(w0), C.c_float(w1), _ptr(loss), C.c_int32(B),
                                    C.c_int32(logits.stride(0)), _stream()), "unimm_nsp_loss_fwd")


def nsp_loss_bwd(logits, labels, w0, w1, g, dlogits, B, extra=None):
    """dlogits: fp32 [B, >= 2]; extra: fp32 [B, 2] contiguous gradient added in (or None)."""
    _dev(logits, labels, g, dlogits, extra)
    if dlogits.dtype != torch.float32 or (extra is not None and (extra.dtype != torch.float32 or not extra.is_contiguous())):
        raise UnimmHipError("nsp_loss_bwd: dlogits / extra must be fp32 (extra contiguous [B, 2])")
    _check(lib().unimm_nsp_loss_bwd(_ptr(logits), _ptr(labels), C.c_float(w0), C.c_float(w1), _ptr(g), _ptr(extra), _ptr(dlogits),
                                    C.c_int32(B), C.c_int32(logits.stride(0)), C.c_int32(dlogits.stride(0)), _stream()),
           "unimm_nsp_loss_bwd")


class LinearF32Args(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("bias", C.c_void_p), ("out", C.c_void_p), ("rowsum", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("sa_m", C.c_int64), ("sa_k", C.c_int64), ("sb_k", C.c_int64), ("sb_n", C.c_int64), ("ldo", C.c_int64),
                ("relu", C.c_int32), ("accumulate", C.c_int32)]


def linear_f32(a, b, out, M, N, K, sa, sb, bias=None, relu=False, accumulate=False, rowsum=None):
    """out[M, N] (+)= act(A . B + bias), all fp32; sa = (stride of A over m, over k), sb = (stride of B over k, over n),
    in elements (include/unimm_hip.h: unimm_linear_f32)."""
    _dev(a, b, out, bias, rowsum)
    for t in (a, b, out, bias, rowsum):
        if t is not None and t.dtype != torch.float32:
            raise UnimmHipError("linear_f32: every operand is fp32")
    g = LinearF32Args()
    g.a, g.b, g.bias, g.out, g.rowsum = a.data_ptr(), b.data_ptr(), _P(bias), out.data_ptr(), _P(rowsum)
    g.M, g.N, g.K = M, N, K
    g.sa_m, g.sa_k, g.sb_k, g.sb_n, g.ldo = sa[0], sa[1], sb[0], sb[1], out.stride(0)
    g.relu, g.accumulate = int(bool(relu)), int(bool(accumulate))
    _check(lib().unimm_linear_f32(C.byref(g), _stream()), "unimm_linear_f32")
    return out


def rows_add_f32(dst, idx, src, n, H):
    """dst (bf16 [*, H] contiguous rows)[idx[r], :] += src (fp32 [n, H] contiguous)[r, :]"""
    _dev(dst, idx, src)
    _check(lib().unimm_rows_add_f32(_ptr(dst), _ptr(idx), _ptr(src), C.c_int32(n), C.c_int32(H), _stream()), "unimm_rows_add_f32")


def reduce_sum(src, n, dst, scale=1.0, n_dev=None, scale_dev=None):
    _dev(src, dst, n_dev, scale_dev)
    _check(lib().unimm_reduce_sum(_ptr(src), C.c_int64(n), _ptr(dst), C.c_float(scale), _ptr(n_dev), _ptr(scale_dev), _stream()),
           "unimm_reduce_sum")


def segment_sum(src, seg, dst, n, sign=1.0):
    _dev(src, seg, dst)
    _check(lib().unimm_segment_sum(_ptr(src), _ptr(seg), _ptr(dst), C.c_int64(n), C.c_float(sign), _stream()),
           "unimm_segment_sum")


def gelu_bwd(dt, u, du, n):
    _dev(dt, u, du)
    _check(lib().unimm_gelu_bwd(_ptr(dt), _ptr(u), _ptr(du), C.c_int64(n), _stream()), "unimm_gelu_bwd")


def gather_rows(src, idx, dst, n, H, scatter=False, n_dev=None):
    _dev(src, idx, dst, n_dev)
    _check(lib().unimm_gather_rows(_ptr(src), _ptr(idx), _ptr(dst), C.c_int32(n), C.c_int32(H),
                                   C.c_int32(1 if scatter else 0), _ptr(n_dev), _stream()), "unimm_gather_rows")


_EPI_NAMES = ["BIAS", "BIAS_GELU", "BIAS_DROP_RESID", "BIAS_RELU", "DGELU", "ADD", "MUL", "BIAS_GELU_DG"]
_TILE_NAMES = {1: "128x128", 3: "256x256", 6: "192x256", 7: "64x128", 8: "256x256pp", 9: "64x128s3", 10: "128x128s3", 12: "192x256x3", 14: "128x128x3", 15: "64x128x3"}
PROF_VARIANTS = 516


def gemm_variant_name(i):
    """Name of a profiler variant = one kernel symbol (csrc/gemm.hip: PROF_VARIANTS): the persistent form is gemm_ntp."""
    if i >= 512:
        return {512: "gemm_tn_pp", 513: "gemm_tn<256x256 lock-step>", 514: "gemm_tn<128x128>"}.get(i, f"variant{i}")
    f32, epi, tile, persist = i & 1, (i >> 1) & 7, (i >> 4) & 15, i >> 8
    return f"gemm_nt{'p' if persist else ''}<{_TILE_NAMES.get(tile, tile)},{_EPI_NAMES[epi]},{'f32' if f32 else 'bf16'}>"


ADAMW_MAX_GROUPS = 8
ADAMW_SKIP = 255          # chunk group id of parameters that never receive a gradient


class AdamWArgs(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("w16", C.c_void_p),
                ("group", C.c_void_p), ("n", C.c_int64), ("n_groups", C.c_int32), ("step", C.c_int32),
                ("lr", C.c_double * ADAMW_MAX_GROUPS), ("weight_decay", C.c_double * ADAMW_MAX_GROUPS),
                ("beta1", C.c_double), ("beta2", C.c_double), ("eps", C.c_double), ("grad_scale", C.c_float),
                ("correct_bias", C.c_int32), ("zero_grad", C.c_int32)]


def adamw_step(p, g, m, v, group, lrs, wds, step, beta1=0.9, beta2=0.999, eps=1e-6, w16=None, grad_scale=1.0,
               correct_bias=True, zero_grad=False):
    """One fused AdamW step over flat fp32 arenas (see include/unimm_hip.h: unimm_adamw_step)."""
    _dev(p, g, m, v, group, w16)
    if len(lrs) != len(wds) or not 1 <= len(lrs) <= ADAMW_MAX_GROUPS:
        raise UnimmHipError(f"adamw_step: 1..{ADAMW_MAX_GROUPS} (lr, weight_decay) groups, got {len(lrs)}")
    a = AdamWArgs()
    a.p, a.g, a.m, a.v, a.w16, a.group = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), \
        (w16.data_ptr() if w16 is not None else None), group.data_ptr()
    a.n, a.n_groups, a.step = p.numel(), len(lrs), int(step)
    for i, (lr, wd) in enumerate(zip(lrs, wds)):
        a.lr[i], a.weight_decay[i] = float(lr), float(wd)
    a.beta1, a.beta2, a.eps, a.grad_scale = beta1, beta2, eps, grad_scale
    a.correct_bias, a.zero_grad = int(bool(correct_bias)), int(bool(zero_grad))
    _check(lib().unimm_adamw_step(C.byref(a), _stream()), "unimm_adamw_step")


NDCG_MAX_OPTIONS, NDCG_MAX_ITER = 128, 64


class NdcgArgs(C.Structure):
    _fields_ = [("pred", C.c_void_p), ("truth", C.c_void_p), ("ndcg", C.c_void_p), ("alive", C.c_void_p), ("dpred", C.c_void_p),
                ("iters", C.c_void_p), ("slates", C.c_int32), ("n", C.c_int32), ("k", C.c_int32),
                ("powered_relevancies", C.c_int32), ("max_iter", C.c_int32), ("pad_label", C.c_float),
                ("temperature", C.c_float), ("tol", C.c_float)]


def neural_ndcg(pred, truth, pad_label=-1.0, temperature=1.0, powered=True, k=None, max_iter=50, tol=1e-6):
    """pred, truth fp32 [slates, n] on the device -> (ndcg [slates], alive [slates], dpred [slates, n], iters [slates])
    (see include/unimm_hip.h: unimm_neural_ndcg)."""
    _dev(pred, truth)
    if pred.dtype != torch.float32 or truth.dtype != torch.float32 or pred.shape != truth.shape or pred.dim() != 2:
        raise UnimmHipError("neural_ndcg: pred and truth must be fp32 [slates, n] of the same shape")
    pred, truth = pred.contiguous(), truth.contiguous()
    S, n = pred.shape
    if not 1 <= n <= NDCG_MAX_OPTIONS or not 1 <= max_iter <= NDCG_MAX_ITER:
        raise UnimmHipError(f"neural_ndcg: n <= {NDCG_MAX_OPTIONS} options and max_iter <= {NDCG_MAX_ITER} (got {n}, {max_iter})")
    ndcg = torch.empty(S, dtype=torch.float32, device=pred.device)
    alive = torch.empty_like(ndcg)
    dpred = torch.empty_like(pred)
    iters = torch.empty(S, dtype=torch.int32, device=pred.device)
    a = NdcgArgs()
    a.pred, a.truth, a.ndcg, a.alive, a.dpred, a.iters = (t.data_ptr() for t in (pred, truth, ndcg, alive, dpred, iters))
    a.slates, a.n, a.k, a.powered_relevancies, a.max_iter = S, n, (0 if k is None else int(k)), int(bool(powered)), int(max_iter)
    a.pad_label, a.temperature, a.tol = float(pad_label), float(temperature), float(tol)
    _check(lib().unimm_neural_ndcg(C.byref(a), _stream()), "unimm_neural_ndcg")
    return ndcg, alive, dpred, iters


def prof_enable(on):
    """False / 0 = off, True / 1 = every GEMM launch, 2 = weight-gradient launches only."""
    _check(lib().unimm_prof_enable(C.c_int32(int(on))), "unimm_prof_enable")


def prof_collect():
    """-> {variant name: (total_ms, total_flops, launches)} for the launches since prof_enable(True)."""
    n = PROF_VARIANTS
    ms, fl, cnt = (C.c_double * n)(), (C.c_double * n)(), (C.c_int32 * n)()
    _check(lib().unimm_prof_collect(ms, fl, cnt, C.c_int32(n)), "unimm_prof_collect")
    return {gemm_variant_name(i): (ms[i], fl[i], cnt[i]) for i in range(n) if cnt[i] > 0}


def prof_tag(tag):
    """Tag (0 .. 7) of the GEMM launches that follow (see include/unimm_hip.h: unimm_prof_tag)."""
    lib().unimm_prof_tag(C.c_int32(int(tag)))


def prof_tagged(ntags=8):
    """-> {tag: (total_ms, total_flops, launches, union_ms)} of the NT launches the last prof_collect() consumed; union_ms = the
    wall time during which at least one launch of the tag was executing (launches of two streams side by side count once)."""
    ms, fl, cnt, un = (C.c_double * ntags)(), (C.c_double * ntags)(), (C.c_int32 * ntags)(), (C.c_double * ntags)()
    _check(lib().unimm_prof_tagged(ms, fl, cnt, un, C.c_int32(ntags)), "unimm_prof_tagged")
    return {t: (ms[t], fl[t], cnt[t], un[t]) for t in range(ntags)}


# ---------------------------------------------------------------------------------------------
# fp32-accuracy mode (csrc/x3ops.hip; include/unimm_hip.h "fp32x3")
# ---------------------------------------------------------------------------------------------
X3_COPY, X3_ADD, X3_GELU, X3_MUL_DGELU = range(4)


class X3SplitArgs(C.Structure):
    _fields_ = [("a", C.c_void_p), ("b", C.c_void_p), ("out32", C.c_void_p), ("out3", C.c_void_p), ("rows", C.c_int64),
                ("cols", C.c_int32), ("cp", C.c_int32), ("lda", C.c_int32), ("ldb", C.c_int32), ("ld32", C.c_int32),
                ("op", C.c_int32), ("wtype", C.c_int32)]


def x3_split(a, out3=None, out32=None, op=X3_COPY, b=None, rows=None, cols=None, cp=None, wtype=False):
    """y = op(a, b) (fp32 [rows, cols]) -> out32 (fp32) and / or out3 (split operand, bf16 [rows, 3 cp])."""
    _dev(a, b, out32, out3)
    st = getattr(_tls, "x3s", None)
    if st is None:
        s = X3SplitArgs()
        st = _tls.x3s = (s, C.addressof(s), lib().unimm_x3_split)
    s, addr, fn = st
    s.a, s.b, s.out32, s.out3 = a.data_ptr(), _P(b), _P(out32), _P(out3)
    s.rows = a.shape[0] if rows is None else rows
    s.cols = a.shape[1] if cols is None else cols
    s.cp = (out3.shape[1] // 3) if cp is None else cp
    s.lda, s.ldb, s.ld32 = a.stride(0), (b.stride(0) if b is not None else 0), (out32.stride(0) if out32 is not None else 0)
    s.op, s.wtype = op, 1 if wtype else 0
    rc = fn(addr, _stream())
    if rc != 0:
        _check(rc, "unimm_x3_split")


def x3_split_wt(src, dst, R, C_, Rp):
    """transposed w-type split: src fp32 [R, C] -> dst bf16 [C, 3 Rp] (zero-filled by the caller once)."""
    _dev(src, dst)
    _check(lib().unimm_x3_split_wt(_ptr(src), _ptr(dst), C.c_int32(R), C.c_int32(C_), C.c_int32(src.stride(0)), C.c_int32(Rp),
                                   _stream()), "unimm_x3_split_wt")


def x3_layernorm_bwd_partials(dy, x, mean, rstd, gamma, dx32, dxd3, partials, M, H, drop=None, out_drop=None, m_dev=None):
    drop = drop or NO_DROP
    out_drop = out_drop or NO_DROP
    _dev(dy, x, mean, rstd, gamma, dx32, dxd3, partials)
    blocks = C.c_int32(0)
    _check(lib().unimm_x3_layernorm_bwd_partials(
        _ptr(dy), _ptr(x), _ptr(mean), _ptr(rstd), _ptr(gamma), _ptr(dx32), _ptr(dxd3), _ptr(partials), C.c_int32(M), C.c_int32(H),
        C.c_uint32(drop[0]), C.c_uint32(drop[1]), C.c_float(drop[2]), C.c_uint32(out_drop[0]), C.c_uint32(out_drop[1]),
        C.c_float(out_drop[2]), C.byref(blocks), _ptr(m_dev), C.c_void_p(_salt(drop, out_drop)), _stream()),
        "unimm_x3_layernorm_bwd_partials")
    return blocks.value


def embed_bwd_f32(ids, pos, typ, word, post, type_, ext, gamma, beta, dy, dword, dpos, dtype, dext, dgamma, dbeta,
                  partials, M, H, type_vocab=2, eps=1e-12, drop=NO_DROP, m_dev=None, rows=None):
    a = _embed_args(ids, pos, typ, word, post, type_, ext, gamma, beta, M, H, type_vocab, eps, drop, m_dev, rows)
    _dev(dy, dword, dpos, dtype, dext, dgamma, dbeta, partials)
    _check(lib().unimm_embed_bwd_f32(C.byref(a), _ptr(dy), _ptr(dword), _ptr(dpos), _ptr(dtype), _ptr(dext), _ptr(dgamma),
                                     _ptr(dbeta), _ptr(partials), _stream()), "unimm_embed_bwd_f32")


def x3_lm_loss_bwd(logits, labels, weights, lse, g, inv_denom, out3, n, V, n_dev=None, inv_dev=None):
    _dev(logits, labels, weights, lse, g, out3)
    _check(lib().unimm_x3_lm_loss_bwd(_ptr(logits), _ptr(labels), _ptr(weights), _ptr(lse), _ptr(g), C.c_float(inv_denom),
                                      _ptr(out3), C.c_int32(n), C.c_int32(V), C.c_int32(logits.stride(0)),
                                      C.c_int32(out3.shape[1] // 3), _ptr(n_dev), _ptr(inv_dev), _stream()), "unimm_x3_lm_loss_bwd")


def x3_kl_loss_bwd(pred, target, label, lse, g, inv_denom, out3, rows, Cn, inv_dev=None):
    _dev(pred, target, label, lse, g, out3)
    _check(lib().unimm_x3_kl_loss_bwd(_ptr(pred), _ptr(target), _ptr(label), _ptr(lse), _ptr(g), C.c_float(inv_denom),
                                      _ptr(out3), C.c_int32(rows), C.c_int32(Cn), C.c_int32(pred.stride(0)),
                                      C.c_int32(out3.shape[1] // 3), _ptr(inv_dev), _stream()), "unimm_x3_kl_loss_bwd")


def x3_rows_add(dst, idx, src, n, H):
    _dev(dst, idx, src)
    _check(lib().unimm_x3_rows_add(_ptr(dst), _ptr(idx), _ptr(src), C.c_int32(n), C.c_int32(H), C.c_int32(dst.stride(0)),
                                   _stream()), "unimm_x3_rows_add")


class X3AttnPlanes(C.Structure):
    _fields_ = [("out3", C.c_void_p), ("dq3", C.c_void_p), ("dk3", C.c_void_p), ("dv3", C.c_void_p), ("ld3", C.c_int32), ("cp3", C.c_int32)]


def x3_attn_fwd(q, k, v, out, lse, mask, B, H, Tq, Tk, D, scale, mask_q_stride, mask_b_stride, drop=NO_DROP, qvar=None, kvar=None,
                out3=None):
    """unimm_attn_fwd on fp32 q / k / v / out (2-D views, row stride = stride(0)).  out3 (bf16 [rows, 3 cp]): the context also
    as an x-type split operand."""
    _dev(q, k, v, out, lse, mask, out3)
    pl = None
    if out3 is not None:
        pl = X3AttnPlanes()
        pl.out3, pl.ld3, pl.cp3 = out3.data_ptr(), out3.stride(0), out3.shape[1] // 3
    a = AttnArgs()
    a.q, a.k, a.v, a.out, a.lse, a.mask = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), _P(lse), mask.data_ptr()
    a.q_off, a.q_len = (qvar[0].data_ptr(), qvar[1].data_ptr()) if qvar is not None else (None, None)
    a.k_off, a.k_len = (kvar[0].data_ptr(), kvar[1].data_ptr()) if kvar is not None else (None, None)
    a.order = _item_order(qvar, kvar)
    a.B, a.H, a.Tq, a.Tk, a.D = B, H, Tq, Tk, D
    a.ldq, a.ldk, a.ldv, a.ldo = q.stride(0), k.stride(0), v.stride(0), out.stride(0)
    a.mask_q_stride, a.mask_b_stride, a.scale = mask_q_stride, mask_b_stride, scale
    a.drop_key, a.drop_thr, a.drop_scale = drop[:3]
    a.drop_salt = _salt(drop)
    _check(lib().unimm_x3_attn_fwd(C.byref(a), C.byref(pl) if pl is not None else None, _stream()), "unimm_x3_attn_fwd")


def x3_attn_bwd(q, k, v, out, dout, lse, delta, dq, dk, dv, mask, B, H, Tq, Tk, D, scale, mask_q_stride, mask_b_stride,
                drop=NO_DROP, qvar=None, kvar=None, planes=None):
    """planes = (cp3, ld3): dq / dk / dv are bf16 views of plane 0 of x-type split buffers (plane stride cp3, row stride ld3) and
    receive the gradients as split operands instead of fp32."""
    _dev(q, k, v, out, dout, lse, delta, dq, dk, dv, mask)
    a = AttnBwdArgs()
    pl = None
    if planes is not None:
        pl = X3AttnPlanes()
        pl.dq3, pl.dk3, pl.dv3, pl.cp3, pl.ld3 = dq.data_ptr(), dk.data_ptr(), dv.data_ptr(), planes[0], planes[1]
    a.q_off, a.q_len = (qvar[0].data_ptr(), qvar[1].data_ptr()) if qvar is not None else (None, None)
    a.k_off, a.k_len = (kvar[0].data_ptr(), kvar[1].data_ptr()) if kvar is not None else (None, None)
    a.order = _item_order(qvar, kvar)
    a.q, a.k, a.v, a.out, a.dout = q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(), dout.data_ptr()
    a.lse, a.delta, a.mask = lse.data_ptr(), delta.data_ptr(), mask.data_ptr()
    if pl is None:
        a.dq, a.dk, a.dv = dq.data_ptr(), dk.data_ptr(), dv.data_ptr()
        a.lddq, a.lddk, a.lddv = dq.stride(0), dk.stride(0), dv.stride(0)
    a.B, a.H, a.Tq, a.Tk, a.D = B, H, Tq, Tk, D
    a.ldq, a.ldk, a.ldv, a.ldo, a.lddo = q.stride(0), k.stride(0), v.stride(0), out.stride(0), dout.stride(0)
    a.mask_q_stride, a.mask_b_stride, a.scale = mask_q_stride, mask_b_stride, scale
    a.drop_key, a.drop_thr, a.drop_scale = drop[:3]
    a.drop_salt = _salt(drop)
    _check(lib().unimm_x3_attn_bwd(C.byref(a), C.byref(pl) if pl is not None else None, _stream()), "unimm_x3_attn_bwd")


def x3_attn_set_impl(impl):
    """1 = fp32 matrix-instruction attention kernels (default), 0 = the vector-ALU kernels (A/B runs)."""
    _check(lib().unimm_x3_attn_set_impl(int(impl)), "unimm_x3_attn_set_impl")


def x3_layernorm_fwd(x, gamma, beta, y32, y3, mean, rstd, M, H, eps=1e-12, drop=NO_DROP):
    """LayerNorm forward writing fp32 rows (or None) and the x-type split operand y3 [M, 3 H]."""
    _dev(x, gamma, beta, y32, y3, mean, rstd)
    _check(lib().unimm_x3_layernorm_fwd(x.data_ptr(), gamma.data_ptr(), beta.data_ptr(), _P(y32), y3.data_ptr(), _P(mean), _P(rstd),
                                        M, H, eps, drop[0], drop[1], drop[2], _salt(drop), _stream()), "unimm_x3_layernorm_fwd")
