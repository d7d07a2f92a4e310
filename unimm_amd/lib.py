"""ctypes binding of libunimm_hip.so (the C ABI in include/unimm_hip.h).

There is no CPU fallback: if the library is missing or a kernel call fails, this raises.
PyTorch is used only for device memory and streams; every compute call below goes straight to a
hand-written HIP kernel on torch's current stream."""
from __future__ import annotations

import ctypes as C
import os

import torch

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libunimm_hip.so")
ABI_VERSION = 1

EPI_BIAS, EPI_BIAS_GELU, EPI_BIAS_DROP_RESID, EPI_BIAS_RELU, EPI_DGELU, EPI_ADD = range(6)

_ERR = {-1: "UNIMM_E_ARG", -2: "UNIMM_E_SHAPE", -3: "UNIMM_E_ALIGN", -4: "UNIMM_E_HIP"}


class UnimmHipError(RuntimeError):
    pass


class GemmNtArgs(C.Structure):
    _fields_ = [("x", C.c_void_p), ("w", C.c_void_p), ("bias", C.c_void_p), ("aux", C.c_void_p),
                ("out", C.c_void_p), ("out2", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("ldx", C.c_int32), ("ldw", C.c_int32), ("ldaux", C.c_int32), ("ldo", C.c_int32),
                ("epilogue", C.c_int32), ("out_f32", C.c_int32),
                ("drop_key", C.c_uint32), ("drop_thr", C.c_uint32), ("drop_scale", C.c_float)]


class GemmTnArgs(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x", C.c_void_p), ("dw", C.c_void_p),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32),
                ("lddy", C.c_int32), ("ldx", C.c_int32), ("lddw", C.c_int32)]


_lib = None


def lib():
    """Load the shared library once; fail loudly when it is absent (no fallback path exists)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise UnimmHipError(
            f"{LIB_PATH} is missing: build it with `python -m unimm_amd.build` (hipcc, gfx950). "
            "unimm_amd has no CPU or PyTorch fallback for its kernels.")
    L = C.CDLL(LIB_PATH)
    L.unimm_version.restype = C.c_int
    L.unimm_arch.restype = C.c_char_p
    if L.unimm_version() != ABI_VERSION:
        raise UnimmHipError(f"libunimm_hip.so ABI {L.unimm_version()} != expected {ABI_VERSION}: rebuild")
    for name in SYMBOLS:
        fn = getattr(L, name)
        if name not in ("unimm_arch",):
            fn.restype = C.c_int
    _lib = L
    return L


# every symbol include/unimm_hip.h declares (tests check the .so exports each of them)
SYMBOLS = ["unimm_version", "unimm_arch", "unimm_gemm_nt", "unimm_gemm_tn"]


def _check(rc, what):
    if rc != 0:
        raise UnimmHipError(f"{what} failed: {_ERR.get(rc, rc)}")


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _dev(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise UnimmHipError("unimm_amd kernels need device tensors (got a CPU tensor)")


def gemm_nt(x, w, out, bias=None, epilogue=EPI_BIAS, aux=None, out2=None, drop=None, M=None, N=None, K=None):
    """out[M,N] = epi(x[M,K] @ w[N,K]^T).  x/w bf16 2-D (row stride = stride(0)); out bf16 or fp32."""
    _dev(x, w, out, bias, aux, out2)
    a = GemmNtArgs()
    a.x, a.w, a.bias, a.aux, a.out, a.out2 = _ptr(x), _ptr(w), _ptr(bias), _ptr(aux), _ptr(out), _ptr(out2)
    a.M = x.shape[0] if M is None else M
    a.N = w.shape[0] if N is None else N
    a.K = x.shape[1] if K is None else K
    a.ldx, a.ldw, a.ldo = x.stride(0), w.stride(0), out.stride(0)
    a.ldaux = aux.stride(0) if aux is not None else 0
    a.epilogue = epilogue
    a.out_f32 = 1 if out.dtype == torch.float32 else 0
    if drop is not None:
        a.drop_key, a.drop_thr, a.drop_scale = drop
    _check(lib().unimm_gemm_nt(C.byref(a), _stream()), "unimm_gemm_nt")
    return out


def gemm_tn(dy, x, dw, M=None, N=None, K=None):
    """dw[N,K] += dy[M,N]^T @ x[M,K] (fp32 atomics)."""
    _dev(dy, x, dw)
    a = GemmTnArgs()
    a.dy, a.x, a.dw = _ptr(dy), _ptr(x), _ptr(dw)
    a.M = dy.shape[0] if M is None else M
    a.N = dy.shape[1] if N is None else N
    a.K = x.shape[1] if K is None else K
    a.lddy, a.ldx, a.lddw = dy.stride(0), x.stride(0), dw.stride(0)
    _check(lib().unimm_gemm_tn(C.byref(a), _stream()), "unimm_gemm_tn")
    return dw
