"""Epilogue cost of the hot GEMM shapes at M = 31,162 (what each fused epilogue adds to the plain bias -> bf16 GEMM of its shape)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
M = 31162
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=100, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (N, K) in [(3072, 768), (768, 3072), (2304, 768), (768, 2304), (768, 768), (3072, 1024)]:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    o16b = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    aux16 = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    rows = [("bias -> bf16", lambda: lib.gemm_nt(x, w, o16, bias=b)),
            ("no bias -> bf16", lambda: lib.gemm_nt(x, w, o16)),
            ("bias + GELU -> bf16", lambda: lib.gemm_nt(x, w, o16, bias=b, epilogue=lib.EPI_BIAS_GELU)),
            ("bias + GELU, GELU' -> 2 x bf16", lambda: lib.gemm_nt(x, w, o16, bias=b, epilogue=lib.EPI_BIAS_GELU_DG, out2=o16b)),
            ("x aux (bf16) -> bf16 [MUL]", lambda: lib.gemm_nt(x, w, o16, epilogue=lib.EPI_MUL, aux=aux16)),
            ("+ aux (bf16) -> bf16 [ADD]", lambda: lib.gemm_nt(x, w, o16, epilogue=lib.EPI_ADD, aux=aux16))]
    timeit(rows[0][1], iters=200)               # the first launches on fresh buffers are slow: not a property of the first row
    for name, fn in rows:
        t = timeit(fn)
        print(f"N={N:5d} K={K:5d} {name:34s} {t:7.1f} us  {2.0*M*N*K/t/1e6:7.1f} TFLOP/s")
