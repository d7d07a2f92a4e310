"""What the residual epilogue's pieces cost (N = 768, M = 31,162):  python tools/exp/resid_epilogue_cost.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib, dropout as DR
M, N = 31162, 768
g = torch.Generator(device="cuda").manual_seed(0)
def timeit(fn, iters=100, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for K in (768, 3072):
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    o32 = torch.empty((M, N), device="cuda")
    o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    res = torch.randn((M, N), device="cuda")
    aux16 = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    mean, rstd = torch.randn(M, device="cuda"), torch.rand(M, device="cuda") + 0.5
    gam, bet = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
    ln = (mean, rstd, gam, bet)
    rows = [("bias -> bf16", lambda: lib.gemm_nt(x, w, o16, bias=b)),
            ("bias -> fp32", lambda: lib.gemm_nt(x, w, o32, bias=b)),
            ("+ aux (bf16) -> bf16 [ADD]", lambda: lib.gemm_nt(x, w, o16, epilogue=lib.EPI_ADD, aux=aux16)),
            ("bias + fp32 residual -> fp32", lambda: lib.gemm_nt(x, w, o32, bias=b, epilogue=lib.EPI_BIAS_DROP_RESID, aux=res)),
            ("  ... + dropout", lambda: lib.gemm_nt(x, w, o32, bias=b, epilogue=lib.EPI_BIAS_DROP_RESID, aux=res, drop=drop)),
            ("  ... + lazy LayerNorm of the residual", lambda: lib.gemm_nt(x, w, o32, bias=b, epilogue=lib.EPI_BIAS_DROP_RESID, aux=res, aux_ln=ln)),
            ("  ... + both (the engine's call)", lambda: lib.gemm_nt(x, w, o32, bias=b, epilogue=lib.EPI_BIAS_DROP_RESID, aux=res, drop=drop, aux_ln=ln))]
    for name, fn in rows:
        t = timeit(fn)
        print(f"K={K:5d} {name:45s} {t:7.1f} us  {2.0*M*N*K/t/1e6:7.1f} TFLOP/s")
