"""The residual GEMM + LayerNorm of a sublayer as two launches and as one (unimm_gemm_nt_args.ln_*), at the row counts of a
30-sequence step: us per sublayer, alone on the chip (in the step the other stream's kernels run beside it)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
BF = torch.bfloat16


def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M, N, K, tile in ((3900, 768, 768, 15), (3900, 768, 3072, 15), (3900, 768, 1024, 15), (1110, 1024, 1024, 14), (1110, 1024, 768, 14),
                      (7800, 768, 768, 15), (7800, 768, 3072, 15), (2220, 1024, 1024, 14)):
    x = torch.randn((M, K), device="cuda").to(BF); w = (torch.randn((N, K), device="cuda") * 0.05).to(BF)
    bias = torch.randn(N, device="cuda"); aux = torch.randn((M, N), device="cuda")
    g, b = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    pre = torch.empty((M, N), device="cuda"); y = torch.empty((M, N), dtype=BF, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    tickets = torch.zeros(1024, dtype=torch.int32, device="cuda")
    kw = dict(bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=aux, tile=tile)
    def two():
        lib.gemm_nt(x, w, pre, **kw)
        lib.layernorm_fwd(pre, g, b, None, y, mean, rstd, M, N)
    t_g = timeit(lambda: lib.gemm_nt(x, w, pre, **kw))
    t_two = timeit(two)
    t_one = timeit(lambda: lib.gemm_nt(x, w, pre, ln=(g, b, y, mean, rstd, tickets, 1e-12), **kw))
    print(f"M={M:5d} N={N:4d} K={K:4d} tile {tile:2d}: GEMM {t_g:6.1f} us   GEMM + LayerNorm launch {t_two:6.1f} us   one launch {t_one:6.1f} us")
