"""The fp32-residual GEMM epilogue (attention output / FFN down: N = 768) on ROTATING buffer sets, so that the residual operand
comes from HBM as in the step (the one-buffer microbenchmark keeps it in the Infinity Cache).  Base library against the
-DUNIMM_AUX_TOUCH variant (touch the tile's residual lines behind the prologue of the main loop).
    python tools/exp/aux_touch_ab.py      (UNIMM_HIP_LIB=... for the variant)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib, dropout as DR
M, N, SETS = 31162, 768, 5
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, iters=60, warm=6):
    for i in range(warm):
        fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print(f"library: {os.environ.get('UNIMM_HIP_LIB', 'unimm_amd/libunimm_hip.so')}")
for K in (768, 1024, 3072):
    xs = [torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16) for _ in range(SETS)]
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    outs = [torch.empty((M, N), device="cuda") for _ in range(SETS)]
    ress = [torch.randn((M, N), device="cuda") for _ in range(SETS)]
    mean, rstd = torch.randn(M, device="cuda"), torch.rand(M, device="cuda") + 0.5
    gam, bet = torch.randn(N, device="cuda"), torch.randn(N, device="cuda")
    drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
    fn = lambda i: lib.gemm_nt(xs[i % SETS], w, outs[i % SETS], bias=b, epilogue=lib.EPI_BIAS_DROP_RESID, aux=ress[i % SETS], drop=drop,
                               aux_ln=(mean, rstd, gam, bet))
    plain = lambda i: lib.gemm_nt(xs[i % SETS], w, outs[i % SETS], bias=b)
    t = min(timeit(fn), timeit(fn))
    tp = min(timeit(plain), timeit(plain))
    print(f"K={K:5d}: bias + dropout + lazy-LN fp32 residual -> fp32 {t:7.1f} us ({2.0*M*N*K/t/1e6:6.0f} TF/s); bias -> fp32 {tp:7.1f} us")
    del xs, outs, ress
