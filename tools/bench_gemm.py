"""Micro-benchmark of the GEMM kernels at the hot-path shapes (random data, HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import lib

def timeit(fn, iters=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

M = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
g = torch.Generator(device="cuda").manual_seed(0)
print(f"M={M}")
for (N, K, epi) in [(2304, 768, lib.EPI_BIAS), (768, 768, lib.EPI_BIAS_DROP_RESID), (3072, 768, lib.EPI_BIAS_GELU),
                    (768, 3072, lib.EPI_BIAS_DROP_RESID), (3072, 768, lib.EPI_BIAS), (768, 1024, lib.EPI_BIAS)]:
    x = (torch.randn((M, K), generator=g, device="cuda")).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    aux = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    out2 = torch.empty_like(out)
    t = timeit(lambda: lib.gemm_nt(x, w, out, bias=b, epilogue=epi, aux=aux, out2=out2 if epi == lib.EPI_BIAS_GELU else None))
    fl = 2.0 * M * N * K
    by = 2.0 * (M * K + N * K + M * N * (2 if epi in (lib.EPI_BIAS_GELU, lib.EPI_BIAS_DROP_RESID) else 1))
    print(f"NT  N={N:5d} K={K:5d} epi={epi}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s  {by/t/1e12:5.2f} TB/s")
    tt = timeit(lambda: torch.matmul(x, w.t()))
    print(f"    torch(hipBLASLt) matmul          : {tt*1e6:8.1f} us  {fl/tt/1e12:7.1f} TFLOP/s")
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    dy = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    dw = torch.zeros((N, K), device="cuda")
    t = timeit(lambda: lib.gemm_tn(dy, x, dw))
    fl = 2.0 * M * N * K
    print(f"TN  N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s")
    tt = timeit(lambda: torch.matmul(dy.t(), x))
    print(f"    torch matmul         : {tt*1e6:8.1f} us  {fl/tt/1e12:7.1f} TFLOP/s")
