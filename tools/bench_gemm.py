"""Micro-benchmark of the GEMM kernels at the hot-path shapes (random data, HIP events)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import lib

def timeit(fn, iters=100, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

M = int(sys.argv[1]) if len(sys.argv) > 1 else 61440
CFGS = [int(c) for c in (sys.argv[2].split(",") if len(sys.argv) > 2 else ["1", "2"])]
g = torch.Generator(device="cuda").manual_seed(0)
print(f"M={M}")
for (N, K, epi) in [(2304, 768, lib.EPI_BIAS), (768, 768, lib.EPI_BIAS_DROP_RESID), (3072, 768, lib.EPI_BIAS_GELU_DG),
                    (768, 3072, lib.EPI_BIAS_DROP_RESID), (3072, 768, lib.EPI_MUL), (768, 2304, lib.EPI_ADD),
                    (768, 3072, lib.EPI_ADD), (768, 768, lib.EPI_ADD), (768, 1024, lib.EPI_BIAS)]:
    x = (torch.randn((M, K), generator=g, device="cuda")).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    aux = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    out2 = torch.empty_like(out)
    fl = 2.0 * M * N * K
    resid = epi == lib.EPI_BIAS_DROP_RESID
    o = torch.empty((M, N), device="cuda") if resid else out
    ax = torch.randn((M, N), device="cuda") if resid else aux
    for cfg in CFGS:
        t = timeit(lambda: lib.gemm_nt(x, w, o, bias=b, epilogue=epi, aux=ax, out2=out2 if epi == lib.EPI_BIAS_GELU_DG else None, tile=cfg))
        print(f"NT  N={N:5d} K={K:5d} epi={epi} cfg={cfg}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s")
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
    dy = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    dw = torch.zeros((N, K), device="cuda")
    t = timeit(lambda: lib.gemm_tn(dy, x, dw))
    fl = 2.0 * M * N * K
    db = torch.zeros(N, device="cuda")
    tb = timeit(lambda: lib.gemm_tn(dy, x, dw, dbias=db))
    print(f"TN  N={N:5d} K={K:5d}: {t*1e6:8.1f} us  {fl/t/1e12:7.1f} TFLOP/s   with dbias: {tb*1e6:8.1f} us {fl/tb/1e12:7.1f} TFLOP/s")
    if hasattr(lib.lib(), "unimm_gemm_tn_grouped_ws"):
        ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
        tw = timeit(lambda: lib.gemm_tn_grouped([(dy, x, dw, None, None, None, db)], shared=False, ws=ws))
        print(f"    slab reducer (workspace): {tw*1e6:8.1f} us  {fl/tw/1e12:7.1f} TFLOP/s")
