"""Isolated timing of the long-K GEMMs of a 30-sequence step with and without split-K:  python tools/exp/splitk_time.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 3900
g = torch.Generator(device="cuda").manual_seed(0)
ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
def timeit(fn, iters=200, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
for (N, K, epi) in [(768, 3072, lib.EPI_ADD), (768, 2304, lib.EPI_ADD), (768, 3072, lib.EPI_BIAS_DROP_RESID), (1024, 3072, lib.EPI_ADD),
                    (768, 768, lib.EPI_ADD), (3072, 768, lib.EPI_MUL), (2304, 768, lib.EPI_BIAS)]:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    resid = epi == lib.EPI_BIAS_DROP_RESID
    o = torch.empty((M, N), device="cuda") if resid else torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ax = torch.randn((M, N), device="cuda") if resid else torch.randn((M, N), device="cuda").to(torch.bfloat16)
    res = []
    for tile, sks in ((7, (0, 2)), (1, (0, 2)), (9, (0, 2)), (10, (0, 2)), (14, (0, 2)), (15, (0, 2))):
        for sk in sks:
            t = timeit(lambda: lib.gemm_nt(x, w, o, bias=b, epilogue=epi, aux=ax, tile=tile, splitk=sk, splitk_ws=ws))
            res.append(f"t{tile}/s{sk} {t:5.1f}")
    print(f"M={M} N={N:5d} K={K:5d} epi={epi}: " + "  ".join(res))
