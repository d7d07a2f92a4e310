"""The fp32x3 engine's GEMM shapes at a small per-rank batch (configs[3] on 8 GPUs: 12-13 sequences, ~1.7k text rows): fp32 outputs,
K = 3 planes of 768 ... 3072; tiles x split-K.  python tools/exp/x3_splitk_time.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 1700
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
for (N, K, epi) in [(768, 9216, lib.EPI_BIAS_DROP_RESID), (768, 2304, lib.EPI_BIAS_DROP_RESID), (1024, 3072, lib.EPI_BIAS_DROP_RESID),
                    (768, 6912, lib.EPI_BIAS), (2304, 2304, lib.EPI_BIAS), (3072, 2304, lib.EPI_BIAS), (3072, 3072, lib.EPI_BIAS)]:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    resid = epi == lib.EPI_BIAS_DROP_RESID
    o = torch.empty((M, N), device="cuda")
    ax = torch.randn((M, N), device="cuda") if resid else None
    res = []
    for tile, sks in ((0, (0,)), (7, (0, 2, 4, -1)), (1, (0, 2, 4)), (9, (0, 2, 4))):
        for sk in sks:
            t = timeit(lambda: lib.gemm_nt(x, w, o, bias=b, epilogue=epi, aux=ax, tile=tile, splitk=sk, splitk_ws=ws))
            res.append(f"t{tile}/s{sk} {t:5.1f}")
    print(f"M={M} N={N:5d} K={K:5d} epi={epi}: " + "  ".join(res))
