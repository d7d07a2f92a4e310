import sys, os
sys.path.insert(0, os.getcwd())
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
g = torch.Generator(device="cuda").manual_seed(0)
for M in ([int(a) for a in sys.argv[1].split(',')] if len(sys.argv) > 1 else (8880, 4000, 1110)):
  print("M", M)
  for (N, K, epi) in [(3072, 1024, lib.EPI_BIAS), (1024, 1024, lib.EPI_BIAS_DROP_RESID), (1024, 1024, lib.EPI_BIAS_GELU_DG), (1024, 1024, lib.EPI_ADD), (1024, 3072, lib.EPI_ADD), (768, 768, lib.EPI_ADD), (768, 3072, lib.EPI_ADD), (3072, 768, lib.EPI_MUL), (768, 768, lib.EPI_BIAS_DROP_RESID), (2304, 768, lib.EPI_BIAS), (768, 2304, lib.EPI_ADD)]:
    x = (torch.randn((M, K), generator=g, device="cuda")).to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    resid = epi == lib.EPI_BIAS_DROP_RESID
    o = torch.empty((M, N), device="cuda") if resid else torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ax = torch.randn((M, N), device="cuda") if resid else torch.randn((M, N), device="cuda").to(torch.bfloat16)
    out2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    res = []
    for cfg in ([int(a) for a in sys.argv[2].split(',')] if len(sys.argv) > 2 else (0, 1, 3, 6, 7)):
        t = timeit(lambda: lib.gemm_nt(x, w, o, bias=b, epilogue=epi, aux=ax, out2=out2 if epi == lib.EPI_BIAS_GELU_DG else None, tile=cfg))
        res.append(f"cfg{cfg} {t*1e6:6.1f}")
    print(f"  N={N:5d} K={K:5d} epi={epi}: " + "  ".join(res))
