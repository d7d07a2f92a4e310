"""Does a GEMM's time depend on where its X operand comes from?  The same launch on one X buffer over and over (it stays in the
256 MiB Infinity Cache) against a rotation over buffers that together exceed the cache (every launch streams X from HBM, as in
the step, where X was just written by the previous kernel with non-temporal stores):  python tools/exp/cold_operand.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
M = 31162
def timeit(fn, iters=60, warm=6):
    for i in range(warm): fn(i)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for i in range(iters): fn(i)
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
E = lib
for (N, K, epi, tile) in [(768, 3072, E.EPI_ADD, 0), (768, 2304, E.EPI_ADD, 0), (768, 3072, E.EPI_BIAS_DROP_RESID, 0), (3072, 768, E.EPI_BIAS_GELU_DG, 0),
                          (3072, 768, E.EPI_MUL, 0), (768, 768, E.EPI_BIAS_DROP_RESID, 0), (2304, 768, E.EPI_BIAS, 0)]:
    nbuf = 6
    xs = [torch.randn((M, K), device="cuda").to(torch.bfloat16) for _ in range(nbuf)]
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    resid = epi == E.EPI_BIAS_DROP_RESID
    os_ = [torch.empty((M, N), device="cuda", dtype=torch.float32 if resid else torch.bfloat16) for _ in range(nbuf)]
    o2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    axs = [(torch.randn((M, N), device="cuda") if resid else torch.randn((M, N), device="cuda").to(torch.bfloat16)) for _ in range(nbuf)]
    def run(i, rot):
        j = i % nbuf if rot else 0
        lib.gemm_nt(xs[j], w, os_[j], bias=b if epi not in (E.EPI_MUL, E.EPI_ADD) else None, epilogue=epi,
                    aux=axs[j] if epi in (E.EPI_MUL, E.EPI_ADD, E.EPI_BIAS_DROP_RESID) else None, out2=o2 if epi == E.EPI_BIAS_GELU_DG else None, tile=tile)
    out = []
    for tl in [int(t) for t in os.environ.get("TILES", "0").split(",")]:
        tile = tl
        timeit(lambda i: run(i, False), iters=20)
        warm = timeit(lambda i: run(i, False))
        cold = timeit(lambda i: run(i, True))
        out.append(f"tile {tl}: resident {warm:6.1f} us, rotating {nbuf} buffers {cold:6.1f} us (+{(cold / warm - 1) * 100:4.1f} %)")
    print(f"N={N:5d} K={K:5d} epi={epi}: " + "   ".join(out))
