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
M, N = 61440, 3072
for K in (768,):
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    aux = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    aux32 = torch.randn((M, N), device="cuda")
    out = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    out2 = torch.empty_like(out)
    o32 = torch.empty((M, N), device="cuda")
    for name, fn in [("BIAS bf16", lambda: lib.gemm_nt(x, w, out, bias=b)),
                     ("GELU bf16 +u", lambda: lib.gemm_nt(x, w, out, bias=b, epilogue=lib.EPI_BIAS_GELU, out2=out2)),
                     ("GELU bf16", lambda: lib.gemm_nt(x, w, out, bias=b, epilogue=lib.EPI_BIAS_GELU)),
                     ("DGELU", lambda: lib.gemm_nt(x, w, out, epilogue=lib.EPI_DGELU, aux=aux)),
                     ("ADD", lambda: lib.gemm_nt(x, w, out, epilogue=lib.EPI_ADD, aux=aux)),
                     ("RESID f32 nodrop", lambda: lib.gemm_nt(x, w, o32, bias=b, epilogue=lib.EPI_BIAS_DROP_RESID, aux=aux32)),
                     ("RESID f32 drop", lambda: lib.gemm_nt(x, w, o32, bias=b, epilogue=lib.EPI_BIAS_DROP_RESID, aux=aux32, drop=(123, 429496729, 1.111)))]:
        ts = []
        for cfg in (3 + 10 * 0 + 1000, 3 + 10 * 1 + 1000, 3 + 10 * 2 + 1000, 3 + 10 * 4 + 1000, 3 + 1000):
            lib.gemm_set_tile(cfg)
            ts.append(timeit(fn))
        print(f"K={K:4d} {name:18s} " + " ".join(f"{t*1e6:8.1f}" for t in ts) + "  us  (stagger off,1,2,4,off)")
