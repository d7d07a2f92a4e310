"""Run one GEMM shape a few times (for rocprofv3 --pmc passes)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import lib
kind, M, N, K = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
epi = int(sys.argv[5]) if len(sys.argv) > 5 else 0
g = torch.Generator(device="cuda").manual_seed(0)
if kind == "nt":
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    resid = epi == lib.EPI_BIAS_DROP_RESID
    out = torch.empty((M, N), device="cuda", dtype=torch.float32 if resid else torch.bfloat16)
    aux = torch.randn((M, N), device="cuda") if resid else torch.randn((M, N), device="cuda").to(torch.bfloat16)
    out2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    for _ in range(5):
        lib.gemm_nt(x, w, out, bias=b, epilogue=epi, aux=aux, out2=out2 if epi == 1 else None)
else:
    dy = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    dw = torch.zeros((N, K), device="cuda")
    db = torch.zeros(N, device="cuda")
    for _ in range(5):
        lib.gemm_tn(dy, x, dw, dbias=db)
torch.cuda.synchronize()
