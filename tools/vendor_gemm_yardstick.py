"""Vendor yardstick: this build's NT / TN kernels against `torch.matmul` (hipBLASLt / rocBLAS through
PyTorch-ROCm) on the 20 hot forward / input-gradient shapes of the headline step and on the weight-gradient shapes, alone on
the chip, same buffers, alternating order.  The vendor GEMM computes ONLY the product (bf16 in, bf16 out): it is the yardstick
for the main loop, i.e. for the plain `bias -> bf16` column; the fused-epilogue column is what the step runs (bias + GELU + GELU',
bias + dropout + fp32 residual, x aux, + aux), work the vendor path would need extra elementwise kernels for -- the `+ eltwise`
column adds those kernels' time (torch ops of the same math) to the vendor product.
    python tools/vendor_gemm_yardstick.py > gpurun_out/.../vendor_gemm_yardstick.txt"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import lib

Mt, Mi = 31162, 8880
g = torch.Generator(device="cuda").manual_seed(0)
E = lib


def timeit(fn, iters=40, warm=4):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


shapes = [  # (name, M, N, K, epilogue of the step)
    ("text qkv fwd", Mt, 2304, 768, E.EPI_BIAS), ("text attn-out fwd", Mt, 768, 768, E.EPI_BIAS_DROP_RESID),
    ("text ff1 fwd", Mt, 3072, 768, E.EPI_BIAS_GELU_DG), ("text ff2 fwd", Mt, 768, 3072, E.EPI_BIAS_DROP_RESID),
    ("text ff2 dgrad", Mt, 3072, 768, E.EPI_MUL), ("text ff1 dgrad", Mt, 768, 3072, E.EPI_ADD),
    ("text attn-out dgrad", Mt, 768, 768, E.EPI_BIAS), ("text qkv dgrad", Mt, 768, 2304, E.EPI_ADD),
    ("conn text qkv2 fwd", Mt, 3072, 768, E.EPI_BIAS), ("conn text bi-out fwd", Mt, 768, 1024, E.EPI_BIAS_DROP_RESID),
    ("conn text bi-out dgrad", Mt, 1024, 768, E.EPI_BIAS), ("conn text qkv2 dgrad", Mt, 768, 3072, E.EPI_ADD),
    ("image qkv fwd", Mi, 3072, 1024, E.EPI_BIAS), ("image attn-out fwd", Mi, 1024, 1024, E.EPI_BIAS_DROP_RESID),
    ("image ff1 fwd", Mi, 1024, 1024, E.EPI_BIAS_GELU_DG), ("image ff2 fwd", Mi, 1024, 1024, E.EPI_BIAS_DROP_RESID),
    ("image ff2 dgrad", Mi, 1024, 1024, E.EPI_MUL), ("image ff1 dgrad", Mi, 1024, 1024, E.EPI_ADD),
    ("image attn-out dgrad", Mi, 1024, 1024, E.EPI_BIAS), ("image qkv dgrad", Mi, 1024, 3072, E.EPI_ADD),
]
_x = torch.randn((Mt, 768), device="cuda").to(torch.bfloat16)
_w = torch.randn((3072, 768), device="cuda").to(torch.bfloat16)
_o = torch.empty((Mt, 3072), device="cuda", dtype=torch.bfloat16)
timeit(lambda: lib.gemm_nt(_x, _w, _o), iters=300)          # clocks up
print(f"torch {torch.__version__}, hip {torch.version.hip}; preferred BLAS backend: {torch.backends.cuda.preferred_blas_library()}")
print("NT GEMMs (forward and input gradients), OUT[M,N] = epi(X[M,K] W[N,K]^T), us per launch alone on the chip (TFLOP/s):")
print(f"{'shape':24s} {'M':>6s} {'N':>5s} {'K':>5s} {'epilogue of the step':>22s} | {'vendor X W^T':>16s} | {'own, plain bias':>16s} | {'own, fused epi':>16s} | {'vendor + eltwise':>16s}")
EN = {E.EPI_BIAS: "bias", E.EPI_BIAS_DROP_RESID: "bias+drop+resid f32", E.EPI_BIAS_GELU_DG: "bias+GELU,GELU'", E.EPI_MUL: "x aux", E.EPI_ADD: "+ aux"}
tot = [0.0, 0.0, 0.0, 0.0]
fl_tot = 0.0
for name, M, N, K, epi in shapes:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    wT = w.t()
    b = torch.randn(N, device="cuda")
    b16 = b.to(torch.bfloat16)
    resid = epi == E.EPI_BIAS_DROP_RESID
    o16 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    o = torch.empty((M, N), device="cuda", dtype=torch.float32) if resid else o16
    o2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ax = torch.randn((M, N), device="cuda") if resid else torch.randn((M, N), device="cuda").to(torch.bfloat16)
    ov = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)

    def vendor():
        torch.matmul(x, wT, out=ov)

    def vendor_elt():
        torch.matmul(x, wT, out=ov)
        if epi == E.EPI_BIAS:
            ov.add_(b16)
        elif epi == E.EPI_BIAS_GELU_DG:
            u = ov + b16
            torch.nn.functional.gelu(u, approximate="none")
            u.float().pow(2)                                   # stand-in for the GELU' pass (one more read + write of the tensor)
        elif epi == E.EPI_BIAS_DROP_RESID:
            torch.add(ax, torch.nn.functional.dropout(ov.float() + b, 0.1, True), out=o)
        elif epi == E.EPI_MUL:
            ov.mul_(ax)
        else:
            ov.add_(ax)

    def own_plain():
        lib.gemm_nt(x, w, o16, bias=b, epilogue=E.EPI_BIAS)

    def own_fused():
        lib.gemm_nt(x, w, o, bias=b if epi not in (E.EPI_MUL, E.EPI_ADD) else None, epilogue=epi,
                    aux=ax if epi in (E.EPI_MUL, E.EPI_ADD, E.EPI_BIAS_DROP_RESID) else None,
                    out2=o2 if epi == E.EPI_BIAS_GELU_DG else None, drop=(0x1234, int(0.1 * 2 ** 32), 1.0 / 0.9) if resid else lib.NO_DROP)

    fns = [vendor, own_plain, own_fused, vendor_elt]
    best = [1e30] * 4
    for rnd in range(2):                                      # alternating order, best of two rounds
        for i, fn in enumerate(fns):
            best[i] = min(best[i], timeit(fn))
    fl = 2.0 * M * N * K
    fl_tot += fl
    for i in range(4):
        tot[i] += best[i]
    print(f"{name:24s} {M:6d} {N:5d} {K:5d} {EN[epi]:>22s} | " + " | ".join(f"{u:7.1f} ({fl / u / 1e6:5.0f})" for u in best))
print(f"{'sum of the 20 shapes':64s} | " + " | ".join(f"{u:7.0f} ({fl_tot / u / 1e6:5.0f})" for u in tot))

print("\nTN GEMMs (weight gradients), DW[N,K] = DY[M,N]^T X[M,K] (fp32 result), us per launch (TFLOP/s):")
print(f"{'shape':24s} {'M':>6s} {'N':>5s} {'K':>5s} | {'vendor dY^T X':>16s} | {'own, one problem':>16s} | {'own, grouped x7':>16s}")
for name, M, N, K in [("text qkv", Mt, 2304, 768), ("text attn-out", Mt, 768, 768), ("text ff1", Mt, 3072, 768), ("text ff2", Mt, 768, 3072),
                      ("conn text bi-out", Mt, 768, 1024), ("image qkv", Mi, 3072, 1024), ("image 1024x1024", Mi, 1024, 1024)]:
    dys = [torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16) for _ in range(7)]
    xs = [torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16) for _ in range(7)]
    dws = [torch.zeros((N, K), device="cuda") for _ in range(7)]
    dwv = torch.empty((N, K), device="cuda", dtype=torch.bfloat16)
    dyT = dys[0].t()
    tv = timeit(lambda: torch.matmul(dyT, xs[0], out=dwv))
    t1 = timeit(lambda: lib.gemm_tn(dys[0], xs[0], dws[0]))
    probs = [(dys[i], xs[i], dws[i], None, None, None, None, None, True) for i in range(7)]
    t7 = timeit(lambda: lib.gemm_tn_grouped(probs, shared=0)) / 7
    fl = 2.0 * M * N * K
    print(f"{name:24s} {M:6d} {N:5d} {K:5d} | " + " | ".join(f"{u:7.1f} ({fl / u / 1e6:5.0f})" for u in (tv, t1, t7)))
print("(vendor TN: bf16 result of one product, no accumulation into the fp32 gradient, no bias gradient; own: fp32 result, written"
      " once per tile; 'grouped x7' = seven problems of the shape in ONE launch, as the engine queues them, time per problem)")
