"""What would desynchronised tile boundaries be worth in STEADY STATE?  The persistent NT kernels run their workgroups in lock
step: every CU reaches its epilogue (the stores, and for the residual / multiplier epilogues the reads of a second operand:
HBM traffic) at the same moment, while during the main loops HBM is nearly idle.  A variant build delays half of every XCD's
workgroups by half a tile period at the start of the launch (-DUNIMM_STAGGER_US=16); on the real shapes that idle time cancels
whatever it gains (rounds 1-4), so this script measures the steady state instead: time(4 x the rows) - time(1 x the
rows) = the cost of the extra rounds alone, with and without the offset.
   python tools/exp/desync_steady_state.py        (run once per library: UNIMM_HIP_LIB=... for the variant)"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib

E = lib
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, iters=30, warm=4):
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


shapes = [("attn-out  bias+drop+resid f32", 768, 768, E.EPI_BIAS_DROP_RESID), ("ff2       bias+drop+resid f32", 768, 3072, E.EPI_BIAS_DROP_RESID),
          ("ff1       bias+GELU,GELU'", 3072, 768, E.EPI_BIAS_GELU_DG), ("ff2 dgrad x aux", 3072, 768, E.EPI_MUL),
          ("qkv       bias", 2304, 768, E.EPI_BIAS), ("ff1 dgrad + aux", 768, 3072, E.EPI_ADD)]
M1 = 31162
print(f"library: {os.environ.get('UNIMM_HIP_LIB', 'unimm_amd/libunimm_hip.so')}")
for name, N, K, epi in shapes:
    res = []
    for mult in (1, 4):
        M = M1 * mult
        x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
        w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
        b = torch.randn(N, device="cuda")
        resid = epi == E.EPI_BIAS_DROP_RESID
        o = torch.empty((M, N), device="cuda", dtype=torch.float32 if resid else torch.bfloat16)
        o2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16) if epi == E.EPI_BIAS_GELU_DG else None
        ax = (torch.randn((M, N), device="cuda") if resid else torch.randn((M, N), device="cuda").to(torch.bfloat16)) \
            if epi in (E.EPI_MUL, E.EPI_ADD, E.EPI_BIAS_DROP_RESID) else None
        fn = lambda: lib.gemm_nt(x, w, o, bias=b if epi not in (E.EPI_MUL, E.EPI_ADD) else None, epilogue=epi, aux=ax, out2=o2,
                                 drop=(0x1234, int(0.1 * 2 ** 32), 1.0 / 0.9) if resid else lib.NO_DROP)
        res.append(min(timeit(fn), timeit(fn)))
        del x, o, o2, ax
    fl = 2.0 * M1 * N * K
    steady = (res[1] - res[0]) / 3.0
    print(f"{name:32s} N={N:5d} K={K:5d}: {res[0]:7.1f} us at {M1} rows ({fl / res[0] / 1e6:5.0f} TF/s), {res[1]:7.1f} us at 4 x; "
          f"steady state {steady:7.1f} us per {M1} rows = {fl / steady / 1e6:5.0f} TF/s")
