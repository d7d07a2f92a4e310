"""LayerNorm forward / backward row kernels alone at the row counts of a 30-sequence step (text ~3,900 x 768, regions
1,110 x 1024) and of the 240-sequence step (31,162 x 768, 8,880 x 1024): us per launch and the bytes they move.
UNIMM_HIP_LIB selects a variant library (tools/exp/variant_lib.sh)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib, dropout as DR
DEV = "cuda"
BF = torch.bfloat16


def timeit(fn, iters=100, warm=10):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


print("library:", os.environ.get("UNIMM_HIP_LIB", "(default)"))
for M, H in ((3900, 768), (1110, 1024), (7800, 768), (31162, 768), (8880, 1024)):
    x = torch.randn((M, H), device=DEV); dy = torch.randn((M, H), device=DEV).to(BF)
    g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    y32, y16 = torch.empty((M, H), device=DEV), torch.empty((M, H), dtype=BF, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    dx, dxd = torch.empty((M, H), dtype=BF, device=DEV), torch.empty((M, H), dtype=BF, device=DEV)
    part = torch.empty(lib.colpartials_bytes(H) // 4, device=DEV)
    drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
    tf = timeit(lambda: lib.layernorm_fwd(x, g, b, y32, y16, mean, rstd, M, H))
    nb = [0]
    def bwd(d=drop, o=dxd):
        nb[0] = lib.layernorm_bwd_partials(dy, x, mean, rstd, g, dx, o, part, M, H, drop=d, m_dev=None)
    tb = timeit(bwd)
    tb0 = timeit(lambda: bwd(None, None))
    by_f, by_b = 10.0 * M * H, 10.0 * M * H + nb[0] * 3 * H * 4
    print(f"M={M:6d} H={H}: forward {tf:6.1f} us ({by_f / tf / 1e6:5.2f} TB/s)   backward+dropout {tb:6.1f} us ({by_b / tb / 1e6:5.2f} TB/s)"
          f"   backward {tb0:6.1f} us   [{nb[0]} blocks]")
