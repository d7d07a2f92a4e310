"""x3 LayerNorm forward / backward alone at the dense step's text shape (12,849 x 768) and image shape (3,700 x 1024)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib, dropout as DR
DEV = "cuda"


def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


for M, H in ((12849, 768), (3700, 1024), (31162, 768)):
    x = torch.randn((M, H), device=DEV); dy = torch.randn((M, H), device=DEV)
    g, b = torch.randn(H, device=DEV), torch.randn(H, device=DEV)
    y32, y3 = torch.empty((M, H), device=DEV), torch.empty((M, 3 * H), dtype=torch.bfloat16, device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    dx32, dxd3 = torch.empty((M, H), device=DEV), torch.empty((M, 3 * H), dtype=torch.bfloat16, device=DEV)
    part = torch.empty(lib.colpartials_bytes(H) // 4, device=DEV)
    drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
    tf = timeit(lambda: lib.x3_layernorm_fwd(x, g, b, y32, y3, mean, rstd, M, H))
    tb = timeit(lambda: lib.x3_layernorm_bwd_partials(dy, x, mean, rstd, g, dx32, dxd3, part, M, H, drop=drop))
    print(f"M={M} H={H}: forward {tf:6.1f} us ({14.0 * M * H / tf / 1e6:5.2f} TB/s)   backward {tb:6.1f} us ({18.0 * M * H / tb / 1e6:5.2f} TB/s)")
