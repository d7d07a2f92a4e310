"""The connection layers' forward / input-gradient GEMM shapes (models/vilbert_dialog.py:655-783) alone on the chip at the
headline batch (31,162 text rows, 8,880 image rows): automatic tile choice against every forced tile."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
Mt, Mi = 31162, 8880
g = torch.Generator(device="cuda").manual_seed(0)


def timeit(fn, iters=60, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3


E = lib
shapes = [  # (name, M, N, K, epilogue)
    ("text qkv2 fwd", Mt, 3072, 768, E.EPI_BIAS), ("text bi-output fwd", Mt, 768, 1024, E.EPI_BIAS_DROP_RESID),
    ("text ff1 fwd", Mt, 3072, 768, E.EPI_BIAS_GELU_DG), ("text ff2 fwd", Mt, 768, 3072, E.EPI_BIAS_DROP_RESID),
    ("text ff2 dgrad", Mt, 3072, 768, E.EPI_MUL), ("text ff1 dgrad", Mt, 768, 3072, E.EPI_ADD),
    ("text bi-output dgrad", Mt, 1024, 768, E.EPI_BIAS), ("text qkv2 dgrad", Mt, 768, 3072, E.EPI_ADD),
    ("image qkv1 fwd", Mi, 3072, 1024, E.EPI_BIAS), ("image bi-output fwd", Mi, 1024, 1024, E.EPI_BIAS_DROP_RESID),
    ("image ff1 fwd", Mi, 1024, 1024, E.EPI_BIAS_GELU_DG), ("image ff2 fwd", Mi, 1024, 1024, E.EPI_BIAS_DROP_RESID),
    ("image ff2 dgrad", Mi, 1024, 1024, E.EPI_MUL), ("image ff1 dgrad", Mi, 1024, 1024, E.EPI_ADD),
    ("image bi-output dgrad", Mi, 1024, 1024, E.EPI_BIAS), ("image qkv1 dgrad", Mi, 1024, 3072, E.EPI_ADD),
]
if os.environ.get("TEXT_LAYER"):          # the text layers' own shapes (models/vilbert_dialog.py:385-483) on top
    shapes += [("text qkv fwd", Mt, 2304, 768, E.EPI_BIAS), ("text attn-out fwd", Mt, 768, 768, E.EPI_BIAS_DROP_RESID),
               ("text attn-out dgrad", Mt, 768, 768, E.EPI_BIAS), ("text qkv dgrad", Mt, 768, 2304, E.EPI_ADD)]
tiles = [int(t) for t in os.environ.get("TILES", "0,8,6,1,7").split(",")]
_x = torch.randn((Mt, 768), device="cuda").to(torch.bfloat16); _w = torch.randn((3072, 768), device="cuda").to(torch.bfloat16)
_o = torch.empty((Mt, 3072), device="cuda", dtype=torch.bfloat16)
timeit(lambda: lib.gemm_nt(_x, _w, _o), iters=300)          # clocks and caches up before the first row is timed
tot = {t: 0.0 for t in tiles}; best = 0.0; flops = 0.0
print(f"{'shape':24s} {'M':>6s} {'N':>5s} {'K':>5s}  " + "  ".join(f"{lib._TILE_NAMES.get(t % 100, 'auto') + ('' if t < 100 else ('p' if t // 100 == 1 else 'n')):>10s}" for t in tiles))
for name, M, N, K, epi in shapes:
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    w = (torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    resid = epi == E.EPI_BIAS_DROP_RESID
    o = torch.empty((M, N), device="cuda", dtype=torch.float32 if resid else torch.bfloat16)
    o2 = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    ax = torch.randn((M, N), device="cuda") if resid else torch.randn((M, N), device="cuda").to(torch.bfloat16)
    row = []
    for t in tiles + tiles:                    # two passes per shape, the second one counts (the first launches on fresh buffers are slow)
        us = timeit(lambda: lib.gemm_nt(x, w, o, bias=b if epi not in (E.EPI_MUL, E.EPI_ADD) else None, epilogue=epi,
                                        aux=ax if epi in (E.EPI_MUL, E.EPI_ADD, E.EPI_BIAS_DROP_RESID) else None,
                                        out2=o2 if epi == E.EPI_BIAS_GELU_DG else None, tile=t))
        row.append(us)
    row = row[len(tiles):]
    for t, us in zip(tiles, row):
        tot[t] += us
    best += min(row); flops += 2.0 * M * N * K
    print(f"{name:24s} {M:6d} {N:5d} {K:5d}  " + "  ".join(f"{u:7.1f} us" for u in row) + f"   auto {2.0 * M * N * K / row[0] / 1e6:6.0f} TF/s")
print("sum per layer: " + "  ".join(f"{lib._TILE_NAMES.get(t % 100, 'auto') + ('' if t < 100 else ('p' if t // 100 == 1 else 'n'))} {tot[t]:.0f} us ({flops / tot[t] / 1e6:.0f} TF/s)" for t in tiles) +
      f"   best-of per shape {best:.0f} us ({flops / best / 1e6:.0f} TF/s)")
