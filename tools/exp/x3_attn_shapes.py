"""fp32 attention cores at the shapes of the dense fine-tune step (100 sequences, ~128 valid tokens of 256, 37 regions):
matrix-instruction kernels against the vector-ALU kernels (unimm_x3_attn_set_impl), per launch, with dropout 0.1."""
import sys, os, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
from unimm_amd import dropout as DR

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--only", default="", help="substring of the case name")
ap.add_argument("--impls", default="1,0", help="1 = matrix kernels, 0 = vector kernels")
ap.add_argument("--iters", type=int, default=30)
ap.add_argument("--batch", type=int, default=100)
ap.add_argument("--lens", default="40,216", help="lo,hi of the uniform length distribution; 'sorted' appended = longest first")
ARGS = ap.parse_args()
IMPLS = [int(x) for x in ARGS.impls.split(",")]
DEV = "cuda"
B = ARGS.batch


def timeit(fn, iters=None, warm=3):
    iters = iters or ARGS.iters
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


def case(name, H, Tq, Tk, D, q_var, k_var, seed=0):
    if ARGS.only not in name:
        return
    g = torch.Generator().manual_seed(seed)
    lo, hi = [int(x) for x in ARGS.lens.split(",")[:2]]
    lens = torch.randint(lo, hi + 1, (B,), generator=g).to(torch.int32)      # 40..216: ~128 on average
    if ARGS.lens.endswith("sorted"):
        lens = lens.sort(descending=True).values
    ql = lens if q_var else torch.full((B,), Tq, dtype=torch.int32)
    kl = lens if k_var else torch.full((B,), Tk, dtype=torch.int32)
    nq, nk = int(ql.sum()), int(kl.sum())
    HD = H * D
    q = torch.randn((nq, HD), device=DEV)
    k = torch.randn((nk, HD), device=DEV)
    v = torch.randn((nk, HD), device=DEV)
    m = torch.ones((B, 1, Tk), dtype=torch.bool, device=DEV)
    for b in range(B):
        m[b, :, int(kl[b]):] = False
    packed = lib.mask_pack(m)
    mq, mb = 0, (Tk + 31) // 32
    off = lambda l: torch.cat([torch.zeros(1, dtype=torch.int32), l.cumsum(0)[:-1].to(torch.int32)]).to(DEV)
    qvar = (off(ql), ql.to(DEV)) if q_var else None
    kvar = (off(kl), kl.to(DEV)) if k_var else None
    out = torch.zeros((nq, HD), device=DEV)
    lse = torch.zeros((B, H, Tq), device=DEV)
    delta = torch.zeros_like(lse)
    dout = torch.randn_like(out)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    drop = DR.drop_arg(0.1, DR.make_key(7, 1, seed))
    scale = 1.0 / math.sqrt(D)
    fwd = lambda: lib.x3_attn_fwd(q, k, v, out, lse, packed, B, H, Tq, Tk, D, scale, mq, mb, drop, qvar=qvar, kvar=kvar)
    bwd = lambda: lib.x3_attn_bwd(q, k, v, out, dout, lse, delta, dq, dk, dv, packed, B, H, Tq, Tk, D, scale, mq, mb, drop,
                                  qvar=qvar, kvar=kvar)
    pairs = float((ql.double() * kl.double()).sum()) * H
    ff, fb = 4.0 * pairs * D, 10.0 * pairs * D
    line = f"{name:34s}"
    for impl in IMPLS:
        lib.x3_attn_set_impl(impl)
        tf, tb = timeit(fwd), timeit(bwd)
        line += f" | {'matrix' if impl else 'vector'}: fwd {tf:7.1f} us ({ff / tf / 1e6:5.1f} TF/s) bwd {tb:7.1f} us ({fb / tb / 1e6:5.1f} TF/s)"
    lib.x3_attn_set_impl(1)
    print(line)


case("text self (12 heads x 64)", 12, 256, 256, 64, True, True)
case("image self (8 x 128, 37 regions)", 8, 37, 37, 128, False, False)
case("text attends regions (8 x 128)", 8, 256, 37, 128, True, False)
case("regions attend text (8 x 128)", 8, 37, 256, 128, False, True)
