"""Micro-benchmark of the attention kernels at the four hot-path shapes (synthetic varlen batch):
text self-attention, image self-attention, co-attention direction 1 (text queries, region keys) and
direction 2 (region queries, text keys).  Prints fwd and bwd time per call."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unimm_amd import lib, dropout as DR

B, T, R = 240, 256, 37
rng = np.random.default_rng(0)
lens = rng.integers(50, 230, size=B)
Mv = int(lens.sum())
off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)).cuda()
ln = torch.from_numpy(lens.astype(np.int32)).cuda()
var = (off, ln)
g = torch.Generator(device="cuda").manual_seed(0)
drop = DR.drop_arg(0.0 if "nodrop" in sys.argv else 0.1, 12345)


def rnd(rows, cols):
    return torch.randn((rows, cols), generator=g, device="cuda").to(torch.bfloat16)


def bench(name, H, D, Tq, Tk, q_rows, k_rows, mask, mq, mb, qvar, kvar):
    HD = H * D
    q, k, v, dout = rnd(q_rows, HD), rnd(k_rows, HD), rnd(k_rows, HD), rnd(q_rows, HD)
    out, dq, dk, dv = torch.empty_like(q), torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lse = torch.empty((B, H, Tq), device="cuda"); delta = torch.empty_like(lse)
    scale = D ** -0.5

    def fwd():
        lib.attn_fwd(q, k, v, out, lse, mask, B, H, Tq, Tk, D, scale, mq, mb, drop, qvar=qvar, kvar=kvar)

    def bwd():
        lib.attn_bwd(q, k, v, out, dout, lse, delta, dq, dk, dv, mask, B, H, Tq, Tk, D, scale, mq, mb, drop, qvar=qvar, kvar=kvar)

    res = []
    for fn in (fwd, bwd):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(10): fn()
        e.record(); torch.cuda.synchronize()
        res.append(s.elapsed_time(e) / 10 * 1e3)
    print(f"{name:6s} H={H} D={D} Tq={Tq} Tk={Tk}: fwd {res[0]:7.1f} us   bwd {res[1]:7.1f} us")


which = [a for a in sys.argv[1:] if a != "nodrop"] or ["text", "img", "dir1", "dir2"]
if "text" in which:
    m = torch.zeros((B, T, T), dtype=torch.bool, device="cuda")
    for b, l in enumerate(lens):
        m[b, :l, :l] = True
    bench("text", 12, 64, T, T, Mv, Mv, lib.mask_pack(m), T // 32, T * (T // 32), var, var)
vm = torch.ones((B, 1, R), dtype=torch.bool, device="cuda")
if "img" in which:
    bench("img", 8, 128, R, R, B * R, B * R, lib.mask_pack(vm), 0, 2, None, None)
if "dir1" in which:   # text queries attend the 37 regions (image key mask)
    bench("dir1", 8, 128, T, R, Mv, B * R, lib.mask_pack(vm), 0, 2, var, None)
if "dir2" in which:   # region queries attend the text keys (co-attention mask, one row per region)
    cm = torch.zeros((B, R, T), dtype=torch.bool, device="cuda")
    for b, l in enumerate(lens):
        cm[b, :, :l] = True
    bench("dir2", 8, 128, R, T, B * R, Mv, lib.mask_pack(cm), T // 32, R * (T // 32), None, var)
