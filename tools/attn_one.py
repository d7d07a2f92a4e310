"""Run the text self-attention kernels (fwd, bwd) on a synthetic varlen batch, for rocprofv3 --pmc passes."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from unimm_amd import lib, dropout as DR
B, H, T, D = 240, 12, 256, 64
HD = H * D
rng = np.random.default_rng(0)
lens = rng.integers(50, 230, size=B)
varlen = len(sys.argv) < 2 or sys.argv[1] != "padded"
g = torch.Generator(device="cuda").manual_seed(0)
m = torch.zeros((B, T, T), dtype=torch.bool, device="cuda")
for b, l in enumerate(lens):
    m[b, :l, :l] = True
packed = lib.mask_pack(m)
nw = T // 32
Mv = int(lens.sum()) if varlen else B * T
qkv = torch.randn((Mv, 3 * HD), generator=g, device="cuda").to(torch.bfloat16)
dout = torch.randn((Mv, HD), generator=g, device="cuda").to(torch.bfloat16)
out = torch.empty((Mv, HD), device="cuda", dtype=torch.bfloat16)
dqkv = torch.empty_like(qkv)
lse = torch.empty((B, H, T), device="cuda"); delta = torch.empty_like(lse)
off = torch.from_numpy(np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int32)).cuda()
ln = torch.from_numpy(lens.astype(np.int32)).cuda()
var = (off, ln) if varlen else None
drop = DR.drop_arg(0.1, 12345)
def run():
    lib.attn_fwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], out, lse, packed, B, H, T, T, D, D ** -0.5, nw, T * nw, drop, qvar=var, kvar=var)
    lib.attn_bwd(qkv[:, :HD], qkv[:, HD:2*HD], qkv[:, 2*HD:], out, dout, lse, delta, dqkv[:, :HD], dqkv[:, HD:2*HD], dqkv[:, 2*HD:],
                 packed, B, H, T, T, D, D ** -0.5, nw, T * nw, drop, qvar=var, kvar=var)
for parts in (1, 2, 1, 2):
    lib.attn_set_parts(parts)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(5): run()
    e.record(); torch.cuda.synchronize()
    print(f"varlen={varlen} rows={Mv} parts={parts} fwd+bwd {s.elapsed_time(e) / 5 * 1e3:.1f} us")
