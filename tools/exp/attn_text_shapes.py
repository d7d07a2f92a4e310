"""The bf16 text self-attention (D = 64, 12 heads) forward + one-kernel backward alone, at the headline batch's lengths (240
sequences, the synthetic batch's own valid lengths, longest first), with dropout 0.1: us per launch.  For rocprofv3 --pmc runs
(tools/exp/attn_text_pmc.sh)."""
import argparse, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import BertConfig, lib, synth
from unimm_amd import dropout as DR

ap = argparse.ArgumentParser()
ap.add_argument("--iters", type=int, default=20)
ap.add_argument("--batch", type=int, default=240)
A = ap.parse_args()
DEV = "cuda"
cfg = BertConfig.from_json_file("unimm_amd/config/bert_base_6layer_6conect.json")
b = synth.make_batch(n_seq=A.batch, cfg=cfg, seed=1234, device=DEV)
am = b["attention_mask"]
lens = am.ne(0).any(-1).sum(1).to(torch.int32)
B, T, H, D = A.batch, 256, 12, 64
HD = H * D
off = (torch.cumsum(lens, 0) - lens).to(torch.int32)
M = int(lens.sum())
order = torch.argsort(lens, descending=True, stable=True).to(torch.int32)
packed = lib.mask_pack(am)
nw = T // 32
g = torch.Generator(device=DEV).manual_seed(1)
qkv = torch.randn((M, 3 * HD), generator=g, device=DEV).to(torch.bfloat16)
dout = torch.randn((M, HD), generator=g, device=DEV).to(torch.bfloat16)
out = torch.empty((M, HD), device=DEV, dtype=torch.bfloat16)
lse = torch.empty((B, H, T), device=DEV)
delta = torch.empty((B, H, T), device=DEV)
dqkv = torch.empty_like(qkv)
drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
var = (off, lens, None, order)
sc = 1.0 / math.sqrt(D)


def fwd():
    lib.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, lse, packed, B, H, T, T, D, sc, nw, T * nw, drop, qvar=var, kvar=var)


def bwd():
    lib.attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out, dout, lse, delta, dqkv[:, :HD], dqkv[:, HD:2 * HD], dqkv[:, 2 * HD:],
                 packed, B, H, T, T, D, sc, nw, T * nw, drop, qvar=var, kvar=var)


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(A.iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / A.iters * 1e3


tf = timeit(fwd)
tb = timeit(bwd)
tiles = ((lens + 31) // 32).float()
print(f"{B} sequences, {M} rows (mean {M / B:.0f}), {H} heads: forward {tf:.1f} us, backward {tb:.1f} us;  32x32 score tiles per head: {int((tiles * tiles).sum())}")
