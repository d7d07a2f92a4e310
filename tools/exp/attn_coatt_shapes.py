"""The two co-attention directions (D = 128, 8 heads, 37 regions against the text rows) forward + one-kernel backward alone, at
the headline batch's lengths (240 sequences, longest first), with dropout 0.1: us per launch.  For rocprofv3 --pmc runs
(ATTN_SHAPES=coatt bash tools/exp/attn_text_pmc.sh <tag>)."""
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
B, T, R, H, D = A.batch, 256, 37, 8, 128
HD = H * D
off = (torch.cumsum(lens, 0) - lens).to(torch.int32)
M = int(lens.sum())
order = torch.argsort(lens, descending=True, stable=True).to(torch.int32)
co = lib.mask_pack(b["co_attention_mask"])                      # [B, R, T]: regions attend text
vm = lib.mask_pack(b["image_attention_mask"])                   # [B, R]: text attends regions (key mask)
nw, nwv = T // 32, vm.shape[-1]
g = torch.Generator(device=DEV).manual_seed(1)
rnd = lambda *s: torch.randn(s, generator=g, device=DEV).to(torch.bfloat16)
qt, kvt = rnd(M, HD), rnd(M, 2 * HD)                             # text rows (packed)
qv, kvv = rnd(B * R, HD), rnd(B * R, 2 * HD)                     # region rows
drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
var = (off, lens, None, order)
sc = 1.0 / math.sqrt(D)
# regions attend text: Tq = R, Tk = T (var keys)
o1, l1, d1 = torch.empty((B * R, HD), device=DEV, dtype=torch.bfloat16), torch.empty((B, H, R), device=DEV), torch.empty((B, H, R), device=DEV)
do1, dq1, dkv1 = rnd(B * R, HD), torch.empty_like(qv), torch.empty_like(kvt)
# text attends regions: Tq = T (var queries), Tk = R
o2, l2, d2 = torch.empty((M, HD), device=DEV, dtype=torch.bfloat16), torch.empty((B, H, T), device=DEV), torch.empty((B, H, T), device=DEV)
do2, dq2, dkv2 = rnd(M, HD), torch.empty_like(qt), torch.empty_like(kvv)


def fwd():
    lib.attn_fwd(qv, kvt[:, :HD], kvt[:, HD:], o1, l1, co, B, H, R, T, D, sc, nw, R * nw, drop, kvar=var)
    lib.attn_fwd(qt, kvv[:, :HD], kvv[:, HD:], o2, l2, vm, B, H, T, R, D, sc, 0, nwv, drop, qvar=var)


def bwd():
    lib.attn_bwd(qv, kvt[:, :HD], kvt[:, HD:], o1, do1, l1, d1, dq1, dkv1[:, :HD], dkv1[:, HD:], co, B, H, R, T, D, sc, nw, R * nw, drop, kvar=var)
    lib.attn_bwd(qt, kvv[:, :HD], kvv[:, HD:], o2, do2, l2, d2, dq2, dkv2[:, :HD], dkv2[:, HD:], vm, B, H, T, R, D, sc, 0, nwv, drop, qvar=var)


def timeit(fn):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(A.iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / A.iters * 1e3


tf1 = timeit(lambda: lib.attn_fwd(qv, kvt[:, :HD], kvt[:, HD:], o1, l1, co, B, H, R, T, D, sc, nw, R * nw, drop, kvar=var))
tf2 = timeit(lambda: lib.attn_fwd(qt, kvv[:, :HD], kvv[:, HD:], o2, l2, vm, B, H, T, R, D, sc, 0, nwv, drop, qvar=var))
print(f"forward alone: regions attend text {tf1:.1f} us, text attends regions {tf2:.1f} us")
tf = timeit(fwd)
tb = timeit(bwd)
print(f"{B} sequences, {M} text rows, {R} regions, {H} heads of {D}: both directions forward {tf:.1f} us, backward {tb:.1f} us")
