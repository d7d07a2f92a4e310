"""unimm_gemm_nt_pair against two unimm_gemm_nt launches: the connection layer's GEMMs that both sides have to do at the same moment
with a plain epilogue (models/vilbert_dialog.py:659-672, :745-748 and autograd), alone on the chip, at the headline batch.
Needs tools/exp/nt_pair_experiment.patch applied (git apply; python -m unimm_amd.build): the pair launch was measured in round 6
(profiles/r6m_nt_pair_launch_experiment.txt: 644 -> 629 us over four pairs, bit-equal results) and not merged."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib

Mt, Mi = 31162, 8880
E = lib
g = torch.Generator(device="cuda").manual_seed(0)


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


def prob(M, N, K, epi):
    d = dict(x=torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16),
             w=(torch.randn((N, K), generator=g, device="cuda") * 0.05).to(torch.bfloat16),
             out=torch.empty((M, N), device="cuda", dtype=torch.bfloat16))
    if epi == E.EPI_BIAS:
        d["bias"] = torch.randn(N, device="cuda")
    else:
        d["aux"] = torch.randn((M, N), device="cuda").to(torch.bfloat16)
    return d


pairs = [("fused Q/K/V projections, forward", (Mt, 3072, 768), (Mi, 3072, 1024), E.EPI_BIAS),
         ("fused Q/K/V projections, input gradients", (Mt, 768, 3072), (Mi, 1024, 3072), E.EPI_ADD),
         ("bi-output dense, input gradients", (Mt, 1024, 768), (Mi, 1024, 1024), E.EPI_BIAS),
         ("FFN up, input gradients", (Mt, 768, 3072), (Mi, 1024, 1024), E.EPI_ADD)]
tot = [0.0, 0.0]
for name, sa, sb, epi in pairs:
    a, b = prob(*sa, epi), prob(*sb, epi)

    def sep():
        lib.gemm_nt(a["x"], a["w"], a["out"], bias=a.get("bias"), epilogue=epi, aux=a.get("aux"))
        lib.gemm_nt(b["x"], b["w"], b["out"], bias=b.get("bias"), epilogue=epi, aux=b.get("aux"))

    def pair():
        lib.gemm_nt_pair(a, b, epilogue=epi)

    sep(); torch.cuda.synchronize()
    want = (a["out"].clone(), b["out"].clone())
    a["out"].zero_(); b["out"].zero_()
    pair(); torch.cuda.synchronize()
    same = torch.equal(want[0], a["out"]) and torch.equal(want[1], b["out"])
    ts = min(timeit(sep), timeit(sep))
    tp = min(timeit(pair), timeit(pair))
    ts = min(ts, timeit(sep)); tp = min(tp, timeit(pair))
    fl = 2.0 * (sa[0] * sa[1] * sa[2] + sb[0] * sb[1] * sb[2])
    tot[0] += ts; tot[1] += tp
    print(f"{name:44s} text {sa}  image {sb}: two launches {ts:7.1f} us ({fl / ts / 1e6:5.0f} TF/s)  one paired launch {tp:7.1f} us "
          f"({fl / tp / 1e6:5.0f} TF/s)  bit-equal {same}")
print(f"sum: two launches {tot[0]:.0f} us, paired {tot[1]:.0f} us")
