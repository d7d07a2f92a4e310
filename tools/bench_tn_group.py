"""Grouped weight-gradient launch of one text encoder block (QKV, out-proj, FFN-up, FFN-down) at M rows:
python tools/bench_tn_group.py [M] [shared 0/1]   (UNIMM_TN_SKEW / UNIMM_TN_SPLITS are read once per process)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 31162
shared = bool(int(sys.argv[2])) if len(sys.argv) > 2 else False     # a third argument "nobias": no bias gradients
g = torch.Generator(device="cuda").manual_seed(0)
probs = []
fl = 0.0
nblocks = int(os.environ.get("TN_BLOCKS", "1"))      # text blocks grouped into ONE launch
for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)] * nblocks:
    dy = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
    x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
    dw = torch.zeros((N, K), device="cuda"); db = torch.zeros(N, device="cuda")
    probs.append((dy, x, dw, None, None, None, None if "nobias" in sys.argv else db, None, "overwrite" in sys.argv))
    fl += 2.0 * M * N * K
def timed(code):
    def run(): lib.gemm_tn_grouped(probs, shared=code)
    for _ in range(5): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(50): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / 50 * 1e-3
# interleaved A/B: bit 1 of the hint selects the lock-step loop of rounds 1-2 for the 256x256 tile (tools only)
for rnd in range(3):
    tl, t = timed(int(shared) | 2), timed(int(shared))
    tl, t = tl / nblocks, t / nblocks
    print(f"blocks per launch {nblocks} round {rnd}: per block: lock-step loop {tl*1e6:7.1f} us {fl/nblocks/tl/1e12:7.1f} TFLOP/s | ping-pong loop {t*1e6:7.1f} us {fl/nblocks/t/1e12:7.1f} TFLOP/s")
# correctness against an fp32 matmul of the first problem
dy, x, dw, *_ = probs[1]
dw.zero_(); lib.gemm_tn_grouped([probs[1]], shared=shared); torch.cuda.synchronize()
ref = dy.float().t() @ x.float()
err = ((dw - ref).abs().max() / ref.abs().max()).item()
print(f"{'nobias ' if 'nobias' in sys.argv else ''}M={M} shared={int(shared)} skew={os.environ.get('UNIMM_TN_SKEW','0')} splits={os.environ.get('UNIMM_TN_SPLITS','auto')}: {t*1e6:7.1f} us  {fl/nblocks/t/1e12:7.1f} TFLOP/s  err {err:.1e}")
