"""Does L2 panel sharing bound the 256x256 weight-gradient kernel?  Same tile count, same reduction length, three operand
layouts of the grouped launch:  (a) the text blocks' shapes (panels shared by 3-12 tiles), (b) 256x256 problems with private
operands (NO panel is shared: every tile streams its own 2 x M x 256 from HBM), (c) 256x256 problems that ALL read the same
two panels (every staged byte is an L2 hit after the first).   python tools/exp/tn_sharing.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
M = int(sys.argv[1]) if len(sys.argv) > 1 else 31162
g = torch.Generator(device="cuda").manual_seed(0)


def timed(probs, iters=20):
    def run():
        for i in range(0, len(probs), 40):
            lib.gemm_tn_grouped(probs[i:i + 40], shared=0)
    for _ in range(3): run()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): run()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3


def report(name, probs, tiles):
    t = timed(probs)
    steps = (M + 63) // 64
    rounds = (tiles + 255) // 256
    print(f"{name:58s} {tiles:5d} tiles  {t*1e3:8.3f} ms  {2.0*M*256*256*tiles/t/1e12:7.1f} TFLOP/s  {t/rounds/steps*1e6:6.2f} us per 256x256x64 step")


# (a) 7 text blocks: 4 problems each, 36+9+36+36 = 117 tiles per block -> 819 tiles
probs = []
for _ in range(7):
    for (N, K) in [(2304, 768), (768, 768), (3072, 768), (768, 3072)]:
        dy = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
        x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
        probs.append((dy, x, torch.zeros((N, K), device="cuda"), None, None, None, None))
report("(a) text-block shapes, panels shared by 3-12 tiles", probs, 7 * 117)
del probs
# (b) 768 private 256x256 problems (3 rounds)
T = 768
dyb = torch.randn((M, 256 * 96), generator=g, device="cuda").to(torch.bfloat16)      # 96 distinct dY panels
xb = torch.randn((M, 256 * 96), generator=g, device="cuda").to(torch.bfloat16)
dws = [torch.zeros((256, 256), device="cuda") for _ in range(T)]
probs = [(dyb[:, 256 * (i % 96):256 * (i % 96) + 256], xb[:, 256 * ((i * 7) % 96):256 * ((i * 7) % 96) + 256], dws[i], None, None, None, None) for i in range(T)]
report("(b) 256x256 problems, 96 + 96 panels (each shared by 8 tiles far apart)", probs, T)
probs = [(dyb[:, 256 * (i % 96):256 * (i % 96) + 256], xb[:, 256 * (i % 96):256 * (i % 96) + 256], dws[i], None, None, None, None) for i in range(96)] * 8
report("(b') the same 96 private pairs, 8 x in sequence", probs, T)
probs = [(dyb[:, :256], xb[:, :256], dws[i], None, None, None, None) for i in range(T)]
report("(c) 256x256 problems that all read the SAME two panels", probs, T)
