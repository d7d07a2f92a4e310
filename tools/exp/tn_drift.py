"""How far do the workgroups of a grouped weight-gradient launch drift apart?  (needs the trace variant of the library:
bash tools/exp/variant_lib.sh tn_trace gemm.hip -DUNIMM_TN_TRACE; UNIMM_HIP_LIB=unimm_amd/_ab/tn_trace.so python tools/exp/tn_drift.py)

Tiles that share an operand panel (the nbk tiles of a tile row share dY, every nbn-th tile shares X) are dispatched next to
each other on ONE XCD; they hit in that XCD's 4 MiB L2 only while they walk the reduction rows within a few steps of each
other (one 64-row step of the ~16 panels of a 32-tile window is ~0.5 MB).  Every workgroup stamps the wall clock at steps
0, 32, 64, ...; per XCD and dispatch round this prints the spread of those stamps in units of the step time."""
import ctypes as C
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from unimm_amd import lib

M = int(sys.argv[1]) if len(sys.argv) > 1 else 31162
NBLOCKS = int(os.environ.get("TN_BLOCKS", "9"))
g = torch.Generator(device="cuda").manual_seed(0)
probs, tiles = [], 0
for _ in range(NBLOCKS):
    for (N, K) in [(768, 3072), (3072, 768), (768, 768), (2304, 768)]:
        dy = torch.randn((M, N), generator=g, device="cuda").to(torch.bfloat16)
        x = torch.randn((M, K), generator=g, device="cuda").to(torch.bfloat16)
        probs.append((dy, x, torch.zeros((N, K), device="cuda"), None, None, None, torch.zeros(N, device="cuda"), None, True))
        tiles += (N // 256) * (K // 256)
L = lib.lib()
trace = torch.zeros(tiles * 32, dtype=torch.int64, device="cuda")
for _ in range(2):
    lib.gemm_tn_grouped(probs, shared=0)
torch.cuda.synchronize()
assert L.unimm_debug_tn_trace(C.c_void_p(trace.data_ptr())) == 0
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
lib.gemm_tn_grouped(probs, shared=0)
e.record()
torch.cuda.synchronize()
L.unimm_debug_tn_trace(C.c_void_p(0))
if os.environ.get("TN_DUMP"): np.save(os.environ["TN_DUMP"], trace.cpu().numpy())
ms = s.elapsed_time(e)
tr = trace.cpu().numpy().reshape(tiles, 32)
xcc = tr[:, 30] & 15
stamps = tr[:, :30].astype(np.float64)
nsteps = (M + 63) // 64
nst = min(30, (nsteps + 31) // 32)
stamps = stamps[:, :nst]
t0 = stamps[stamps > 0].min()
us = (stamps - t0) / 100.0                         # wall_clock64: 100 MHz
print(f"{tiles} tiles ({NBLOCKS} text blocks, M = {M}), launch {ms*1e3:.0f} us, {2.0*M*256*256*tiles/ms/1e9:.0f} TFLOP/s")
print("XCD of blockIdx b: b % 8 ==", "yes" if all(int(xcc[b]) == int(xcc[b % 8]) for b in range(tiles)) else "NO", " ids", [int(xcc[b]) for b in range(8)])
dur = us[:, nst - 1] - us[:, 0]
step_us = float(np.median(dur)) / (32 * (nst - 1))
print(f"median step time {step_us:.2f} us; workgroup duration (first to last stamp) min / median / max {dur.min():.0f} / {np.median(dur):.0f} / {dur.max():.0f} us")
for x in range(8):
    ids = np.arange(x, tiles, 8)                     # this XCD's workgroups in dispatch order
    print(f"XCD {x}:")
    for r in range(0, len(ids), 32):
        w = ids[r:r + 32]
        st = us[w]
        spread = (st.max(axis=0) - st.min(axis=0)) / step_us      # in steps, per stamp
        d = st[:, -1] - st[:, 0]
        print(f"   workgroups {r:3d}..{r+len(w)-1:3d}: duration mean {d.mean():6.0f} min {d.min():6.0f} max {d.max():6.0f} us; start spread {st[:,0].max()-st[:,0].min():7.1f} us; spread at steps 0,32,..: " +
              " ".join(f"{v:5.1f}" for v in spread) + "  steps")
