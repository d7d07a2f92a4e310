"""Can torch.cuda.graph capture the C-ABI launches (ctypes -> hipLaunchKernelGGL on torch's current stream), with the
engine's two-stream fork / join pattern and allocations inside the capture?  (probe for the step executor; tools only)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib as L
dev = torch.device("cuda", 0)
g = torch.Generator(device="cuda").manual_seed(0)
M, N, K = 3900, 768, 768
x = torch.randn((M, K), generator=g, device=dev).to(torch.bfloat16)
w = (torch.randn((N, K), generator=g, device=dev) * 0.05).to(torch.bfloat16)
b = torch.randn(N, device=dev)
side = torch.cuda.Stream(device=dev)

def body():
    outs = []
    y = x
    for i in range(6):
        o = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        L.gemm_nt(y, w, o, bias=b)
        outs.append(o)
        y = o
    # fork: side stream computes from outs[2]; join before the end
    ev = torch.cuda.Event(); ev.record(torch.cuda.current_stream()); side.wait_event(ev)
    with torch.cuda.stream(side), L.stream_scope(side):
        s = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
        L.gemm_nt(outs[2], w, s, bias=b)
    ev2 = torch.cuda.Event(); ev2.record(side); torch.cuda.current_stream().wait_event(ev2)
    z = torch.empty((M, N), dtype=torch.bfloat16, device=dev)
    L.gemm_nt(s, w, z, bias=b, epilogue=L.EPI_ADD, aux=y)
    return z

ref = body(); torch.cuda.synchronize()
ref = ref.clone()
cs = torch.cuda.Stream(device=dev)
cs.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(cs):
    body(); body()
torch.cuda.current_stream().wait_stream(cs)
torch.cuda.synchronize()
graph = torch.cuda.CUDAGraph()
try:
    with torch.cuda.graph(graph, stream=cs):
        with L.stream_scope(torch.cuda.current_stream()):
            out = body()
    print("capture ok")
except Exception as e:
    print("capture FAILED:", type(e).__name__, e)
    sys.exit(0)
x.mul_(1.0)   # inputs unchanged: replay must reproduce ref
graph.replay(); torch.cuda.synchronize()
print("replay equal to eager:", torch.equal(out, ref))
t0 = time.perf_counter()
for _ in range(200): graph.replay()
torch.cuda.synchronize()
tg = (time.perf_counter() - t0) / 200
t0 = time.perf_counter()
for _ in range(200): body()
torch.cuda.synchronize()
te = (time.perf_counter() - t0) / 200
print(f"8 GEMMs: graph replay {tg*1e6:.1f} us, eager {te*1e6:.1f} us")
