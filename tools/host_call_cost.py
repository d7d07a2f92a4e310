"""Host-side cost per call of the bindings (enqueue only, tiny problem sizes, queue kept short)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import lib as L

dev = "cuda"
x = torch.randn(256, 768, device=dev).bfloat16()
w = torch.randn(768, 768, device=dev).bfloat16()
o = torch.empty(256, 768, device=dev).bfloat16()
b = torch.randn(768, device=dev)
x32 = torch.randn(256, 768, device=dev)
g = torch.ones(768, device=dev)
y32 = torch.empty_like(x32)
y16 = torch.empty_like(o)
mean = torch.empty(256, device=dev)
rstd = torch.empty(256, device=dev)


def t(fn, n=2000):
    for _ in range(50):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(n):
        fn()
        if i % 200 == 199:
            torch.cuda.synchronize()      # keep the queue from filling (that would measure the GPU, not the host)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


print("lib.unimm_version()        %.2f us" % t(lambda: L.lib().unimm_version()))
print("torch.empty                %.2f us" % t(lambda: torch.empty((256, 768), dtype=torch.bfloat16, device=dev)))
print("torch.cuda.current_stream  %.2f us" % t(lambda: torch.cuda.current_stream().cuda_stream))
print("gemm_nt                    %.2f us" % t(lambda: L.gemm_nt(x, w, o, bias=b)))
with L.stream_scope(torch.cuda.current_stream()):
    print("gemm_nt (scoped stream)    %.2f us" % t(lambda: L.gemm_nt(x, w, o, bias=b)))
    print("layernorm_fwd (scoped)     %.2f us" % t(lambda: L.layernorm_fwd(x32, g, b, y32, y16, mean, rstd, 256, 768)))
print("torch add_ (eager op)      %.2f us" % t(lambda: y32.add_(1.0)))
ev = torch.cuda.Event()
print("event record               %.2f us" % t(lambda: ev.record()))
