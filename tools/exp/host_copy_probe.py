"""Host-side staging costs on the GPU box: cgroup CPU quota, thread counts, threaded memcpy pageable -> pinned and the mask packer."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from unimm_amd import lib
print("cpu_count", os.cpu_count(), "affinity", len(os.sched_getaffinity(0)), "torch threads", torch.get_num_threads())
for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpu/cpu.cfs_period_us", "/sys/fs/cgroup/cpu.stat"):
    try:
        print(f, open(f).read().strip().replace("\n", " | "))
    except Exception as e:
        print(f, "n/a")
a = torch.randn(240, 37, 2048)
p = torch.empty(a.shape, pin_memory=True)
m = (torch.rand(240, 256, 256) < 0.5).to(torch.int64)
w = torch.empty((240, 256, 8), dtype=torch.int32, pin_memory=True)
for rep in range(2):
    for th in (1, 2, 4, 8, 16):
        ts = []
        for _ in range(6):
            t0 = time.perf_counter(); lib.host_copy(p, a, threads=th); ts.append((time.perf_counter() - t0) * 1e3)
        tp = []
        for _ in range(6):
            t0 = time.perf_counter(); lib.host_mask_pack(m, out=w, threads=th); tp.append((time.perf_counter() - t0) * 1e3)
        print(f"threads {th:2d}: copy 73 MB -> pinned {min(ts):6.2f} .. {max(ts):6.2f} ms   pack 126 MB int64 {min(tp):6.2f} .. {max(tp):6.2f} ms")
t0 = time.perf_counter(); p.copy_(a); print("torch copy_ -> pinned", (time.perf_counter() - t0) * 1e3, "ms")
try:
    print("/sys/fs/cgroup/cpu.stat", open("/sys/fs/cgroup/cpu.stat").read().strip().replace("\n", " | "))
except Exception:
    pass
