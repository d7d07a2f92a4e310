"""A reproducer for the segfault in hip::Graph::UpdateStreams (ROCm 7.0 runtime shipped with torch 2.10) that the step
executor ran into (tests/test_gpu_graphs.py::test_graph_entries_are_evicted_and_recaptured, 1 run in ~8).

hipGraphInstantiate gives an exec with N parallel branches N internal streams; the FIRST hipGraphLaunch picks N - 1 of them
whose hardware queue differs from the launch stream's, walking the list WITHOUT a bound -- when two of the N share the launch
stream's queue it reads past the end.  Streams are dealt to the least-used of 4 hardware queues (per priority class), so two
new streams land on one queue only after destroyed execs have left the queues unevenly loaded.

argv[1]: "default" = first launch on the current (normal-priority) stream; "prio" = on a high-priority stream, whose queue
comes from another pool (what unimm_amd/graphs.py does).  Small graphs with two parallel branches are created, launched and
destroyed at random; with the default 4 queues that alone did not trip it in 400 rounds, so make it certain:

    GPU_MAX_HW_QUEUES=1 python tools/exp/hip_graph_stream_alias.py default    -> SIGSEGV at the first launch
    GPU_MAX_HW_QUEUES=1 python tools/exp/hip_graph_stream_alias.py prio       -> runs through

(measured on MI355X, gpurun_out/r5y; the native backtrace of the executor's own fault: hip::Graph::UpdateStreams <-
hip::GraphExec::Run <- hipGraphLaunch <- at::cuda::CUDAGraph::replay, gpurun_out/r5t/gdb.log)."""
import faulthandler, random, sys
import torch
faulthandler.enable()
mode = sys.argv[1] if len(sys.argv) > 1 else "default"
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
dev = torch.device("cuda", 0)
cap = torch.cuda.Stream(dev)
sides = [torch.cuda.Stream(dev) for _ in range(2)]
def cumask_stream():
    """A normal-priority stream with a (full) CU mask: such streams get a hardware queue of their own, outside the pool."""
    import ctypes, os
    hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
    n = torch.cuda.get_device_properties(dev).multi_processor_count
    words = (n + 31) // 32
    mask = (ctypes.c_uint32 * words)(*([0xFFFFFFFF] * words))
    if n % 32:
        mask[words - 1] = (1 << (n % 32)) - 1
    h = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(h), ctypes.c_uint32(words), mask)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(h.value, device=dev)


launch = torch.cuda.Stream(dev, priority=-1) if mode == "prio" else cumask_stream() if mode == "cumask" else None
xs = [torch.zeros(1024, device=dev) for _ in range(3)]


def make():
    g = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=cap):
        xs[0].add_(1)
        for s, x in zip(sides, xs[1:]):
            s.wait_stream(cap)
            with torch.cuda.stream(s):
                x.add_(1)
        for s in sides:
            cap.wait_stream(s)
        xs[0].add_(xs[1]).add_(xs[2])
    return g


def first_launch(g):
    if launch is None:
        g.replay()
    else:
        launch.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(launch):
            g.replay()
        torch.cuda.current_stream().wait_stream(launch)


random.seed(0)
live = []
for it in range(iters):
    for _ in range(random.randint(1, 3)):
        g = make()
        first_launch(g)
        live.append(g)
    random.shuffle(live)
    for _ in range(random.randint(0, min(3, len(live) - 1))):
        live.pop()
    if len(live) > 12:
        del live[:6]
    if it % 50 == 0:
        torch.cuda.synchronize()
        print(f"{mode}: iteration {it}, {len(live)} execs alive", flush=True)
torch.cuda.synchronize()
print(f"{mode}: {iters} iterations without a fault", flush=True)
