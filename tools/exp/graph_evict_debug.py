"""Where did tests/test_gpu_graphs.py::test_graph_entries_are_evicted_and_recaptured die?  argv: evict | all | all-nogc

The investigation tool behind the note at the top of unimm_amd/graphs.py: runs the graph tests in the suite's order with a
line per capture / replay, under switches that move the fault around (HIST = which earlier tests run, COLLECT = where
gc.collect() runs, DUAL=0 = single-stream graphs, BURN = pool streams taken first, SHARED=1 = one capture stream).  With
graph execs destroyed (the state before that note) `all` died at the third step's backward replay -- under rocgdb in
hip::Graph::UpdateStreams; since execs are kept alive every mode runs through."""
import contextlib, faulthandler, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
faulthandler.enable()
import torch
import unimm_amd.graphs as G
import test_gpu_graphs as T
mode = sys.argv[1] if len(sys.argv) > 1 else "all"
gd = os.path.join(ROOT, "tests", "golden")
def say(*a):
    print(*a, file=sys.stderr, flush=True)
if mode == "all-nogc":
    G._quiet_collector = contextlib.nullcontext
HIST = os.environ.get("HIST", "bf16,fp32x3").split(",")
COLLECT = os.environ.get("COLLECT", "both")          # both | fwd | none: where gc.collect() runs (the collector is held off in all)
if COLLECT != "both":
    import gc, threading
    @contextlib.contextmanager
    def qc():
        if COLLECT == "fwd" and threading.current_thread() is threading.main_thread():
            gc.collect()
        was = gc.isenabled(); gc.disable()
        try:
            yield
        finally:
            if was: gc.enable()
    G._quiet_collector = qc
if int(os.environ.get("BURN", "0")):
    burned = [torch.cuda.Stream() for _ in range(int(os.environ["BURN"]))]
    say("burned", len(burned), "pool streams")
    if os.environ.get("BURN_DROP") == "1":
        del burned
if os.environ.get("SHARED") == "1":
    _one = {}
    class _T:
        def __getattr__(self, n):
            return getattr(torch, n)
    class _C:
        def __getattr__(self, n):
            return getattr(torch.cuda, n)
        @staticmethod
        def Stream(device=None, **k):
            if "s" not in _one:
                _one["s"] = torch.cuda.Stream(device=device, **k)
            return _one["s"]
    t = _T(); t.__dict__["cuda"] = _C()
    G.torch = t
if os.environ.get("DUAL") == "0":
    import unimm_amd.engine as E
    oi = E.Engine.__init__
    def init(self, *a, **k):
        oi(self, *a, **k); self.dual_stream = False
    E.Engine.__init__ = init
for name in ("_capture_forward", "_capture_backward", "backward"):
    orig = getattr(G.StepGraphs, name)
    def wrap(self, *a, _o=orig, _n=name, **k):
        say(f"  > {_n} entries={len(self.entries)} stats={self.stats}")
        r = _o(self, *a, **k)
        e = self.eng
        hs = lambda st: None if st is None else hex(st.cuda_stream)
        ents = [hs(v.stream) for v in self.entries.values()]
        say(f"  < {_n}  entry streams {ents}  image-side {hs(e._vside)}  wgrad-side {hs(getattr(e, '_side', None))}  current {hs(torch.cuda.current_stream())}")
        return r
    setattr(G.StepGraphs, name, wrap)
if mode != "evict":
    for compute in HIST:
        for train in (True, False):
            say(f"replay_equals {compute} {train}")
            T.test_graph_replay_equals_eager_steps(gd, train, compute)
    say("fallback")
    T.test_graph_executor_falls_back_when_not_eligible(gd)
say("evict")
T.test_graph_entries_are_evicted_and_recaptured(gd)
say("done", mode)
