"""The training step as replayed hipGraphs (the executor of the small-batch regime).

Eagerly a step is ~650 Python -> ctypes -> hipLaunchKernelGGL calls plus ~750 tensor allocations: ~10 ms of host work, as
much as the GPU needs for the 30 sequences a GPU holds when 240 are split over 8 (the strong-scaling
ceiling).  Here the launch sequence of `Engine._forward` + `_losses` and of `Engine._backward` is captured ONCE per
signature (`torch.cuda.graph`: our C-ABI launches go to torch's capturing stream, the image side forks and joins through
events, every `torch.empty` of the step comes from the graph's private pool = a static activation arena) and replayed with
two `hipGraphLaunch` calls per step.  What a replay cannot take from frozen launch arguments lives in device memory:

  * row counts: launches are sized for CAPACITIES (valid text rows / decoded rows rounded up to `row_bucket` /
    `lm_bucket`), kernels that reduce over rows read the real counts written by unimm_plan_build (the step's own device words, `out["dyn"]`);
  * loss denominators (1 / decoded rows, 1 / masked regions): the same words;
  * dropout: the launch argument is the per-(seed, site) key, the per-step salt is one device word (Engine.salt_word);
  * inputs: copied into static buffers before the replay; gradients of the three losses likewise.

The host still reads ONE header per step (valid lengths, decoded rows: `Engine.count_rows`) to pick the bucket; a
signature is captured the second time it is seen (the first time runs eagerly) and at most `max_entries` signatures are
kept (each owns its activations: ~0.19 GB per sequence at the full config).

Data parallelism: the gradient exchange is issued per bucket from INSIDE backward (Engine._bucket_done -> the wrapper's
hook -> a collective on the comm stream), i.e. between launches.  With a hook registered the backward is therefore captured
as a CHAIN of graphs cut at the points where buckets are handed over (~11 per step, since the weight-gradient launches are
grouped over blocks): a replayed step is `graph, hook(s), graph, hook(s), ...` in capture order -- the collectives stay
ordinary eager calls of whatever backend the wrapper uses.  At a cut both streams are joined (the eager path joins them
there too: the exchange reads what the image side produced) and the image stream is forked back in at the start of the
next segment.

Not covered (falls back to the eager path): mask descriptors (`DialogMaskSpec`), dense LM scores (`output_lm_scores=True`), inputs
left on the host (`engine.host_staging = False`; with the default staging ring, CPU tensors handed to `forward()` are device
tensors -- their masks bit-packed words -- by the time the executor sees them)."""
from __future__ import annotations

import contextlib
import gc
import weakref
from collections import OrderedDict

import torch

from . import dropout as DR
from . import lib as L
from .inputs import PackedMask

_TENSOR_KEYS = ("input_ids", "image_feat", "image_loc", "token_type_ids", "position_ids", "attention_mask",
                "image_attention_mask", "co_attention_mask", "masked_lm_labels", "image_label", "image_target",
                "next_sentence_label", "lm_weight", "image_index")


def _rup(x, m):
    return (x + m - 1) // m * m


# ------------------------------------------------------------------------------------------------
# Graph execs are never destroyed; their memory pools are recycled.
#
# A hipGraph exec with parallel branches (ours: text side / image side) owns internal streams, created at instantiation and
# dealt to the least-used of the 4 hardware queues of their priority class.  On EVERY hipGraphLaunch the runtime shipped
# with torch 2.10 (hip::Graph::UpdateStreams) picks those whose queue differs from the launch stream's -- walking the list
# without a bound: when ALL of an exec's internal streams share the launch stream's queue it reads past the end (SIGSEGV;
# native backtrace and a reproducer: tools/exp/hip_graph_stream_alias.py).  While streams are only ever created, the deal
# is round-robin and two streams created back to back never share a queue; once streams have been DESTROYED (= execs
# destroyed: evicted entries, dropped models) the queues are unevenly loaded and they can -- which is how the GPU suite
# died once in ~8 runs.  Launching from a stream of another priority class (or with a CU mask) is immune but costs 5 ms of
# a 9 ms step (StepGraphs(launch="isolated")).  So: every CUDAGraph this module creates stays referenced here for the life
# of the process (an exec is ~1 MB of host memory), and what would otherwise be lost with it -- the private memory pool
# holding an entry's activations -- goes back to a free list when the entry dies, together with the stream the entry was captured
# on, and is captured into again by the next entry (the dead entry's graphs are never launched again, so sharing their pool is
# harmless).  tools/soak_graph_churn.py: 300 steps of five signatures evicting each other + a second model coming and going =
# 204 captures, 236 execs alive at the end, reserved memory level at 14-19 GB (profiles/r5w_graph_churn_soak.txt).
# ------------------------------------------------------------------------------------------------
_KEPT = []                      # every graph exec ever instantiated here
_FREE_POOLS = {}                # device index -> [(signature, side stream, pool handle, capture stream)] of dead entries, oldest first


def _new_graph():
    g = torch.cuda.CUDAGraph()
    _KEPT.append(g)
    if len(_KEPT) in (1024, 8192, 65536):
        import warnings
        warnings.warn(f"unimm_amd.graphs: {len(_KEPT)} graph execs captured so far (none is ever destroyed, ~1 MB of host memory "
                      "each): signatures keep being evicted and captured again -- raise max_entries or widen the row buckets")
    return g


def _dev_index(dev):
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _take_pool(dev, sig, side):
    """(pool, capture stream) for a new entry.  The caching allocator hands a freed block only to allocations on the stream it was
    allocated on, so a pool is recycled TOGETHER WITH the stream its entry was captured on (a fresh stream per entry made every
    capture reserve its memory anew: 17 GB per pool after 100 captures of 2 GB entries, tools/soak_graph_churn.py); the image
    side's blocks belong to the engine's side stream, so pools of the same engine are preferred -- and among those the one a dead
    entry of the same signature left behind (that capture allocates what the old one did, in the same order)."""
    free = _FREE_POOLS.get(_dev_index(dev))
    best = -1
    if free:
        for i in range(len(free) - 1, -1, -1):
            if free[i][1] == side:
                if free[i][0] == sig:
                    best = i
                    break
                if best < 0:
                    best = i
        if best < 0:
            best = len(free) - 1
        _, _, pool, stream = free.pop(best)
        return pool, stream
    return torch.cuda.graph_pool_handle(), torch.cuda.Stream(device=dev)


def _give_pool(index, sig, side, pool, stream):
    _FREE_POOLS.setdefault(index, []).append((sig, side, pool, stream))


def release_all():
    """Opt-in, for a process that will NEVER launch a captured step again (e.g. before an evaluation phase that needs the
    memory): drop every graph exec and every recycled pool this module holds, so that the caching allocator can return the
    private pools (the activations of dead entries: 14-19 GB under churn).  Live executors must be dropped by their owners
    first (`engine.enable_graphs(False)`); destroying execs is exactly what the keep-alive list above exists to avoid (the
    runtime fault it documents needs a LATER graph launch to fire), hence: only when no graph will be launched afterwards.
    -> number of execs released."""
    n = len(_KEPT)
    _KEPT.clear()
    _FREE_POOLS.clear()
    gc.collect()
    if torch.cuda.is_available():
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
    return n


@contextlib.contextmanager
def _quiet_collector():
    """No cyclic garbage collection while a stream is capturing.  The collector runs wherever an allocation count trips it;
    if that is in the middle of a capture and the garbage holds another executor's graphs (a dropped model: engine <->
    executor <-> entries is a cycle), their destructors synchronise the device and return pool memory -- inside the capture.
    Collect first (outside), then hold the collector off until the capture has ended."""
    gc.collect()
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class _Entry:
    __slots__ = ("pool", "sin", "gF", "out", "losses", "nsp", "gB", "gin", "gkey", "stream", "lvec", "lshape", "inflight", "salt_val",
                 "__weakref__")


class _Token:
    """Lives on the autograd context of a replayed forward until its backward ran (or the autograd graph was dropped): while
    it is alive the entry's static inputs and activations belong to that step and must not be overwritten."""
    __slots__ = ("__weakref__",)


class StepGraphs:
    def __init__(self, engine, row_bucket=128, lm_bucket=64, max_entries=4, capture_after=1, launch="caller"):
        self.eng = engine
        self.row_bucket, self.lm_bucket = row_bucket, lm_bucket
        self.max_entries, self.capture_after = max_entries, capture_after
        self.entries = OrderedDict()
        self.seen = {}
        self.salt = None
        self._salt_val = None
        self.stats = dict(replays=0, captures=0, eager=0, busy=0)
        if launch not in ("caller", "isolated"):
            raise ValueError("launch: 'caller' or 'isolated'")
        self.launch = launch
        self._launch_stream = None

    def _replay(self, g):
        """hipGraphLaunch on the caller's stream, or (launch="isolated") on a high-priority stream fenced against it on both
        sides: its hardware queue comes from another pool than the execs' internal streams, which makes the runtime fault
        described at the top of this file impossible whatever else the process creates and destroys -- at 14.0 instead of
        9.1 ms per 30-sequence step (a CU-masked or a low-priority launch stream measured the same or worse)."""
        if self.launch == "caller":
            g.replay()
            return
        dev = self.eng.arena.device
        if self._launch_stream is None:
            self._launch_stream = torch.cuda.Stream(device=dev, priority=-1)
        cur = torch.cuda.current_stream(dev)
        self._launch_stream.wait_stream(cur)
        with torch.cuda.stream(self._launch_stream):
            g.replay()
        cur.wait_stream(self._launch_stream)

    # ------------------------------------------------------------------------------------------
    def eligible(self, inp, opts):
        eng = self.eng
        if opts.get("want_seq") or eng.text_priority or eng.cfg.predict_feature:      # (the MSE branch's divisor is a launch argument)
            return False
        for k in _TENSOR_KEYS:
            v = inp.get(k)
            if v is None:
                continue
            if isinstance(v, PackedMask):          # a mask bit-packed on the host side of the copy (inputs.HostStager): its words are the input
                v = v.words
            if not torch.is_tensor(v) or not v.is_cuda:
                return False
        for k in ("attention_mask", "co_attention_mask", "masked_lm_labels", "image_target", "next_sentence_label"):
            if inp.get(k) is None:
                return False
        nw = inp.get("nsp_weight")
        return nw is None or (torch.is_tensor(nw) and not nw.is_cuda)

    @staticmethod
    def _key0(inp, opts):
        parts = [bool(opts["train"])]
        for k in _TENSOR_KEYS:
            v = inp.get(k)
            parts.append(None if v is None else (("packed",) + tuple(v.shape) if isinstance(v, PackedMask) else (tuple(v.shape), v.dtype)))
        nw = inp.get("nsp_weight")
        parts.append(None if nw is None else tuple(float(x) for x in nw.reshape(-1).tolist()))
        return tuple(parts)

    def _set_salt(self, v=None):
        """The per-step dropout salt word.  v = None: the salt of the engine's current step; otherwise the value a replayed
        forward used (its backward regenerates the masks from the same word, whatever ran in between)."""
        eng = self.eng
        if self.salt is None:
            self.salt = torch.zeros(1, dtype=torch.int32, device=eng.arena.device)
        if v is None:
            v = DR.step_salt(eng.seed, eng.step)
        if v != self._salt_val:
            self.salt.fill_(v - (1 << 32) if v >= (1 << 31) else v)
            self._salt_val = v
        return v

    # ------------------------------------------------------------------------------------------
    def forward(self, inp, opts, defer_outputs=False):
        """-> (lm_loss, img_loss, nsp_loss, nsp, entry) from a replay, or None (the caller runs the eager path).
        defer_outputs: -> (None, None, None, None, entry); the caller copies the losses out of the entry's static buffers
        (`outputs(entry)`) after it has enqueued the backward (BertForMultiModalPreTraining.forward_backward)."""
        eng = self.eng
        eng.refresh_weights()
        hh = inp.get("_plan_header")                           # a prefetcher may have read the step's header already (Engine.count_rows)
        B, T = inp["input_ids"].shape
        if hh is None:
            hh = eng.count_rows(inp)                           # the step's one host sync
        else:
            # A caller-supplied header alone sizes the capacities of the replayed launches.  What the host can check, it checks
            # (shape and value ranges); that the header belongs to THIS batch only the device can know: the replayed plan kernel
            # recomputes the counts, stays inside the capacities whatever they are and turns the losses into NaN when the batch
            # does not fit (csrc/rowops.hip: plan_build_kernel).
            hh = [int(x) for x in hh]
            if len(hh) != 3 * B + 2:
                raise ValueError(f"plan_header has {len(hh)} entries, expected 3 * B + 2 = {3 * B + 2} (Engine.count_rows of this batch)")
            if any(not 1 <= x <= T for x in hh[:B]) or any(not 0 <= x <= T for x in hh[B:2 * B]):
                raise ValueError(f"plan_header: valid lengths must lie in [1, {T}] and decoded-row counts in [0, {T}]")
        Mv, n_lm = sum(hh[:B]), sum(hh[B:2 * B])
        Mcap = min(_rup(Mv, self.row_bucket), B * T)
        ncap = _rup(n_lm, self.lm_bucket) if n_lm > 0 else 0
        sig = self._key0(inp, opts) + (Mcap, ncap, eng.grad_bucket_hook is not None) + eng.schedule_key()   # with a hook the backward is a chain of graphs
        ent = self.entries.get(sig)
        if ent is not None and ent.inflight is not None and ent.inflight() is not None:
            # the previous forward of this signature has not been back-propagated yet (two losses summed before .backward(),
            # interleaved micro-batches): its activations live in the entry's static buffers -> this step runs eagerly
            self.stats["busy"] += 1
            self.stats["eager"] += 1
            return None
        if ent is None:
            n = self.seen.get(sig, 0)
            self.seen[sig] = n + 1
            if n < self.capture_after:
                self.stats["eager"] += 1
                return None
            ent = self._capture_forward(sig, inp, opts, hh)
        else:
            self.entries.move_to_end(sig)
            for k, t in ent.sin.items():
                if isinstance(t, PackedMask):
                    t.words.copy_(inp[k].words, non_blocking=True)
                elif torch.is_tensor(t) and t is not inp[k]:   # (a caller that feeds the entry's own static inputs back skips the copy)
                    t.copy_(inp[k], non_blocking=True)
        ent.salt_val = self._set_salt()
        self._replay(ent.gF)
        self.stats["replays"] += 1
        tok = _Token()
        ent.inflight = weakref.ref(tok)
        ent.out["_token"] = tok                                # handed to the autograd context by _HotPath.forward
        if defer_outputs:
            return None, None, None, None, ent
        return self.outputs(ent) + (ent,)

    @staticmethod
    def outputs(ent):
        """(lm_loss, img_loss, nsp_loss, nsp) copied out of the entry's static buffers (the next replay overwrites them)."""
        ls = ent.lvec.clone()
        return ls[0].reshape(ent.lshape), ls[1].reshape(ent.lshape), ls[2].reshape(ent.lshape), ent.nsp.clone()

    def _capture_forward(self, sig, inp, opts, hh):
        eng = self.eng
        dev = eng.arena.device
        while len(self.entries) >= self.max_entries:           # each signature owns its activations
            self.entries.popitem(last=False)
        ent = _Entry()
        side = int(eng._side_stream().cuda_stream) if eng._dual() else 0
        ent.pool, ent.stream = _take_pool(dev, sig, side)
        weakref.finalize(ent, _give_pool, _dev_index(dev), sig, side, ent.pool, ent.stream)   # with its tensors gone the pool is free for the next entry
        ent.sin = {k: (v.clone() if torch.is_tensor(v) and v.is_cuda else
                       (PackedMask(v.words.clone(), v.shape) if isinstance(v, PackedMask) else v)) for k, v in inp.items()}
        ent.gB, ent.gin, ent.gkey = None, None, None
        ent.inflight, ent.salt_val = None, None
        was = (eng.row_bucket, eng.lm_bucket, eng.salt_word, eng._inject_header)
        if self.salt is None:
            self._set_salt()
        eng.row_bucket, eng.lm_bucket, eng.salt_word, eng._inject_header = self.row_bucket, self.lm_bucket, self.salt, hh
        try:
            # warm-up on the capture stream with the capture's own sizes (first-use initialisation of kernels / tiles)
            ent.stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(ent.stream):
                with torch.no_grad():
                    out = eng.forward(ent.sin, train=opts["train"], save=False, lm_rows="labelled", want_pred_v=True)
                    eng.losses(out, ent.sin)
                del out
            torch.cuda.current_stream().wait_stream(ent.stream)
            torch.cuda.synchronize()
            ent.gF = _new_graph()
            with _quiet_collector(), torch.cuda.graph(ent.gF, pool=ent.pool, stream=ent.stream, capture_error_mode="thread_local"):
                ent.out = eng.forward(ent.sin, train=opts["train"], save=True, lm_rows="labelled", want_pred_v=True)
                ent.losses = eng.losses(ent.out, ent.sin)
                ent.nsp = ent.out["nsp"]
                ls = ent.losses
                ent.lshape = tuple(ls["lm_loss"].shape)
                ent.lvec = torch.stack([ls["lm_loss"].reshape(()), ls["img_loss"].reshape(()), ls["nsp_loss"].reshape(())])
        finally:
            eng.row_bucket, eng.lm_bucket, eng.salt_word, eng._inject_header = was
        self.entries[sig] = ent
        self.stats["captures"] += 1
        return ent

    # ------------------------------------------------------------------------------------------
    def _capture_backward(self, ent):
        """-> the backward as a program: [graph] without a bucket hook, else [graph, (group, more), ..., graph, ...]."""
        eng = self.eng
        prog, cur = [], [None]
        hook = eng.grad_bucket_hook

        def begin():
            g = _new_graph()
            g.capture_begin(pool=ent.pool, capture_error_mode="thread_local")   # other threads (a process group's watchdog) may touch the device
            cur[0] = g

        def end():
            cur[0].capture_end()
            prog.append(cur[0])
            cur[0] = None

        def cut(group, more=False):
            # Engine._bucket_done has joined the image stream into the text stream (the capturing one) just before
            if cur[0] is not None:
                end()
            prog.append((group, more))
            if not more and group != "text_embeddings":       # the last bucket: nothing is enqueued after it
                begin()
                eng._to_img()                                  # the image stream rejoins the capture

        was = (eng.row_bucket, eng.lm_bucket, eng.salt_word)
        eng.row_bucket, eng.lm_bucket, eng.salt_word = self.row_bucket, self.lm_bucket, self.salt
        if hook is not None:
            eng.grad_bucket_hook = cut
        try:
            torch.cuda.synchronize()
            ent.stream.wait_stream(torch.cuda.current_stream())
            with _quiet_collector(), torch.cuda.stream(ent.stream):
                begin()
                try:
                    eng.backward(ent.out, *ent.gin)
                finally:
                    if cur[0] is not None:
                        end()
            torch.cuda.current_stream().wait_stream(ent.stream)
        finally:
            eng.row_bucket, eng.lm_bucket, eng.salt_word = was
            eng.grad_bucket_hook = hook
        if hook is not None and not any(isinstance(i, tuple) and i[0] == "text_embeddings" for i in prog):
            raise RuntimeError("graph capture of backward: the last gradient bucket was never handed over")
        return prog

    def backward(self, ent, g_lm, g_img, g_nsp, g_scores):
        eng = self.eng
        dev = eng.arena.device
        grads = (g_lm, g_img, g_nsp, g_scores)
        eng.arena.attach_grads()        # BEFORE a capture: it zeroes the arena when .grad was dropped, which must not be replayed
        shapes = tuple(None if g is None else tuple(g.shape) for g in grads)
        fresh = bool(eng.arena.fresh)   # write-vs-add of the weight gradients is frozen into the launches: one program per state
        self._set_salt(ent.salt_val)    # the masks of THIS step's forward (another replay may have re-salted since)
        if ent.gB is None or ent.gkey != shapes:
            ent.gkey = shapes
            ent.gin = [None if g is None else torch.zeros(g.shape, dtype=torch.float32, device=dev) for g in grads]
            ent.gB = {}
        prog = ent.gB.get(fresh)
        if prog is None:
            for s, g in zip(ent.gin, grads):
                if s is not None:
                    s.copy_(g.detach().to(torch.float32))
            prog = ent.gB[fresh] = self._capture_backward(ent)
            self.stats["captures"] += 1
        else:
            pairs = [(s, g.detach()) for s, g in zip(ent.gin, grads) if s is not None]
            if all(g.dtype == torch.float32 and g.shape == s.shape for s, g in pairs):
                torch._foreach_copy_([s for s, _ in pairs], [g for _, g in pairs])     # one launch
            else:
                for s, g in pairs:
                    s.copy_(g, non_blocking=True)
        hook = eng.grad_bucket_hook
        for item in prog:                                      # graph segments and, between them, the bucket hand-overs
            if isinstance(item, tuple):
                if hook is not None:
                    hook(*item)
            else:
                self._replay(item)
        eng.arena.fresh = False                                # (a replay does not run Engine._backward, which clears it eagerly)
