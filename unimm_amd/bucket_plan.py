"""The gradient-exchange plan of one backward pass, as host arithmetic.

Under data parallelism (`parallel.DataParallelRCCL`, the replacement of utils/data_parallel.py:91-132) the engine hands a
gradient bucket to the wrapper once every weight-gradient launch that covers it has been enqueued, and the engine launches
weight gradients GROUPED over several encoder blocks (`Engine._flush_due`).  Which buckets travel together, in how many
collectives and of what size is therefore a function of the config, the step's row counts and `wgrad_group_rounds` alone.
This module restates that function without a device: `Engine` takes its flush rule from here (so the two cannot drift),
`tests/test_bucket_plan_cpu.py` checks the N = 8 plan of the full config on the CPU, and `tests/test_gpu_dp2.py` holds the
plan against the hand-overs a real backward produces.

Covers the default encoder options (every layer trained, connection layers on: what bert_base_6layer_6conect.json and every script of
the reference use); under `fixed_t_layer` / `with_coattention=False` the engine reports the buckets of skipped blocks at the end of backward.

Nothing here touches torch."""
from __future__ import annotations

from typing import List, Tuple

from . import params as PM

ARENA_ALIGN = 64          # elements (arena.ALIGN)
CHIP_SLOTS = 256          # 256x256 weight-gradient tiles resident at once (one per CU)


def big_tiles_of(M: int, N: int, K: int) -> int:
    """256x256 output tiles of one weight gradient dW[N, K] = dY[M, N]^T X[M, K] when it takes the big tile
    (csrc/gemm.hip: tn_is_big), else 0."""
    if N >= 256 and K >= 256 and M >= 1024:
        return ((N + 255) // 256) * ((K + 255) // 256)
    return 0


def flush_due(n_problems: int, tiles: int, rounds: int) -> bool:
    """Is a queue of `n_problems` weight gradients with `tiles` big tiles due for its grouped launch?  Due = the tile count
    sits just below a multiple of 256 (<= 12.5 % of the last round idle) and covers at least half of `rounds`, or exceeds
    `rounds` rounds of the chip, or the launch's descriptor table (48 problems) is nearly full."""
    if n_problems >= 40:
        return True
    if tiles >= rounds * CHIP_SLOTS:
        return True
    waste = (-tiles) % CHIP_SLOTS
    return tiles >= max(2, rounds // 2) * CHIP_SLOTS - 32 and waste <= CHIP_SLOTS // 8


def arena_ranges(cfg) -> List[Tuple[str, int, int]]:
    """(group, lo, hi) element ranges of the flat arena (arena.FlatArena.buckets) without building one."""
    out, off = [], 0
    for gname, items in PM.arena_groups(cfg):
        lo = off
        for _, shape in items:
            n = 1
            for s in shape:
                n *= s
            off += (n + ARENA_ALIGN - 1) // ARENA_ALIGN * ARENA_ALIGN
        out.append((gname, lo, off))
    return out


def _self_block(h, inter):
    return [(h, inter), (inter, h), (h, h), (3 * h, h)]          # backward order: ff2, ff1, attention output, fused QKV


def backward_events(cfg, n_seq: int, text_rows: int, lm_rows: int, regions: int = 37, dual_stream: bool = True,
                    image_head_side: bool = True):
    """The backward pass as the engine enqueues it (Engine._backward): a list of
    ("w", side, M, N, K) weight gradients queued (side 0 = text queue, 1 = image queue) and
    ("b", group, on_side, force, force_img) bucket-done marks, in order.  `dual_stream` does not change the plan: on one stream the
    image side keeps its own queue (Engine._img), so the same launches and hand-overs happen, one after the other."""
    del dual_stream
    H, Hv, Hb = cfg.hidden_size, cfg.v_hidden_size, cfg.bi_hidden_size
    I, Iv = cfg.intermediate_size, cfg.v_intermediate_size
    Mt, Mi = text_rows, n_seq * regions
    img = 1            # the image side keeps its own queue on one stream as on two (Engine._img)
    ev = []
    hs = img if image_head_side else 0
    ev.append(("w", hs, Mi, cfg.v_target_size, Hv))              # image head decoder, transform
    ev.append(("w", hs, Mi, Hv, Hv))
    if lm_rows > 0:
        ev.append(("w", 0, lm_rows, cfg.vocab_size, H))          # tied decoder, MLM transform
        ev.append(("w", 0, lm_rows, H, H))
    ev.append(("b", "heads", False, True, False))
    sched = PM.encoder_schedule(cfg)
    entries = list(reversed(sched))
    pos = 0
    tail_done = False

    def image_tail():          # Engine._backward_encoder: the image embedding as soon as no image / connection layer is left
        nonlocal tail_done
        if tail_done or any(k in ("c", "v") for k, _ in entries[pos:]):
            return
        tail_done = True
        ev.extend([("w", img, Mi, Hv, cfg.v_feature_size), ("w", img, Mi, Hv, 5)])
        ev.append(("b", "image_embeddings", bool(img), False, True))

    while pos < len(entries):
        image_tail()
        seg = []
        while pos < len(entries) and entries[pos][0] != "c":
            seg.append(entries[pos])
            pos += 1
        for kind, i in seg:
            if kind == "v":
                ev += [("w", img, Mi, n, k) for n, k in _self_block(Hv, Iv)]
                ev.append(("b", f"v{i}", bool(img), False, False))
        for kind, i in seg:
            if kind == "t":
                ev += [("w", 0, Mt, n, k) for n, k in _self_block(H, I)]
                ev.append(("b", f"t{i}", False, False, False))
        if pos < len(entries):
            _, i = entries[pos]
            pos += 1
            ev += [("w", img, Mi, Hv, Iv), ("w", img, Mi, Iv, Hv), ("w", img, Mi, Hv, Hb)]
            ev += [("w", 0, Mt, H, I), ("w", 0, Mt, I, H), ("w", 0, Mt, H, Hb)]
            ev += [("w", img, Mi, 3 * Hb, Hv), ("w", 0, Mt, 3 * Hb, H)]
            ev.append(("b", f"c{i}", False, False, False))
    image_tail()
    ev.append(("b", "text_embeddings", False, True, False))
    return ev


def hand_overs(cfg, n_seq, text_rows, lm_rows, regions=37, wgrad_group_rounds=2, dual_stream=True, image_head_side=True):
    """-> ([(group, more), ...] in the order `Engine._bucket_done` calls the data-parallel hook,
           [number of weight-gradient problems per grouped launch, per queue: (side, n_problems, big_tiles), ...])."""
    q = [[], []]
    nq, nf = [0, 0], [0, 0]
    pending, calls, launches = [], [], []

    def flush(side):
        if q[side]:
            launches.append((side, len(q[side]), sum(big_tiles_of(*p) for p in q[side])))
        q[side] = []
        nf[side] = nq[side]

    def due(side):
        return flush_due(len(q[side]), sum(big_tiles_of(*p) for p in q[side]), wgrad_group_rounds)

    for e in backward_events(cfg, n_seq, text_rows, lm_rows, regions, dual_stream, image_head_side):
        if e[0] == "w":
            _, side, M, N, K = e
            q[side].append((M, N, K))
            nq[side] += 1
            continue
        _, group, on_side, force, force_img = e
        pending.append((group, nq[0], nq[1]))
        if q[1] and (force or force_img or due(1)):   # Engine._flush_wgrad: the image queue first ...
            flush(1)
        if not on_side and (force or due(0)):         # ... the text queue only from the text side
            flush(0)
        if on_side:
            continue
        ready = [p for p in pending if nf[0] >= p[1] and nf[1] >= p[2]]   # Engine._bucket_done: whatever is complete, oldest first
        pending = [p for p in pending if p not in ready]
        calls += [(g, j + 1 < len(ready)) for j, (g, _, _) in enumerate(ready)]
    assert not pending, pending
    return calls, launches


def collectives(cfg, calls):
    """The (lo, hi, n_buckets) element ranges `DataParallelRCCL._on_bucket` exchanges for that sequence of hook calls:
    adjacent slices handed over in one run travel as ONE collective."""
    ranges = {g: (lo, hi) for g, lo, hi in arena_ranges(cfg)}
    out, run = [], None
    for group, more in calls:
        lo, hi = ranges[group]
        if run and (run[0] == hi or run[1] == lo):
            run = [min(lo, run[0]), max(hi, run[1]), run[2] + 1]
        else:
            if run:
                out.append(tuple(run))
            run = [lo, hi, 1]
        if not more:
            out.append(tuple(run))
            run = None
    assert run is None
    return out


def exchange_plan(cfg, n_seq, text_rows, lm_rows, **kw):
    """-> dict(collectives=[(lo, hi, n_buckets)], bytes=[fp32 bytes per collective], launches=[...], calls=[...])."""
    calls, launches = hand_overs(cfg, n_seq, text_rows, lm_rows, **kw)
    cs = collectives(cfg, calls)
    return dict(calls=calls, launches=launches, collectives=cs, bytes=[(hi - lo) * 4 for lo, hi, _ in cs])
