"""Multi-GPU readiness without a multi-GPU box: the gradient-exchange plan of the N = 8 split of the
headline batch at the full config, as host arithmetic (`unimm_amd.bucket_plan`, the rule `Engine` itself uses to decide when
a grouped weight-gradient launch is due).  The first real N > 1 run should test RCCL, not bookkeeping: here the plan is
checked to tile the gradient arena exactly once, to hand every bucket over exactly once, and to leave a bounded tail after
the end of backward.  `tests/test_gpu_dp2.py::test_exchange_plan_matches_the_engine` holds the same plan against the
hand-overs of a real backward pass on the device.  Reference: utils/data_parallel.py:91-132, train.py:164-166."""
import os

import pytest

from unimm_amd import BertConfig, bucket_plan as BP

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FULL = os.path.join(ROOT, "unimm_amd", "config", "bert_base_6layer_6conect.json")

# the 8-GPU share of BASELINE configs[2]: 30 sequences, ~3.9k valid text rows, ~630 decoded rows (SURVEY 8d lengths)
SHARE8 = dict(n_seq=30, text_rows=3900, lm_rows=630)


def _cfg():
    return BertConfig.from_json_file(FULL)


def test_arena_ranges_are_the_arena_and_the_reference_parameter_count():
    cfg = _cfg()
    r = BP.arena_ranges(cfg)
    assert r[0][1] == 0 and all(a[2] == b[1] for a, b in zip(r, r[1:]))          # contiguous, in arena order
    from unimm_amd.arena import FlatArena, ALIGN
    assert ALIGN == BP.ARENA_ALIGN
    assert [g for g, _, _ in r] == ["text_embeddings", "image_embeddings"] + [f"t{i}" for i in range(6)] + \
        [x for k in range(6) for x in (f"c{k}", f"v{k}", f"t{6 + k}")] + ["heads"]
    assert 250_090_109 <= r[-1][2] <= 250_090_109 + 64 * 600                      # 250.09 M parameters + alignment padding


@pytest.mark.parametrize("rounds", [2, 4])
@pytest.mark.parametrize("share", [SHARE8, dict(n_seq=60, text_rows=7800, lm_rows=1250), dict(n_seq=240, text_rows=31162, lm_rows=5016)])
def test_exchange_plan_tiles_the_arena_once(rounds, share):
    cfg = _cfg()
    pl = BP.exchange_plan(cfg, wgrad_group_rounds=rounds, **share)
    groups = [g for g, _ in pl["calls"]]
    assert sorted(groups) == sorted(g for g, _, _ in BP.arena_ranges(cfg))        # every bucket exactly once
    assert groups[0] == "heads" and groups[-1] == "text_embeddings" and pl["calls"][-1][1] is False
    spans = sorted((lo, hi) for lo, hi, _ in pl["collectives"])
    assert spans[0][0] == 0 and spans[-1][1] == BP.arena_ranges(cfg)[-1][2]
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))                    # the collectives tile the arena, no overlap
    assert sum(n for _, _, n in pl["collectives"]) == len(groups)
    assert sum(pl["bytes"]) == 4 * BP.arena_ranges(cfg)[-1][2]


def test_the_eight_rank_plan_of_the_headline_batch():
    """DataParallelRCCL sets wgrad_group_rounds = 2 with more than one rank.  14 collectives of 8-118 MB; what is left after
    the last backward kernel is the last grouped launch's buckets: text layers 2..0 (85 MB) and the text embedding (96 MB, the
    tied decoder matrix: its gradient is only complete after the embedding's own backward) -- 181 MB, 18 % of the arena.
    (Before round 6 buckets were handed over in completion ORDER and the image side's last grouped launch was only forced at
    the end of backward: the same step ended with ONE 392 MB collective.)"""
    cfg = _cfg()
    pl = BP.exchange_plan(cfg, wgrad_group_rounds=2, **SHARE8)
    mb = [round(b / 1e6) for b in pl["bytes"]]
    assert len(mb) == 14, mb
    assert max(mb) <= 120 and min(mb) >= 8, mb
    # grouped launches: (queue, problems, 256x256 tiles); no launch above ~2.3 rounds of the chip, image launches of 480 tiles
    text = [t for s, _, t in pl["launches"] if s == 0 and t > 0]
    assert all(t <= 2.3 * 256 for t in text), pl["launches"]
    # the tail: everything handed over by the LAST _bucket_done call
    runs, cur = [], []
    for g, more in pl["calls"]:
        cur.append(g)
        if not more:
            runs.append(cur)
            cur = []
    last_run = runs[-1]
    ranges = {g: (hi - lo) * 4 for g, lo, hi in BP.arena_ranges(cfg)}
    tail_mb = sum(ranges[g] for g in last_run) / 1e6
    assert set(last_run) == {"t2", "t1", "t0", "text_embeddings"}, last_run
    assert 175 <= tail_mb <= 190, tail_mb
    # the image side's last launch goes out with the image embedding, BEFORE text layers 5..0
    order = [g for g, _ in pl["calls"]]
    assert order.index("image_embeddings") < order.index("t5") < order.index("t0")


def test_single_stream_and_no_lm_rows_variants_still_cover_every_bucket():
    cfg = _cfg()
    for kw in (dict(dual_stream=False), dict(image_head_side=False), dict()):
        for lm in (0, 630):
            pl = BP.exchange_plan(cfg, 30, 3900, lm, wgrad_group_rounds=2, **kw)
            assert sorted(g for g, _ in pl["calls"]) == sorted(g for g, _, _ in BP.arena_ranges(cfg))


def test_flush_rule():
    assert BP.big_tiles_of(31162, 3072, 768) == 36 and BP.big_tiles_of(8880, 1024, 5) == 0 and BP.big_tiles_of(600, 768, 768) == 0
    assert BP.flush_due(40, 0, 4) and BP.flush_due(3, 1024, 4) and not BP.flush_due(3, 564, 4)
    assert BP.flush_due(3, 1020, 4) and BP.flush_due(3, 480, 2) and not BP.flush_due(3, 456, 2)
