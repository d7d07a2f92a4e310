"""The step executor (unimm_amd/graphs.py): the training step replayed as two hipGraphs per row-count bucket must equal
the eager step -- same losses, same NSP logits, same gradients -- over a sequence of different batches, with dropout on
(the per-step salt is read from device memory) and with valid-row / decoded-row counts that move inside and across
buckets (the real counts are read from device memory; launches are sized for the bucket)."""
import copy
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _build(golden_dir, compute="bf16"):
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    cfgd = json.load(open(os.path.join(golden_dir, "small_config.json")))
    model = BertForMultiModalPreTraining(BertConfig.from_dict(cfgd), compute_dtype=compute)
    model.load_state_dict(R.init_state_dict(R.make_config(cfgd), seed=11), strict=True)
    return model.cuda()


def _step(model, b):
    model.engine.arena.zero_grads() if model.engine.arena is not None else None
    lm, img, nsp_l, _, _, nsp = model(
        b["input_ids"], b["image_feat"], b["image_loc"], token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"],
        attention_mask=b["attention_mask"], image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
        masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
        next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"], _want_lm_scores=False)
    (lm + 0.5 * img + 2.0 * nsp_l).sum().backward()
    torch.cuda.synchronize()
    return (torch.stack([lm, img, nsp_l]).flatten().detach().clone(), nsp.detach().clone(), model.engine.arena.grad_flat.clone())


@pytest.mark.parametrize("compute", ["bf16", "fp32x3"])
@pytest.mark.parametrize("train", [True, False])
def test_graph_replay_equals_eager_steps(golden_dir, train, compute):
    from unimm_amd import synth
    ref, gm = _build(golden_dir, compute), _build(golden_dir, compute)
    for m in (ref, gm):
        m.train(train)
        m.set_dropout_seed(321)
    cfg = ref.config
    seeds = [5, 5, 6, 5, 7, 6, 8, 5]                       # repeated batches (replays) and new ones (other row counts)
    batches = {s: synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=s, device="cuda") for s in set(seeds)}
    # eager first (both models see the same step counter)
    want = [_step(ref, batches[s]) for s in seeds]
    g = gm.engine
    g.ensure(torch.device("cuda", 0))
    graphs = g.enable_graphs(row_bucket=64, lm_bucket=16, capture_after=1, max_entries=8)
    got = [_step(gm, batches[s]) for s in seeds]
    st = graphs.stats
    print(f"\\ngraph executor: {st}")
    assert st["replays"] >= 3 and st["captures"] >= 2
    for i, (w, h) in enumerate(zip(want, got)):
        assert torch.isfinite(h[0]).all() and torch.isfinite(h[2]).all(), i
        assert (w[0] - h[0]).abs().max() <= 2e-6 * max(1.0, float(w[0].abs().max())), (i, w[0], h[0])
        assert (w[1] - h[1]).abs().max() <= 2e-6, i
        d = float((w[2] - h[2]).abs().max() / w[2].abs().max())
        assert d <= 2e-5, (i, d)                          # fp32 summation order of the weight gradients


def test_graph_executor_falls_back_when_not_eligible(golden_dir):
    """Inputs left on the host (`host_staging = False`) keep the eager path (and still work)."""
    from unimm_amd import synth
    m = _build(golden_dir)
    m.train(False)
    m.engine.ensure(torch.device("cuda", 0))
    m.engine.host_staging = False
    graphs = m.engine.enable_graphs(capture_after=0)
    b = synth.make_batch(n_seq=6, T=64, R=37, cfg=m.config, seed=3, device="cpu")
    _step(m, b)
    assert graphs.stats["replays"] == 0 and graphs.stats["captures"] == 0


def test_host_tensors_are_replayed_too(golden_dir):
    """CPU tensors handed to forward() (the reference's calling convention) reach the executor as device tensors -- their masks as
    bit-packed words (inputs.HostStager / PackedMask) -- and are replayed like resident inputs: same losses, NSP logits and
    gradients as the eager step on device tensors, over different batches, dropout on."""
    from unimm_amd import synth
    ref, gm = _build(golden_dir), _build(golden_dir)
    for m in (ref, gm):
        m.train(True)
        m.set_dropout_seed(55)
    cfg = ref.config
    seeds = [5, 6, 5, 7, 5]
    host = {s: synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=s, device="cpu", mask_dtype=torch.int64) for s in set(seeds)}
    dev = {s: {k: (v.cuda() if torch.is_tensor(v) and k != "nsp_weight" else v) for k, v in b.items()} for s, b in host.items()}
    want = [_step(ref, dev[s]) for s in seeds]
    gm.engine.ensure(torch.device("cuda", 0))
    graphs = gm.engine.enable_graphs(row_bucket=64, lm_bucket=16, capture_after=0, max_entries=8)
    got = [_step(gm, host[s]) for s in seeds]
    assert graphs.stats["replays"] >= 3 and graphs.stats["eager"] == 0, graphs.stats
    for i, (w, h) in enumerate(zip(want, got)):
        assert (w[0] - h[0]).abs().max() <= 2e-6 * max(1.0, float(w[0].abs().max())), (i, w[0], h[0])
        assert (w[1] - h[1]).abs().max() <= 2e-6, i
        assert float((w[2] - h[2]).abs().max() / w[2].abs().max()) <= 2e-5, i


def test_graph_entries_are_evicted_and_recaptured(golden_dir):
    """max_entries = 1: two signatures alternate, so every second step evicts the other signature's graphs (and their private
    activation pool) and captures again; results stay equal to the eager steps."""
    from unimm_amd import synth
    ref, gm = _build(golden_dir), _build(golden_dir)
    for m in (ref, gm):
        m.train(True)
        m.set_dropout_seed(77)
    cfg = ref.config
    b1 = synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=5, device="cuda")
    b2 = synth.make_batch(n_seq=6, T=64, R=37, cfg=cfg, seed=6, device="cuda")      # another batch size = another signature
    order = [b1, b2, b1, b2, b1]
    want = [_step(ref, b) for b in order]
    g = gm.engine
    g.ensure(torch.device("cuda", 0))
    graphs = g.enable_graphs(row_bucket=64, lm_bucket=16, capture_after=0, max_entries=1)
    got = [_step(gm, b) for b in order]
    assert len(graphs.entries) == 1 and graphs.stats["captures"] >= 8, graphs.stats      # forward + backward per step after the first
    for i, (w, h) in enumerate(zip(want, got)):
        assert (w[0] - h[0]).abs().max() <= 2e-6 * max(1.0, float(w[0].abs().max())), i
        d = float((w[2] - h[2]).abs().max() / w[2].abs().max())
        assert d <= 2e-5, (i, d)


def test_isolated_launch_survives_a_single_hardware_queue(golden_dir):
    """The HIP runtime of this image reads past the end of an exec's internal stream list (SIGSEGV in
    hip::Graph::UpdateStreams) when all of them share the launch stream's hardware queue -- certain with one queue per
    priority class (GPU_MAX_HW_QUEUES=1), and possible with the default four once destroyed execs have left the queues
    unevenly loaded (unimm_amd/graphs.py never destroys one for that reason).  enable_graphs(launch="isolated") launches from
    a high-priority stream, whose queue comes from another pool: a child process restricted to one queue must run
    captured steps through (and reproduce the eager losses)."""
    import subprocess
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    code = """
import sys
sys.path[:0] = [%r, %r]
import torch
import test_gpu_graphs as T
from unimm_amd import synth
ref, gm = T._build(%r), T._build(%r)
for m in (ref, gm):
    m.train(True); m.set_dropout_seed(3)
b = synth.make_batch(n_seq=12, T=64, R=37, cfg=ref.config, seed=5, device="cuda")
want = [T._step(ref, b) for _ in range(3)]
gm.engine.ensure(torch.device("cuda", 0))
gx = gm.engine.enable_graphs(row_bucket=64, lm_bucket=16, capture_after=0, launch="isolated")
got = [T._step(gm, b) for _ in range(3)]
assert gx.stats["replays"] == 3, gx.stats
for w, h in zip(want, got):
    assert (w[0] - h[0]).abs().max() <= 2e-6 * max(1.0, float(w[0].abs().max()))
    assert float((w[2] - h[2]).abs().max() / w[2].abs().max()) <= 2e-5
print("ran through")
""" % (os.path.dirname(here), here, golden_dir, golden_dir)
    env = dict(os.environ, GPU_MAX_HW_QUEUES="1")
    r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ran through" in r.stdout, (r.returncode, r.stdout[-500:], r.stderr[-2000:])


def test_graph_execs_outlive_their_entries_and_pools_are_recycled(golden_dir):
    """No graph exec is ever destroyed (see unimm_amd/graphs.py); an evicted entry's memory pool is captured into again."""
    import gc
    from unimm_amd import graphs as G, synth
    gm = _build(golden_dir)
    gm.train(True)
    cfg = gm.config
    b1 = synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=5, device="cuda")
    b2 = synth.make_batch(n_seq=6, T=64, R=37, cfg=cfg, seed=6, device="cuda")
    gm.engine.ensure(torch.device("cuda", 0))
    gx = gm.engine.enable_graphs(row_bucket=64, lm_bucket=16, capture_after=0, max_entries=1)
    kept0 = len(G._KEPT)
    pools = []
    for b in (b1, b2, b1, b2):
        _step(gm, b)
        pools.append(next(iter(gx.entries.values())).pool)
    assert len(G._KEPT) - kept0 == gx.stats["captures"] == 8           # forward + backward of four entries, all alive
    assert len(set(pools)) <= 2, pools                                   # the third entry took over the first one's pool
    del gx
    gm.engine.graphs = None
    gc.collect()
    assert len(G._KEPT) - kept0 == 8
    assert pools[-1] in [f[2] for f in G._FREE_POOLS[0]]


def test_graph_replays_follow_optimizer_steps(golden_dir):
    """Weights change between replays (FusedAdamW updates the arena in place, the engine re-casts its bf16 copies before the
    next replay): six optimizer steps over two alternating batches under the graph executor reproduce the eager run's losses
    and final weights."""
    from unimm_amd import synth
    from unimm_amd.optim import FusedAdamW
    ref, gm = _build(golden_dir), _build(golden_dir)
    cfg = ref.config
    batches = [synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=s, device="cuda") for s in (5, 6)]
    res = []
    for m, use_graphs in ((ref, False), (gm, True)):
        m.train(True)
        m.set_dropout_seed(11)
        m.engine.ensure(torch.device("cuda", 0))
        opt = FusedAdamW([dict(params=[p for p in m.parameters()], lr=1e-3, weight_decay=0.01)], m.engine, lr=1e-3)
        if use_graphs:
            gx = m.engine.enable_graphs(row_bucket=64, lm_bucket=16, capture_after=0)
        losses = []
        for it in range(6):
            b = batches[it % 2]
            opt.zero_grad()
            lm, img, nsp_l, _, _, _ = m(
                b["input_ids"], b["image_feat"], b["image_loc"], token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"],
                attention_mask=b["attention_mask"], image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
                masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
                next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"], _want_lm_scores=False)
            (lm + img + nsp_l).sum().backward()
            opt.step()
            losses.append(torch.stack([lm, img, nsp_l]).flatten().detach().clone())
        torch.cuda.synchronize()
        res.append((torch.stack(losses), m.engine.arena.flat.detach().clone()))
    assert gx.stats["replays"] == 6 and gx.stats["eager"] == 0, gx.stats
    (l0, w0), (l1, w1) = res
    assert float((l0[0] - l0[-2]).abs().max()) > 1e-3                      # the weights really moved between replays of one batch
    assert (l0 - l1).abs().max() <= 2e-4 * l0.abs().max(), (l0, l1)
    # weights: Adam normalises every gradient element, so where the two runs' gradients differ in their last bits around zero a
    # step may go the other way -- bounded by the step size per update, and rare
    dw = (w0 - w1).abs()
    assert float(dw.max()) <= 6 * 1.01e-3 and float(dw.mean()) <= 2e-5, (float(dw.max()), float(dw.mean()))


def _call(model, b):
    return model(
        b["input_ids"], b["image_feat"], b["image_loc"], token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"],
        attention_mask=b["attention_mask"], image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
        masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
        next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"], _want_lm_scores=False)


@pytest.mark.parametrize("executor", ["eager", "graphs"])
def test_two_forwards_before_one_backward(golden_dir, executor):
    """`loss = model(b1) + model(b2); loss.backward()` (and an eval forward between a step's forward and its backward): each
    backward must read ITS step's row counts, loss denominators, activations and dropout salt.  Eagerly the row counts live in
    per-forward device words; under the graph executor an entry whose forward has not been back-propagated yet is not replayed
    again (the second forward runs eagerly)."""
    from unimm_amd import synth
    ref, m = _build(golden_dir), _build(golden_dir)
    cfg = ref.config
    b1 = synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=5, device="cuda")
    b2 = synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=9, device="cuda")     # same shapes, other lengths / label counts
    for mm in (ref, m):
        mm.train(True)
        mm.set_dropout_seed(5)
        mm.engine.ensure(torch.device("cuda", 0))
    # reference: the two steps one after the other, gradients accumulated (same step numbers -> same dropout masks)
    ref.engine.arena.zero_grads()
    for b in (b1, b2):
        lm, img, nsp_l, *_ = _call(ref, b)
        (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    want = ref.engine.arena.grad_flat.clone()
    if executor == "graphs":
        gx = m.engine.enable_graphs(row_bucket=4096, lm_bucket=512, capture_after=0)     # one signature for both batches
    m.engine.arena.zero_grads()
    l1 = _call(m, b1)
    l2 = _call(m, b2)                                   # overwrites nothing of step 1
    m.eval()
    with torch.no_grad():
        _call(m, b2)                                    # an evaluation forward in between (no step number, no tape)
    m.train(True)
    ((l1[0] + l1[1] + l1[2]) + (l2[0] + l2[1] + l2[2])).sum().backward()
    torch.cuda.synchronize()
    got = m.engine.arena.grad_flat
    d = float((want - got).abs().max() / want.abs().max())
    assert d <= 2e-5, d
    if executor == "graphs":
        assert gx.stats["busy"] == 1 and gx.stats["replays"] == 1, gx.stats
        m.engine.arena.zero_grads()                      # and the entry is free again afterwards
        la = _call(m, b1)
        (la[0] + la[1] + la[2]).sum().backward()
        assert gx.stats["replays"] == 2, gx.stats


@pytest.mark.parametrize("executor", ["eager", "graphs"])
def test_forward_backward_in_one_call_equals_the_autograd_step(golden_dir, executor):
    """`model.forward_backward(..., loss_weights=(c_lm, c_nsp, c_img))` (both halves enqueued back to back, no autograd round
    trip) against `loss = c_lm * lm + c_nsp * nsp + c_img * img; loss.backward()`: the same losses, NSP logits and gradients,
    eagerly and under the step executor, over steps with different batches and dropout on; gradients accumulate across two
    calls as they do across two backward()s."""
    from unimm_amd import synth
    ref, m = _build(golden_dir), _build(golden_dir)
    cfg = ref.config
    for mm in (ref, m):
        mm.train(True)
        mm.set_dropout_seed(9)
        mm.engine.ensure(torch.device("cuda", 0))
    if executor == "graphs":
        gx = m.engine.enable_graphs(row_bucket=64, lm_bucket=16, capture_after=0)
    c = (1.0, 2.0, 0.5)                                    # (c_lm, c_nsp, c_img)
    kw = lambda b: dict(token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"], attention_mask=b["attention_mask"],
                        image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
                        masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
                        next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"])
    for it, seed in enumerate((5, 6, 5, 5)):
        b = synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=seed, device="cuda")
        if it != 3:                                        # the last step accumulates on top of the third
            ref.engine.arena.zero_grads()
            m.engine.arena.zero_grads()
        lm, img, nsp_l, _, _, nsp = ref(b["input_ids"], b["image_feat"], b["image_loc"], _want_lm_scores=False, **kw(b))
        want = c[0] * lm.sum() + c[1] * nsp_l.sum() + c[2] * img.sum()
        want.backward()
        plan = m.engine.count_rows({**kw(b), "input_ids": b["input_ids"], "image_feat": b["image_feat"]}) if it == 2 else None
        loss, lm2, img2, nsp_l2, nsp2 = m.forward_backward(b["input_ids"], b["image_feat"], b["image_loc"], c, plan_header=plan, **kw(b))
        torch.cuda.synchronize()
        assert abs(float(loss) - float(want)) <= 2e-6 * max(1.0, abs(float(want))), (it, float(loss), float(want))
        for a, bb in ((lm, lm2), (img, img2), (nsp_l, nsp_l2)):
            assert (a.detach() - bb).abs().max() <= 2e-6 * max(1.0, float(a.abs().max())), it
        assert (nsp.detach() - nsp2).abs().max() <= 2e-6, it
        d = float((ref.engine.arena.grad_flat - m.engine.arena.grad_flat).abs().max() / ref.engine.arena.grad_flat.abs().max())
        assert d <= 2e-5, (it, d)
    if executor == "graphs":
        assert gx.stats["replays"] >= 3 and gx.stats["eager"] == 0, gx.stats


def test_stale_plan_header_degrades_instead_of_corrupting_memory(golden_dir):
    """`plan_header` alone sizes the capacities of a replayed step.  A header that UNDERSTATES the batch (a
    prefetcher handing over the header of batch k-1) must not make the replayed plan kernel write past its lists: the
    kernel stays inside the capacities, clamps the row counts the other kernels read and marks the step -- the losses
    come out NaN.  Malformed headers are refused on the host.  Afterwards the executor still runs correct steps."""
    from unimm_amd import lib as L
    from unimm_amd import synth
    ref, m = _build(golden_dir), _build(golden_dir)
    for mm in (ref, m):
        mm.train(False)
        mm.engine.ensure(torch.device("cuda", 0))
    cfg = m.config
    m.engine.enable_graphs(row_bucket=16, lm_bucket=8, capture_after=0)
    c = (1.0, 1.0, 1.0)
    kw = lambda b: dict(token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"], attention_mask=b["attention_mask"],
                        image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
                        masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
                        next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"])
    big = synth.make_batch(n_seq=12, T=64, R=37, cfg=cfg, seed=5, device="cuda")
    B, T = big["input_ids"].shape
    hdr = m.engine.count_rows({**kw(big), "input_ids": big["input_ids"], "image_feat": big["image_feat"]})
    # --- the kernel alone: capacities smaller than the batch's real counts, canaries behind every list
    from unimm_amd.engine import Engine
    eng = m.engine
    eng._dev_masks = []
    tmask = eng._pack_mask(big["attention_mask"], torch.device("cuda", 0), T)
    comask = eng._pack_mask(big["co_attention_mask"], torch.device("cuda", 0), 37)
    eng._dev_masks = []
    lab32 = big["masked_lm_labels"].reshape(B, T).to(torch.int32).contiguous()
    w32 = big["lm_weight"].reshape(B, T).to(torch.int32).contiguous()
    header = L.plan_lengths(tmask, comask, 37, lab32, w32, None, B, T)
    Mv, n_lm = sum(hdr[:B]), sum(hdr[B:2 * B])
    assert Mv > 64 and n_lm > 8
    di, df = torch.zeros(8, dtype=torch.int32, device="cuda"), torch.zeros(8, device="cuda")
    built = L.plan_build(header, lab32, w32, B, T, Mv // 2, n_lm // 2, dims=(di, df))
    torch.cuda.synchronize()
    assert int(di[0]) == Mv // 2 and int(di[1]) == n_lm // 2 and int(di[3]) == 1 and torch.isnan(df[:2]).all()
    assert int(built["rows"].max()) < B * T and int(built["rows"].min()) >= 0
    off, lens = built["off"].tolist(), built["lens"].tolist()
    assert all(0 <= o and ln >= 1 and o + ln <= Mv // 2 for o, ln in zip(off, lens))
    assert int(built["lm_idx"].max()) < Mv // 2 and int(built["lm_pos"].max()) < B * T
    di2, df2 = torch.zeros(8, dtype=torch.int32, device="cuda"), torch.zeros(8, device="cuda")
    L.plan_build(header, lab32, w32, B, T, Mv, n_lm, dims=(di2, df2))
    assert int(di2[0]) == Mv and int(di2[1]) == n_lm and int(di2[3]) == 0 and torch.isfinite(df2[0])
    # --- the executor: a header that claims half the rows.  Nothing may fault, losses are NaN, and memory next to the
    # step's buffers is untouched (the model's parameters are the canary: they live in the same allocator)
    params0 = m.engine.arena.flat.clone()
    stale = list(hdr)
    for i in range(B):
        stale[i] = max(1, hdr[i] // 2)
        stale[B + i] = hdr[B + i] // 2
    out = m.forward_backward(big["input_ids"], big["image_feat"], big["image_loc"], c, plan_header=stale, **kw(big))
    torch.cuda.synchronize()
    assert not torch.isfinite(out[1]).all(), "a batch that does not fit its capacities must be flagged by a NaN loss"
    assert torch.equal(params0, m.engine.arena.flat)
    # malformed headers never reach a capture
    with pytest.raises(ValueError):
        m.forward_backward(big["input_ids"], big["image_feat"], big["image_loc"], c, plan_header=hdr[:-1], **kw(big))
    bad = list(hdr); bad[0] = T + 1
    with pytest.raises(ValueError):
        m.forward_backward(big["input_ids"], big["image_feat"], big["image_loc"], c, plan_header=bad, **kw(big))
    # ... and a correct header (or none) afterwards gives the eager losses again
    m.engine.arena.zero_grads()
    lm, img, nsp_l, _, _, nsp = ref(big["input_ids"], big["image_feat"], big["image_loc"], _want_lm_scores=False, **kw(big))
    for plan in (hdr, None):
        _, lm2, img2, nsp_l2, _ = m.forward_backward(big["input_ids"], big["image_feat"], big["image_loc"], c, plan_header=plan, **kw(big))
        torch.cuda.synchronize()
        for a, bb in ((lm, lm2), (img, img2), (nsp_l, nsp_l2)):
            assert (a.detach() - bb).abs().max() <= 2e-6 * max(1.0, float(a.abs().max()))
