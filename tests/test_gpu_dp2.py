"""The N > 1 path of the HIP ENGINE (row A27 / (e)): two data-parallel ranks of the real engine, each a child process
on device 0 exchanging gradients over gloo, against the 1-rank HIP result on the concatenated batch under the
reference's semantics: per-replica mean losses, then the mean over replicas (utils/data_parallel.py:120-129,
train.py:164-166).  Covers the parameter broadcast (incl. the refresh of the bf16 weight copies it must trigger),
the per-bucket exchange issued from inside backward beside the two compute streams, `no_sync`, and the opt-in bf16
wire format / reduce-scatter + all-gather exchange."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run_ranks(tmp_path, tag, extra):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    outs = [str(tmp_path / f"{tag}_rank{r}.pt") for r in range(2)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dp2_worker.py"), str(r), "2", str(port), outs[r]] + extra,
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(2)]
    logs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(o)
    for p, o in zip(procs, logs):
        assert p.returncode == 0, o[-3000:]
    return [torch.load(o) for o in outs]


def _single_rank(golden_dir, compute="bf16"):
    """1-rank HIP gradients of each shard (same kernels as the ranks run), and rank 0's initial weights."""
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    from unimm_amd.parallel import shard_range
    cfgd = json.load(open(os.path.join(golden_dir, "small_config.json")))
    model = BertForMultiModalPreTraining(BertConfig.from_dict(cfgd), compute_dtype=compute)
    model.load_state_dict(R.init_state_dict(R.make_config(cfgd), seed=11), strict=True)
    model = model.cuda().eval()
    g = np.load(os.path.join(golden_dir, "small_mixed.npz"))
    batch = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in::")}
    n = batch["input_ids"].shape[0]
    grads, losses = [], []
    for r in range(2):
        lo, hi = shard_range(n, r, 2)
        sh = {k: (v if k == "nsp_weight" else v[lo:hi]) for k, v in batch.items()}
        model.zero_grad(set_to_none=True)
        lm, img, nsp_l, _, _, _ = model(sh["input_ids"], sh["image_feat"], sh["image_loc"], token_type_ids=sh["token_type_ids"],
                                        position_ids=sh["position_ids"], attention_mask=sh["attention_mask"],
                                        image_attention_mask=sh["image_attention_mask"], co_attention_mask=sh["co_attention_mask"],
                                        masked_lm_labels=sh["masked_lm_labels"], image_label=sh["image_label"],
                                        image_target=sh["image_target"], next_sentence_label=sh["next_sentence_label"],
                                        nsp_weight=sh["nsp_weight"], lm_weight=sh["lm_weight"], _want_lm_scores=False)
        (lm + img + nsp_l).sum().backward()
        torch.cuda.synchronize()
        grads.append(model.engine.arena.grad_flat.clone().cpu())
        losses.append([float(lm.detach()), float(img.detach()), float(nsp_l.detach())])
    return dict(grads=grads, losses=losses, flat=model.engine.arena.flat.detach().cpu().clone())


@pytest.fixture(scope="module")
def single_rank(golden_dir):
    return _single_rank(golden_dir)


@pytest.mark.parametrize("wire,algo,tol", [("fp32", "allreduce", 2e-5), ("bf16", "allreduce", 6e-3), ("fp32", "rs_ag", 2e-5)])
def test_two_engine_ranks_equal_single_rank_mean_of_means(tmp_path, single_rank, wire, algo, tol):
    res = _run_ranks(tmp_path, f"{wire}_{algo}", [wire, algo])
    want = 0.5 * (single_rank["grads"][0] + single_rank["grads"][1])
    scale = float(want.abs().max())
    for r, out in enumerate(res):
        assert torch.equal(out["flat"], single_rank["flat"]), "replica weights != rank 0's after the broadcast"
        # the rank's loss is the mean over ITS shard (and equals the 1-rank run of that shard: the broadcast values
        # reached the bf16 weight copies although a forward had already run with the rank's own init)
        assert np.allclose(out["losses"], single_rank["losses"][r], rtol=0, atol=2e-5), (out["losses"], single_rank["losses"][r])
        err = float((out["grad"] - want).abs().max())
        print(f"rank {r} [{wire}/{algo}]: max |g_2rank - mean-of-means| = {err:.3e} of max|g| {scale:.3e}; stats {out['stats']}")
        assert err <= tol * scale, (r, err, scale)
        # no_sync: the second backward accumulated the rank's LOCAL gradient on top of the averaged one
        acc_err = float((out["acc"] - (out["grad"] + single_rank["grads"][r])).abs().max())
        assert acc_err <= 2e-5 * scale, (r, acc_err)
        assert out["stats"]["buckets"] == out["stats"]["n_buckets_expected"]
    assert res[0]["seed"] != res[1]["seed"]              # every replica draws its own dropout masks


def test_two_engine_ranks_with_graph_executor(tmp_path, single_rank):
    """Data parallelism + the step executor: backward is replayed as a chain of graphs cut where gradient buckets are
    handed over to the exchange; gradients equal the eager 2-rank semantics (mean of the ranks' means), no_sync accumulates
    the local gradient, a further replay reproduces the first step."""
    res = _run_ranks(tmp_path, "graphs", ["fp32", "allreduce", "gloo", "graphs"])
    want = 0.5 * (single_rank["grads"][0] + single_rank["grads"][1])
    scale = float(want.abs().max())
    for r, out in enumerate(res):
        gs = out["graphs"]
        print(f"rank {r}: {gs}; comm {out['stats']}")
        # captures: the forward, the backward into a freshly zeroed arena (weight gradients WRITTEN) and the backward of the
        # no_sync micro-step that accumulates on top of it (weight gradients ADDED): write-vs-add is frozen into the launches
        assert gs["replays"] == 3 and gs["eager"] == 0 and gs["captures"] == 3, gs
        assert gs["segments"] and gs["segments"][0] >= 2, gs         # really a chain, not one graph
        assert np.allclose(out["losses"], single_rank["losses"][r], rtol=0, atol=2e-5)
        assert np.allclose(gs["replay_loss"], single_rank["losses"][r], rtol=0, atol=2e-5)
        err = float((out["grad"] - want).abs().max())
        assert err <= 2e-5 * scale, (r, err, scale)
        acc_err = float((out["acc"] - (out["grad"] + single_rank["grads"][r])).abs().max())
        assert acc_err <= 2e-5 * scale, (r, acc_err)
        assert gs["replay_err"] <= 2e-5 * scale, gs
        assert out["stats"]["buckets"] == out["stats"]["n_buckets_expected"]      # third step: every bucket exactly once


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="RCCL needs one device per rank: this box has a single GPU")
@pytest.mark.parametrize("wire,algo,tol", [("fp32", "allreduce", 2e-5), ("bf16", "rs_ag", 6e-3)])
def test_two_engine_ranks_over_rccl(tmp_path, single_rank, wire, algo, tol):
    """The same two ranks with the exchange on RCCL over xGMI (backend "nccl", rank r on device r) -- what
    `bench.py --gpus N` runs.  Skipped on one-GPU boxes; there the gloo variant above covers the engine-side logic."""
    res = _run_ranks(tmp_path, f"rccl_{wire}_{algo}", [wire, algo, "nccl"])
    want = 0.5 * (single_rank["grads"][0] + single_rank["grads"][1])
    scale = float(want.abs().max())
    for r, out in enumerate(res):
        assert torch.equal(out["flat"], single_rank["flat"])
        assert np.allclose(out["losses"], single_rank["losses"][r], rtol=0, atol=2e-5)
        err = float((out["grad"] - want).abs().max())
        assert err <= tol * scale, (r, err, scale)
        assert out["stats"]["buckets"] == out["stats"]["n_buckets_expected"]


def test_two_engine_ranks_of_the_fp32_accuracy_engine(tmp_path, golden_dir):
    """The same two ranks on `compute_dtype="fp32x3"` (unimm_amd/engine_x3.py: the arithmetic class of the reference's dense
    fine-tune, which the driver's configs[3] runs data-parallel): bucket order, hooks and `no_sync` are the base engine's."""
    one = _single_rank(golden_dir, compute="fp32x3")
    res = _run_ranks(tmp_path, "x3", ["fp32", "allreduce", "gloo", "eager", "fp32x3"])
    want = 0.5 * (one["grads"][0] + one["grads"][1])
    scale = float(want.abs().max())
    for r, out in enumerate(res):
        assert torch.equal(out["flat"], one["flat"])
        assert np.allclose(out["losses"], one["losses"][r], rtol=0, atol=2e-5), (out["losses"], one["losses"][r])
        err = float((out["grad"] - want).abs().max())
        assert err <= 2e-5 * scale, (r, err, scale)
        acc_err = float((out["acc"] - (out["grad"] + one["grads"][r])).abs().max())
        assert acc_err <= 2e-5 * scale, (r, acc_err)
        assert out["stats"]["buckets"] == out["stats"]["n_buckets_expected"]


@pytest.mark.parametrize("rounds", [2, 4])
def test_exchange_plan_matches_the_engine(rounds):
    """`unimm_amd.bucket_plan` (host arithmetic; tests/test_bucket_plan_cpu.py checks the N = 8 plan with it) against the
    hand-overs a REAL backward produces at the full config on the 8-GPU share of the headline batch: the same buckets in the
    same order with the same `more` flags, for both grouping depths."""
    from unimm_amd import BertConfig, BertForMultiModalPreTraining, bucket_plan as BP, synth
    cfg = BertConfig.from_json_file(os.path.join(ROOT, "unimm_amd", "config", "bert_base_6layer_6conect.json"))
    torch.manual_seed(0)
    model = BertForMultiModalPreTraining(cfg).cuda().train()
    eng = model.engine
    eng.ensure(torch.device("cuda", 0))
    eng.wgrad_group_rounds = rounds
    calls = []
    eng.grad_bucket_hook = lambda group, more=False: calls.append((group, bool(more)))
    b = synth.make_batch(n_seq=30, T=256, R=37, cfg=cfg, seed=3, device="cuda")
    kw = dict(token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"], attention_mask=b["attention_mask"],
              image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
              masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
              next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"])
    hdr = eng.count_rows({**kw, "input_ids": b["input_ids"], "image_feat": b["image_feat"]})
    B = 30
    lm, img, nsp_l, _, _, _ = model(b["input_ids"], b["image_feat"], b["image_loc"], _want_lm_scores=False, **kw)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    want, launches = BP.hand_overs(cfg, B, sum(hdr[:B]), sum(hdr[B:2 * B]), wgrad_group_rounds=rounds)
    assert calls == want, (calls, want)
    assert sorted(g for g, _ in calls) == sorted(g for g, _, _ in eng.arena.buckets)
    assert BP.arena_ranges(cfg) == list(eng.arena.buckets)
    # one stream: the image side's problems join the text queue; the plan follows
    calls.clear()
    eng.dual_stream = False
    lm, img, nsp_l, _, _, _ = model(b["input_ids"], b["image_feat"], b["image_loc"], _want_lm_scores=False, **kw)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    want1, _ = BP.hand_overs(cfg, B, sum(hdr[:B]), sum(hdr[B:2 * B]), wgrad_group_rounds=rounds, dual_stream=False)
    assert calls == want1, (calls, want1)
