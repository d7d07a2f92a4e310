"""Row F4: the dense fine-tune objective and the evaluation metrics against values produced by the
reference's own utils/rank_loss.py and utils/visdial_metrics.py (tests/golden/rankloss.npz, written by
oracle/make_goldens.py `rankloss`)."""
import json
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from unimm_amd import metrics, ranking

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rankloss.npz"))
NAMES = json.loads(str(G["names"]))


@pytest.mark.parametrize("name", NAMES)
def test_neural_ndcg_matches_reference(name):
    kw = json.loads(str(G[name + "_kw"]))
    seed = kw.pop("seed")
    pred = torch.from_numpy(G[name + "_pred"].copy()).requires_grad_(True)
    true = torch.from_numpy(G[name + "_true"].copy())
    if seed is not None:
        torch.manual_seed(seed)
    loss = ranking.neuralNDCG_transposed(pred, true, **kw)
    np.testing.assert_allclose(loss.detach().numpy(), G[name + "_loss"], rtol=2e-5, atol=2e-6)
    grad = torch.autograd.grad(loss, pred)[0].numpy() if loss.requires_grad else np.zeros_like(G[name + "_grad"])
    np.testing.assert_allclose(grad, G[name + "_grad"], rtol=2e-4, atol=2e-6)


def test_neural_ndcg_properties():
    torch.manual_seed(0)
    rel = torch.tensor([[1.0, 0.0, 0.5, 0.0, 0.0, 0.2]])
    good = ranking.neuralNDCG_transposed(rel * 4, rel, temperature=0.05)       # scores ordered like the labels
    bad = ranking.neuralNDCG_transposed(-rel * 4, rel, temperature=0.05)
    assert -1.0 - 1e-4 <= float(good) < -0.99 and float(bad) > float(good) + 0.2
    perm = ranking.sinkhorn(ranking.relaxed_sort(torch.rand(2, 9), 1.0, torch.zeros(2, 9, dtype=torch.bool)))
    assert torch.allclose(perm.sum(1), torch.ones(2, 9), atol=1e-5) and torch.allclose(perm.sum(2), torch.ones(2, 9), atol=1e-5)


def test_dense_finetune_loss_composition():
    torch.manual_seed(1)
    nsp_scores = torch.randn(2 * 100, 2)
    labels = torch.randint(0, 2, (2, 100))
    rel = torch.from_numpy(G["three100_true"][:2].copy())
    lm = torch.tensor([2.5, 3.5])
    loss, parts = ranking.dense_finetune_loss(nsp_scores, labels, rel, lm, nsp_loss_coeff=0.5)
    want = ranking.neuralNDCG_transposed(F.softmax(nsp_scores.view(2, 100, 2), -1)[:, :, 0], rel) + 3.0 \
        + 0.5 * F.cross_entropy(nsp_scores, labels.view(-1))
    assert torch.allclose(loss, want, atol=1e-6)
    loss_nan, _ = ranking.dense_finetune_loss(nsp_scores, labels, rel, torch.tensor([float("nan")]), 0.5)
    assert torch.allclose(loss_nan, want - 3.0, atol=1e-6)          # NaN LM loss is dropped (finetune loop :289-292)


def test_sparse_and_ndcg_metrics_match_reference():
    sp = metrics.SparseGTMetrics()
    for i in range(2):
        sp.observe(torch.from_numpy(G["sparse_scores"][i].copy()), torch.from_numpy(G["sparse_gt"][i].copy()))
    got = sp.retrieve()
    keys = json.loads(str(G["sparse_keys"]))
    assert sorted(got) == keys
    np.testing.assert_allclose([float(got[k]) for k in keys], G["sparse_vals"], rtol=1e-6)
    assert sp.retrieve() == {}
    nd = metrics.NDCG()
    for i in range(2):
        nd.observe(torch.from_numpy(G["ndcg_scores"][i].copy()), torch.from_numpy(G["ndcg_rel"][i].copy()))
    np.testing.assert_allclose(nd.retrieve()["ndcg"], float(G["ndcg"]), rtol=1e-6)
    one = metrics.NDCG()                                            # a single dialog (the reference's squeeze() cannot)
    one.observe(torch.from_numpy(G["ndcg_scores"][0][:1].copy()), torch.from_numpy(G["ndcg_rel"][0][:1].copy()))
    assert 0.0 < one.retrieve()["ndcg"] <= 1.0
