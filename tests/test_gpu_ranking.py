"""Row F4 on the device: `unimm_neural_ndcg` (one launch: relaxed sort, Sinkhorn, NDCG and the gradient) against
the values the reference's utils/rank_loss.py produced (tests/golden/rankloss.npz) and against the PyTorch
restatement on the same inputs."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "rankloss.npz"))
NAMES = [n for n in json.loads(str(G["names"])) if n != "stoch"]


@pytest.mark.parametrize("name", NAMES)
def test_fused_neural_ndcg_matches_reference_values(name):
    from unimm_amd import ranking
    kw = json.loads(str(G[name + "_kw"]))
    kw.pop("seed")
    pred = torch.from_numpy(G[name + "_pred"].copy()).cuda().requires_grad_(True)
    true = torch.from_numpy(G[name + "_true"].copy()).cuda()
    loss = ranking.neuralNDCG_transposed(pred, true, **kw)
    assert loss.is_cuda
    np.testing.assert_allclose(loss.detach().cpu().numpy(), G[name + "_loss"], rtol=1e-4, atol=2e-6)   # fp32 tolerance 1e-3 (north_star)
    loss.backward()
    want = G[name + "_grad"]
    np.testing.assert_allclose(pred.grad.cpu().numpy(), want, rtol=2e-3, atol=1e-3 * max(np.abs(want).max(), 1e-6))


def test_fused_neural_ndcg_sweeps_temperature_and_sizes():
    """Random slates of several sizes / temperatures / truncations with padding anywhere: value, gradient and the
    number of Sinkhorn sweeps against the PyTorch restatement (golden-pinned in the CPU suite)."""
    from unimm_amd import lib as L, ranking
    rng = np.random.Generator(np.random.PCG64(3))
    for n, tau, k, powered in [(100, 1.0, None, True), (128, 1.0, None, True), (100, 0.1, 10, True), (37, 2.0, 5, False),
                               (1, 1.0, None, True), (2, 0.5, 1, True), (65, 0.05, None, True)]:
        pred = rng.random((5, n), dtype=np.float32)
        true = rng.choice(np.array([0, 0, 0, 0.2, 0.5, 1.0], np.float32), size=(5, n))
        if n > 4:
            true[1, n - 3:] = -1
            true[2, 1] = -1
            true[3, :] = 0
            true[4, :] = -1
        p_cpu = torch.from_numpy(pred.copy()).requires_grad_(True)
        want = ranking.neuralNDCG_transposed_torch(p_cpu, torch.from_numpy(true), temperature=tau, k=k, powered_relevancies=powered)
        gw = torch.autograd.grad(want, p_cpu)[0].numpy() if want.requires_grad else np.zeros_like(pred)
        p_dev = torch.from_numpy(pred.copy()).cuda().requires_grad_(True)
        got = ranking.neuralNDCG_transposed(p_dev, torch.from_numpy(true).cuda(), temperature=tau, k=k, powered_relevancies=powered)
        got.backward()
        assert abs(float(got) - float(want)) <= 1e-4 * max(1.0, abs(float(want))), (n, tau, float(got), float(want))
        err = np.abs(p_dev.grad.cpu().numpy() - gw).max()
        assert err <= 2e-3 * max(np.abs(gw).max(), 1e-6), (n, tau, k, err, np.abs(gw).max())
        assert torch.isfinite(p_dev.grad).all()
    # per-slate outputs of the C-ABI call itself
    pred = torch.rand(3, 100, device="cuda")
    true = torch.zeros(3, 100, device="cuda")
    true[0, :7] = 1.0
    true[2, 5] = 0.5
    ndcg, alive, dpred, iters = L.neural_ndcg(pred, true)
    assert alive.tolist() == [1.0, 0.0, 1.0] and float(ndcg[1]) == 0.0 and float(dpred[1].abs().max()) == 0.0
    assert all(1 <= int(t) <= 50 for t in iters) and (0 < ndcg[0] <= 1.0 + 1e-5)
    with pytest.raises(L.UnimmHipError):
        L.neural_ndcg(torch.rand(1, 129, device="cuda"), torch.zeros(1, 129, device="cuda"))
    with pytest.raises(L.UnimmHipError):
        L.neural_ndcg(pred, true, max_iter=65)


def test_fused_neural_ndcg_gumbel_branch_and_finite_difference():
    from unimm_amd import ranking
    pred = torch.rand(1, 20, device="cuda") + 0.05
    true = torch.tensor([[0, 1.0, 0, 0.5, 0, 0, 0.2, 0, 0, 0, 1.0, 0, 0, 0, 0.4, 0, 0, 0, 0, 0]], device="cuda")
    a = pred.clone().requires_grad_(True)
    b = pred.clone().requires_grad_(True)
    torch.manual_seed(11)
    la = ranking.neuralNDCG_transposed(a, true, stochastic=True, n_samples=4)
    torch.manual_seed(11)
    lb = ranking.neuralNDCG_transposed_torch(b, true, stochastic=True, n_samples=4)
    la.backward(); lb.backward()
    assert abs(float(la) - float(lb)) <= 1e-4
    assert (a.grad - b.grad).abs().max() <= 2e-3 * b.grad.abs().max()
    # central differences in fp64-ish steps on the deterministic loss (tau large enough to be smooth)
    p = pred.clone().requires_grad_(True)
    loss = ranking.neuralNDCG_transposed(p, true, temperature=1.0)
    loss.backward()
    h = 1e-2
    for j in (1, 3, 7):
        e = torch.zeros_like(pred); e[0, j] = h
        fd = (float(ranking.neuralNDCG_transposed(pred + e, true)) - float(ranking.neuralNDCG_transposed(pred - e, true))) / (2 * h)
        assert abs(fd - float(p.grad[0, j])) <= 0.1 * abs(fd) + 2e-4, (j, fd, float(p.grad[0, j]))
