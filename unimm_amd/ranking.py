"""Dense-annotation fine-tuning objective: NeuralNDCG-transposed over the
100 answer options of one round, plus the loss composition of the fine-tuning loop.

Mirrors `neuralNDCG_transposed` (utils/rank_loss.py:518-581), which is the only ranking loss the
reference calls (dense_annotation_finetuning.py:288), together with the pieces it uses:
`deterministic_neural_sort` (:79-112), `stochastic_neural_sort` (:125-153), `sinkhorn_scaling`
(:55-78) and `dcg` (:18-54).  The slate is `[batch, 100]`, so this stays in PyTorch on whatever
device the NSP scores live on.  In PyTorch the work is ~25 tiny kernels per Sinkhorn sweep plus a host sync
for the stop test, up to 50 sweeps, and the same again in backward: measured 5.1 ms per step next to the
29 ms encoder step of the dense workload.  So for device tensors the deterministic branch (the one the
reference uses) runs as ONE launch of `unimm_neural_ndcg` (csrc/ranking.hip: value and gradient, one
workgroup per slate, everything in LDS); the PyTorch formulation below remains for the Gumbel branch with
more than one slate, for CPU tensors, and as the restatement the kernel is tested against (it is pinned
to the reference's own values by tests/golden/rankloss.npz).

Closed form used here.  With `m` the number of unpadded items of a slate, `s_j` its scores and
`a_j = sum_k |s_j - s_k|` over unpadded k, row `i` of the relaxed permutation is
`softmax_j(((m + 1 - 2(i+1)) s_j - a_j) / tau)`; the reference's masking conventions are kept:
a row whose own index is a padded item spreads uniformly over the padded columns, padded columns
are otherwise excluded, and rows past `m` use a zero slope."""
from __future__ import annotations

import math

import torch
import torch.nn.functional as F

EPS = 1e-8          # utils/rank_loss.py:6
PAD_LABEL = -1      # utils/rank_loss.py:7


def relaxed_sort(scores: torch.Tensor, tau: float, pad: torch.Tensor) -> torch.Tensor:
    """[N, n] scores, [N, n] bool pad -> [N, n, n] row-stochastic relaxed permutation (row = rank
    position, column = item).  utils/rank_loss.py:79-112."""
    n = scores.shape[1]
    valid = ~pad
    s = scores.masked_fill(pad, 0.0)
    pair_ok = valid[:, :, None] & valid[:, None, :]
    spread = ((s[:, :, None] - s[:, None, :]).abs() * pair_ok).sum(-1)                 # a_j
    m = valid.sum(1, keepdim=True)
    pos = torch.arange(1, n + 1, device=scores.device)[None, :]
    slope = torch.where(pos <= m, (m + 1 - 2 * pos), torch.zeros_like(pos)).to(s.dtype)   # [N, n] per row i
    logits = slope[:, :, None] * s[:, None, :] - spread[:, None, :]
    either = pad[:, :, None] | pad[:, None, :]
    both = pad[:, :, None] & pad[:, None, :]
    logits = logits.masked_fill(either, -math.inf).masked_fill(both, 1.0)
    return torch.softmax(logits / tau, dim=-1)


def gumbel_relaxed_sort(scores, n_samples, tau, pad, beta=1.0, log_scores=True, eps=1e-10):
    """[S, N, n, n] relaxed permutations of Gumbel-perturbed scores (utils/rank_loss.py:113-153).
    Draw order matches the reference: one `torch.rand([S, N, n, 1])` on the scores' device."""
    N, n = scores.shape
    base = scores + scores.min().abs()
    u = torch.rand([n_samples, N, n, 1], device=scores.device)
    noise = beta * -torch.log(-torch.log(u + eps) + eps)
    if log_scores:
        base = torch.log(base + eps)
    pert = (base[None, :, :, None] + noise).view(n_samples * N, n)
    # the reference pairs perturbed slate r with mask row r // n_samples (repeat_interleave), although
    # the slates are laid out sample-major; reproduced as is (identical whenever N == 1, its only use)
    return relaxed_sort(pert, tau, pad.repeat_interleave(n_samples, dim=0)).view(n_samples, N, n, n)


def sinkhorn(mat: torch.Tensor, pad: torch.Tensor = None, tol=1e-6, max_iter=50) -> torch.Tensor:
    """Alternating column / row normalisation until both marginals are within `tol` of one
    (utils/rank_loss.py:55-78).  The stop test is evaluated every sweep like the reference's, so the
    number of sweeps, and therefore the result, is the same."""
    if pad is not None:
        either = pad[:, None, :] | pad[:, :, None]
        both = pad[:, None, :] & pad[:, :, None]
        mat = mat.masked_fill(either, 0.0).masked_fill(both, 1.0)
    for _ in range(max_iter):
        mat = mat / mat.sum(1, keepdim=True).clamp(min=EPS)
        mat = mat / mat.sum(2, keepdim=True).clamp(min=EPS)
        err = torch.maximum((mat.sum(2) - 1.0).abs().max(), (mat.sum(1) - 1.0).abs().max())
        if err < tol:
            break
    if pad is not None:
        mat = mat.masked_fill(either, 0.0)
    return mat


def rank_discounts(n: int, k: int, device) -> torch.Tensor:
    d = 1.0 / torch.log2(torch.arange(n, dtype=torch.float, device=device) + 2.0)
    d[k:] = 0.0
    return d


def ideal_dcg(labels: torch.Tensor, k: int, pad_label=PAD_LABEL) -> torch.Tensor:
    """DCG@k of the labels ranked by themselves with gain 2^y - 1 (utils/rank_loss.py:18-54 called
    as `dcg(y_true, y_true, ats=[k])`); padded items count as label 0 at the tail."""
    y = labels.masked_fill(labels == pad_label, 0.0)
    key = labels.masked_fill(labels == pad_label, -math.inf)
    y = torch.gather(y, 1, key.sort(dim=-1, descending=True)[1])
    n = y.shape[1]
    k = min(k, n)
    d = 1.0 / torch.log2(torch.arange(n, dtype=torch.float, device=y.device) + 2.0)
    return torch.cumsum(((torch.pow(2.0, y) - 1.0) * d)[:, :k], dim=1)[:, k - 1]


class _FusedNDCG(torch.autograd.Function):
    """Per-slate NeuralNDCG^T through `unimm_neural_ndcg`; the gradient comes out of the same launch."""

    @staticmethod
    def forward(ctx, scores, labels, pad, tau, powered, k, max_iter, tol):
        from . import lib as L
        ndcg, alive, dpred, _ = L.neural_ndcg(scores.detach().float(), labels.float(), pad_label=pad, temperature=tau,
                                              powered=powered, k=k, max_iter=max_iter, tol=tol)
        ctx.save_for_backward(dpred)
        ctx.mark_non_differentiable(alive)
        return ndcg, alive

    @staticmethod
    def backward(ctx, g_ndcg, _g_alive):
        (dpred,) = ctx.saved_tensors
        return (g_ndcg[:, None] * dpred, None, None, None, None, None, None, None)


def _fused_ok(y_pred, y_true, max_iter):
    from . import lib as L
    return y_pred.is_cuda and y_true.shape[1] <= L.NDCG_MAX_OPTIONS and 1 <= max_iter <= L.NDCG_MAX_ITER


def neuralNDCG_transposed(y_pred, y_true, padded_value_indicator=PAD_LABEL, temperature=1.,
                          powered_relevancies=True, k=None, stochastic=False, n_samples=32, beta=0.1,
                          log_scores=True, max_iter=50, tol=1e-6):
    """Negative mean NeuralNDCG^T of the slates (utils/rank_loss.py:518-581; same arguments).  Device tensors
    take the single-launch HIP path (deterministic branch; Gumbel branch when there is one slate)."""
    if _fused_ok(y_pred, y_true, max_iter) and (not stochastic or y_true.shape[0] == 1):
        scores, labels = y_pred, y_true
        if stochastic:                      # perturb in PyTorch (autograd carries d pert / d y_pred), sort + Sinkhorn + NDCG fused
            base = y_pred + y_pred.min().abs()
            u = torch.rand([n_samples, 1, y_pred.shape[1], 1], device=y_pred.device)
            noise = beta * -torch.log(-torch.log(u + 1e-10) + 1e-10)
            if log_scores:
                base = torch.log(base + 1e-10)
            scores = (base[None, :, :, None] + noise).view(n_samples, y_pred.shape[1])
            labels = y_true.expand(n_samples, -1)
        ndcg, alive = _FusedNDCG.apply(scores, labels.contiguous(), float(padded_value_indicator), float(temperature),
                                       bool(powered_relevancies), k, int(max_iter), float(tol))
        # all slates dead -> 0 like the reference's early return, without reading a flag back to the host
        return -(ndcg.sum() / alive.sum().clamp(min=1.0))
    return neuralNDCG_transposed_torch(y_pred, y_true, padded_value_indicator, temperature, powered_relevancies, k,
                                       stochastic, n_samples, beta, log_scores, max_iter, tol)


def neuralNDCG_transposed_torch(y_pred, y_true, padded_value_indicator=PAD_LABEL, temperature=1.,
                                powered_relevancies=True, k=None, stochastic=False, n_samples=32, beta=0.1,
                                log_scores=True, max_iter=50, tol=1e-6):
    """The same loss in PyTorch ops (any device)."""
    N, n = y_true.shape
    if k is None:
        k = n
    pad = y_true == padded_value_indicator
    if stochastic:
        perm = gumbel_relaxed_sort(y_pred, n_samples, temperature, pad, beta=beta, log_scores=log_scores)
    else:
        perm = relaxed_sort(y_pred, temperature, pad)[None]
    S = perm.shape[0]
    perm = sinkhorn(perm.reshape(S * N, n, n), pad.repeat_interleave(S, dim=0), tol=tol, max_iter=max_iter)
    perm = perm.view(S, N, n, n)
    # expected discount of each item: column j collects the discounts of the positions it occupies
    exp_disc = torch.einsum("snij,i->snj", perm, rank_discounts(n, k, y_pred.device))
    gains = (torch.pow(2.0, y_true) - 1.0) if powered_relevancies else y_true
    idcg = ideal_dcg(y_true, k, padded_value_indicator)
    ndcg = (gains[None] * exp_disc).sum(2) / (idcg + EPS)
    dead = idcg == 0.0
    if bool(dead.all()):
        return torch.tensor(0.)
    ndcg = ndcg.masked_fill(dead[None], 0.0)
    return -(ndcg.sum() / ((~dead).sum() * S))


def dense_finetune_loss(nsp_scores, nsp_labels, gt_relevance, lm_loss, nsp_loss_coeff, num_options=None):
    """Objective of one dense-annotation fine-tuning step (dense_annotation_finetuning.py:263-293):
    NeuralNDCG^T between P(option is the answer) and the relevance annotations, plus the LM loss
    (skipped when it is NaN, i.e. the batch held no masked token) and the weighted NSP cross entropy.
    `nsp_scores` [batch*options, 2] as the encoder returns them, `gt_relevance` [batch, options]
    already permuted like the options.  Returns (loss, parts)."""
    if num_options is None:
        num_options = gt_relevance.shape[1]
    scores = nsp_scores.view(-1, num_options, 2).float()
    nsp = F.cross_entropy(scores.view(-1, 2), nsp_labels.view(-1))
    p_answer = F.softmax(scores, dim=-1)[:, :, 0]
    target = neuralNDCG_transposed(p_answer, gt_relevance.float())
    lm = lm_loss.mean()
    loss = target + nsp_loss_coeff * nsp
    if not bool(torch.isnan(lm)):
        loss = loss + lm
    return loss, {"target": target, "nsp": nsp, "lm": lm}
