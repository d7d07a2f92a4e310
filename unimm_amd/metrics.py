"""Retrieval metrics of the evaluation loop: accumulators with the
reference's interface (`observe` / `retrieve` / `reset`), computed without per-element Python loops.

`SparseGTMetrics` follows utils/visdial_metrics.py:41-117 (R@1/5/10, mean rank, MRR, overall and per
round); `NDCG` follows :119-193 (NDCG over the k options with non-zero relevance)."""
from __future__ import annotations

import numpy as np
import torch

from .harness import scores_to_ranks


class SparseGTMetrics:
    def __init__(self):
        self.num_rounds = None
        self.reset()

    def reset(self):
        self._gt_ranks = []          # one [batch, rounds] int array per observe()

    def observe(self, predicted_scores: torch.Tensor, target_ranks: torch.Tensor):
        """predicted_scores [batch, rounds, options]; target_ranks [batch, rounds] index of the
        ground-truth option."""
        ranks = scores_to_ranks(predicted_scores.detach())
        b, r, _ = ranks.shape
        self.num_rounds = r
        gt = ranks.gather(2, target_ranks.view(b, r, 1).long().to(ranks.device)).squeeze(2)
        self._gt_ranks.append(gt.cpu().numpy())

    def retrieve(self, reset: bool = True):
        metrics = {}
        if self._gt_ranks:
            per_round = np.concatenate(self._gt_ranks).astype(float)        # [examples, rounds]
            flat = torch.from_numpy(per_round.reshape(-1)).float()          # overall figures in fp32 like the reference
            metrics = {"r@1": (flat <= 1).float().mean().item(), "r@5": (flat <= 5).float().mean().item(),
                       "r@10": (flat <= 10).float().mean().item(), "mean": flat.mean().item(),
                       "mrr": flat.reciprocal().mean().item()}
            cols = {"r_1": (per_round <= 1).mean(0), "r_5": (per_round <= 5).mean(0), "r_10": (per_round <= 10).mean(0),
                    "mean": per_round.mean(0), "mrr": np.reciprocal(per_round).mean(0)}
            for rnd in range(1, self.num_rounds + 1):
                for name, col in cols.items():
                    metrics[f"{name}_round_{rnd}"] = col[rnd - 1]
        if reset:
            self.reset()
        return metrics


class NDCG:
    def __init__(self):
        self.reset()

    def reset(self):
        self._num = 0.0
        self._den = 0.0

    def observe(self, predicted_scores: torch.Tensor, target_relevance: torch.Tensor):
        """predicted_scores, target_relevance: [batch, options] (one annotated round per dialog)."""
        scores = predicted_scores.detach()
        b, o = scores.shape
        rel = target_relevance.to(scores.device).float()
        k = (rel != 0).sum(-1, keepdim=True)
        # order of the options by predicted rank (ties resolved as scores_to_ranks resolves them)
        order = scores_to_ranks(scores.unsqueeze(1)).squeeze(1).sort(-1)[1]
        best = rel.sort(-1, descending=True)[1]
        top = (torch.arange(o, device=scores.device)[None, :] < k).float()
        disc = torch.log2(torch.arange(o, device=scores.device).float() + 2.0)[None, :]
        dcg = (rel.gather(1, order) / disc * top).sum(-1)
        ideal = (rel.gather(1, best) / disc * top).sum(-1)
        self._num += float((dcg / ideal).sum())
        self._den += b

    def retrieve(self, reset: bool = True):
        metrics = {"ndcg": float(self._num / self._den)} if self._den > 0 else {}
        if reset:
            self.reset()
        return metrics
