"""Training-loop shell around the hot path: what train.py:350-507 does between the
dataloader and the checkpoint file, without the data plane.

* `train_step`       - one iteration of the loop body (train.py:411-463): forward through the harness, the
                       reference's loss combination, `/ batch_multiply`, backward, optimizer step + zero_grad
                       on accumulation boundaries only, scheduler step every iteration.  bf16 needs no
                       GradScaler; with the data-parallel wrapper the gradient exchange is skipped on
                       non-boundary micro-steps (`no_sync`).
* `save_checkpoint`  - the dict of train.py:503-505 (`model_state_dict` with the `bert_pretrained.` prefix,
                       `scheduler_state_dict`, `optimizer_state_dict`, `iter_id`).
* `load_checkpoint`  - warm start by key intersection (train.py:352-364) or `-continue` (train.py:366-389).
* `expand_image_fields` - train.py:413-432 (one image's features repeated per round and per sample).
* `dense_finetune_step` - one iteration of dense_annotation_finetuning.py:146-301 (row F4): one annotated
                       round of one image against its 100 options (ground truth first, the rest permuted),
                       NeuralNDCG^T + LM + weighted NSP objective, scheduler stepped before the optimizer.
* `visdial_evaluate`  - the validation pass of train.py:180-290 (chunked NSP scoring -> SparseGTMetrics + NDCG).
* `generative_evaluate` - the validation pass of val_lm.py:38-150 (candidates ranked by sequence log-likelihood)."""
from __future__ import annotations

import os
from typing import Optional

import torch

from . import harness, metrics, ranking


def expand_image_fields(batch: dict) -> dict:
    """[n_img, ...] image tensors -> [n_img, rounds, samples, ...] as `forward` flattens them (train.py:413-432)."""
    rounds, samples = batch["tokens"].shape[1], batch["tokens"].shape[2]
    out = dict(batch)
    for k in ("image_feat", "image_loc", "image_target", "image_label", "image_mask"):
        if k not in batch:
            continue                          # evaluation batches carry no targets / labels
        v = batch[k]
        out[k] = v.unsqueeze(1).unsqueeze(1).expand(v.shape[0], rounds, samples, *v.shape[1:]).contiguous()
    return out


def train_step(dialog_encoder, optimizer, scheduler, batch, params, iter_id, sample_size=None):
    """One loop iteration (iter_id is the 1-based counter of train.py:411).  Returns (loss, lm, nsp, img) floats."""
    bm = int(params.get("batch_multiply", 1))
    boundary = iter_id % bm == 0
    dialog_encoder.train()
    sync_ctx = dialog_encoder.no_sync() if (hasattr(dialog_encoder, "no_sync") and not boundary) else _null()
    with sync_ctx:
        loss, lm_loss, nsp_loss, img_loss = harness.forward(dialog_encoder, batch, params, sample_size=sample_size)
        (loss / bm).backward()
    if boundary:
        if hasattr(dialog_encoder, "sync_gradients"):
            dialog_encoder.sync_gradients()           # no-op when every bucket was reduced inside backward
        optimizer.step()
        optimizer.zero_grad()
    scheduler.step()
    return float(loss.detach()), lm_loss, nsp_loss, img_loss


_OPTION_FIELDS = ("tokens", "segments", "positions", "weights", "sep_indices", "mask", "hist_len",
                  "next_sentence_labels", "txt_attention_mask", "co_attention_mask")


def select_options(batch: dict, option_indices: torch.Tensor) -> dict:
    """Reorder / subsample the option axis (dim 2) of every per-option field
    (dense_annotation_finetuning.py:211-220)."""
    out = dict(batch)
    for k in _OPTION_FIELDS:
        out[k] = batch[k].index_select(2, option_indices.to(batch[k].device))
    return out


def dense_finetune_step(dialog_encoder, optimizer, scheduler, batch, params, iter_id, num_options=100, option_indices=None):
    """One dense-annotation iteration.  `batch` holds ONE image (dense_annotation_finetuning.py:159) with
    `tokens` [1, 1, 100, T], `gt_option` and `gt_relevance` [1, 100].  `option_indices` overrides the random
    draw (tests).  Returns (loss, parts) with parts = target / nsp / lm tensors."""
    if batch["image_feat"].shape[0] != 1:
        raise ValueError("dense fine-tuning takes one image per step")
    bm = int(params.get("batch_multiply", 1))
    dialog_encoder.train()
    if option_indices is None:
        gt = int(batch["gt_option"].item())
        n_all = batch["tokens"].shape[2]
        rest = torch.cat([torch.arange(gt), torch.arange(gt + 1, n_all)])[torch.randperm(n_all - 1)[:num_options - 1]]
        option_indices = torch.cat([batch["gt_option"].view(-1).cpu().long(), rest])
    sel = expand_image_fields(select_options(batch, option_indices))
    boundary = iter_id % bm == 0 and iter_id > 0
    sync_ctx = dialog_encoder.no_sync() if (hasattr(dialog_encoder, "no_sync") and not boundary) else _null()
    with sync_ctx:
        _, lm_loss, _, _, nsp_scores = harness.forward(dialog_encoder, sel, params, output_nsp_scores=True)
        dev = nsp_scores.device
        relevance = batch["gt_relevance"].to(dev)[:, option_indices.to(dev)]
        loss, parts = ranking.dense_finetune_loss(nsp_scores, sel["next_sentence_labels"].to(dev), relevance, lm_loss,
                                                  params["nsp_loss_coeff"], num_options=len(option_indices))
        (loss / bm).backward()
    scheduler.step()
    if boundary:
        if hasattr(dialog_encoder, "sync_gradients"):
            dialog_encoder.sync_gradients()
        optimizer.step()
        optimizer.zero_grad()
    return float(loss.detach()) / bm, {k: v.detach() for k, v in parts.items()}


def eval_chunk_size(n_gpus: int) -> int:
    """Sequences per evaluation forward: the largest divisor-friendly size not above 500 * n_gpus / 2 (train.py:186-190)."""
    cap = 500 * (n_gpus / 2)
    sizes = [1, 2, 4, 5, 100, 1000, 200, 8, 10, 40, 50, 500, 20, 25, 250, 125]
    return min(sizes, key=lambda x: abs(x - cap) if x <= cap else float("inf"))


_EVAL_TEXT = (("tokens", 1), ("segments", 1), ("positions", 1), ("weights", 1), ("sep_indices", 1), ("mask", 1),
              ("hist_len", 0), ("txt_attention_mask", 2), ("co_attention_mask", 2))


def _evaluate(dataloader, params, eval_batch_size, dialog_encoder, chunk, score_chunk, collect_ranks=None):
    """Shared loop of the two validation passes: flatten an image's (round, option) sequences, score them chunk by
    chunk with `score_chunk(item) -> [chunk] scores (higher = better answer)`, accumulate the retrieval metrics."""
    sparse, ndcg = metrics.SparseGTMetrics(), metrics.NDCG()
    was_training = dialog_encoder.training
    dialog_encoder.eval()
    with torch.no_grad():
        for batch in dataloader:
            rounds, options = batch["tokens"].shape[1], batch["tokens"].shape[2]
            total = eval_batch_size * rounds * options
            if batch["tokens"].shape[0] != eval_batch_size or total % chunk:
                raise ValueError(f"evaluation batch of {batch['tokens'].shape[0]} images x {rounds} x {options} sequences "
                                 f"does not split into chunks of {chunk}")
            flat = {k: batch[k].reshape((-1,) + tuple(batch[k].shape[-keep:])) if keep else batch[k].reshape(-1)
                    for k, keep in _EVAL_TEXT}
            img = expand_image_fields({k: batch[k] for k in ("tokens", "image_feat", "image_loc", "image_mask")})
            for k in ("image_feat", "image_loc"):
                flat[k] = img[k].reshape((-1,) + tuple(img[k].shape[-2:]))
            flat["image_mask"] = img["image_mask"].reshape(-1, img["image_mask"].shape[-1])
            scores = [score_chunk({k: v[lo:lo + chunk] for k, v in flat.items()}) for lo in range(0, total, chunk)]
            output = torch.cat(scores).view(eval_batch_size, rounds, options)
            dev = output.device
            sparse.observe(output, batch["gt_option_inds"].to(dev))
            rid = batch["round_id"].reshape(-1).to(dev).long()
            ndcg.observe(output[torch.arange(eval_batch_size, device=dev), rid - 1, :], batch["gt_relevance"].to(dev))
            if collect_ranks is not None:
                collect_ranks.append(harness.scores_to_ranks(output).cpu())
    if was_training:
        dialog_encoder.train()
    out = sparse.retrieve(reset=True)
    out.update(ndcg.retrieve(reset=True))
    return out


def visdial_evaluate(dataloader, params, eval_batch_size, dialog_encoder, chunk_size=None):
    """Discriminative evaluation over a validation loader (train.py:180-290): every (round, option) sequence of
    each image is scored in chunks with the NSP head, P(option is the answer) is ranked against the ground-truth
    option (R@k, mean rank, MRR) and, on the densely annotated round, against the relevance scores (NDCG).
    Batches hold `tokens` [eval_batch_size, rounds, options, T], `gt_option_inds`, `gt_relevance`, `round_id`."""
    chunk = int(chunk_size or eval_chunk_size(int(params.get("n_gpus", 1))))

    def score(item):
        nsp_scores = harness.forward(dialog_encoder, item, params, output_nsp_scores=True, evaluation=True)[4]
        return torch.softmax(nsp_scores.float(), dim=1)[:, 0]

    return _evaluate(dataloader, params, eval_batch_size, dialog_encoder, chunk, score)


def generative_evaluate(dataloader, params, eval_batch_size, dialog_encoder, chunk_size=None, average=False, ranks_out=None):
    """Generative evaluation (val_lm.py:38-150; token-mean variant val_avg_lm.py:135 with average=True): every
    candidate answer is scored by the sum of its tokens' log-likelihoods under the masked-LM head, the candidates of a
    round are ranked by that score.  The reference decodes all 256 rows of every sequence into [chunk, 256, 30522]
    logits and cross-entropies them; here only the labelled rows are decoded (`sequence_log_likelihood`).  Chunks of
    `250 * n_gpus / 2` rounded down to the reference's size table (val_lm.py:47-49).  `ranks_out`: optional list that
    receives the [eval_batch_size, rounds, options] rank tensors (what val_lm.py:139-149 writes to its json)."""
    if chunk_size is None:
        cap = 250 * (int(params.get("n_gpus", 1)) / 2)
        sizes = [1, 2, 4, 5, 100, 1000, 200, 8, 10, 40, 50, 500, 20, 25, 250, 125]
        chunk_size = min(sizes, key=lambda x: abs(x - cap) if x <= cap else float("inf"))
    model = _unwrap(dialog_encoder).bert_pretrained

    def score(item):
        s, _ = model.sequence_log_likelihood(
            item["tokens"], item["image_feat"], item["image_loc"], item["mask"], average=average,
            token_type_ids=item["segments"], position_ids=item["positions"], attention_mask=item["txt_attention_mask"],
            image_attention_mask=item["image_mask"], co_attention_mask=item["co_attention_mask"])
        return s

    return _evaluate(dataloader, params, eval_batch_size, dialog_encoder, int(chunk_size), score, collect_ranks=ranks_out)


class _null:
    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def _unwrap(m):
    return m.module if hasattr(m, "module") else m


def save_checkpoint(path, dialog_encoder, optimizer, scheduler, iter_id):
    torch.save({"model_state_dict": _unwrap(dialog_encoder).state_dict(), "scheduler_state_dict": scheduler.state_dict(),
                "optimizer_state_dict": optimizer.state_dict(), "iter_id": iter_id}, path)
    return path


def load_checkpoint(path_or_dict, dialog_encoder, optimizer=None, scheduler=None, resume=False):
    """resume=False: copy every tensor whose key exists in the model (accepts a bare state_dict or the
    training dict); returns the number of keys transferred.  resume=True: also restore optimizer and
    scheduler state and return the saved iter_id."""
    ck = torch.load(path_or_dict, map_location="cpu") if isinstance(path_or_dict, (str, os.PathLike)) else path_or_dict
    model = _unwrap(dialog_encoder)
    model_dict = model.state_dict()
    if not resume:
        sd = ck["model_state_dict"] if "model_state_dict" in ck else ck
        sd = {k: v for k, v in sd.items() if k in model_dict}
        if not sd:
            raise ValueError("load_checkpoint: no key of the checkpoint exists in the model")
        model_dict.update(sd)
        model.load_state_dict(model_dict)
        return len(sd)
    model_dict.update({k: v for k, v in ck["model_state_dict"].items() if k in model_dict})
    model.load_state_dict(model_dict)
    if optimizer is not None:
        optimizer.load_state_dict(ck["optimizer_state_dict"])
    if scheduler is not None:
        scheduler.load_state_dict(ck["scheduler_state_dict"])
    return ck["iter_id"]
