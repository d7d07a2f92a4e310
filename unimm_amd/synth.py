"""Synthetic UniMM-UL batches with the structure the reference's dataloader emits (SURVEY.md 8a row A28,
8d): generative / discriminative attention masks, MLM labels, likelihood / unlikelihood token weights,
36 region features + the global <IMG> row.  Used by bench.py, smoke() and the tests (there is no
VisDial data or tokenizer in this environment).

Mask rules (utils/data_utils.py:139-288, :291-428), for a sequence [CLS] u1 [SEP] ... ans [SEP] of
length L, n = len(ans)+1, c = L-n:
  discriminative: text mask 1 on [0,L)x[0,L); co-attention mask 1 on [0,L)
  generative    : a second, all-[MASK] copy of the answer is appended at [L, L+n) with the answer's
                  position ids; CLS row -> [0,L+n); context rows [1,c) -> [1,c); answer rows -> [1,row]
                  (causal, inclusive); copy row L+i -> [1,c+i) and itself; rows >= L+n empty;
                  co-attention mask 1 on [1,c)
"""
from __future__ import annotations

import numpy as np
import torch

CLS, SEP, MASK = 101, 102, 103


def build_sequence(utt_lens, mode, negative, T=256, vocab=30522, mask_prob=0.15, rng=None, tokens=None,
                   start_segment=0, weight=1):
    """One sequence from utterance lengths (last one = the answer).  Returns a dict of [T]-shaped
    int64 arrays + masks ([T,T] bool, [T] int64).  `tokens`: optional list of per-utterance token lists."""
    rng = rng or np.random.default_rng(0)
    n_utt = len(utt_lens)
    L = 1 + sum(l + 1 for l in utt_lens)
    n = utt_lens[-1] + 1
    c = L - n
    gen = mode == "gen"
    total = L + n if gen else L
    tok = np.zeros(total, dtype=np.int64)
    seg = np.zeros(total, dtype=np.int64)
    pos = np.zeros(total, dtype=np.int64)
    picked = np.zeros(total, dtype=bool)       # positions the MLM predicts
    wgt = np.zeros(total, dtype=np.int64)
    tok[0], seg[0] = CLS, start_segment
    p, s = 1, start_segment
    for ui, l in enumerate(utt_lens):
        last = ui == n_utt - 1
        words = np.asarray(tokens[ui]) if tokens is not None else rng.integers(min(1000, vocab // 2), vocab, size=l)
        tok[p:p + l] = words
        tok[p + l] = SEP
        seg[p:p + l + 1] = s
        pos[p:p + l + 1] = np.arange(p, p + l + 1)
        if not (last and l <= 1):
            pick = rng.random(l) < mask_prob
            picked[p:p + l] = pick
            if not (last and negative):            # no likelihood on a negative's last utterance (:183-186)
                wgt[p:p + l] = pick
        p += l + 1
        if not last:
            s ^= 1
    if gen:
        tok[L:L + n] = tok[c:L]
        seg[L:L + n] = s
        pos[L:L + n] = pos[c:L]
        picked[L:L + n] = True
        wgt[L:L + n] = -weight if negative else weight
    labels = np.where(picked, tok, -1)
    tok = np.where(picked, MASK, tok)

    ids = np.arange(T)
    if gen:
        txt = ids[None, :] == ids[:, None]
        txt[0, :L + n] = True
        txt[1:c, 1:c] = True
        rows = np.arange(c, L)
        txt[c:L, 1:L] = ids[None, 1:L] <= rows[:, None]
        if L + n <= T:
            txt[L:L + n, 1:L] = ids[None, 1:L] < rows[:, None]
            txt[L + n:, :] = False
        else:
            txt[L:T, 1:L] = ids[None, 1:L] < rows[:T - L, None]
        co = np.zeros(T, dtype=np.int64)
        co[1:c] = 1
    else:
        txt = np.zeros((T, T), dtype=bool)
        txt[:L, :L] = True
        co = np.zeros(T, dtype=np.int64)
        co[:L] = 1

    def pad(a, fill=0):
        out = np.full(T, fill, dtype=np.int64)
        m = min(T, len(a))
        out[:m] = a[:m]
        return out

    return dict(tokens=pad(tok), segments=pad(seg), positions=pad(pos), labels=pad(labels, -1), weights=pad(wgt),
                txt_attention_mask=txt, co_attention_mask=co, L=L, n=n)


def random_utterances(rng, T=256, c_range=(40, 200), a_range=(2, 12)):
    """Utterance lengths with context length c ~ U{c_range}, answer length a ~ U{a_range}, L + copy <= T."""
    a = int(rng.integers(a_range[0], a_range[1] + 1))
    c_max = min(c_range[1], T - 2 * (a + 1))
    c = int(rng.integers(min(c_range[0], c_max), c_max + 1))
    n_ctx = int(rng.integers(2, 8))
    n_ctx = max(1, min(n_ctx, (c - 1) // 2))
    body = c - 1 - n_ctx                      # context tokens excluding CLS and the SEPs
    cuts = np.sort(rng.integers(0, body + 1, size=n_ctx - 1)) if n_ctx > 1 else np.array([], dtype=np.int64)
    lens = np.diff(np.concatenate([[0], cuts, [body]])).astype(int).tolist()
    return lens + [a]


def make_batch(n_seq=240, T=256, R=37, cfg=None, seed=1234, dis_rate=0.5, num_negative=5, mask_prob=0.15,
               sequences_per_image=6, device="cpu", mask_dtype=torch.bool, modes=None, compact=False):
    """A full training batch (keyword names = VisualDialogEncoder.forward's).  One image per
    `sequences_per_image` sequences (features expanded per sequence, as train.py:422-432 does)."""
    vocab = cfg.vocab_size if cfg is not None else 30522
    F = cfg.v_feature_size if cfg is not None else 2048
    C = cfg.v_target_size if cfg is not None else 1601
    rng = np.random.default_rng(seed)
    rows = []
    for i in range(n_seq):
        mode = modes[i] if modes is not None else ("dis" if rng.random() < dis_rate else "gen")
        neg = (i % (num_negative + 1)) != 0
        rows.append(build_sequence(random_utterances(rng, T), mode, neg, T=T, vocab=vocab, mask_prob=mask_prob, rng=rng,
                                   start_segment=int(rng.integers(0, 2))))
        rows[-1]["neg"] = neg
        rows[-1]["mode"] = mode
    stack = lambda k: torch.from_numpy(np.stack([r[k] for r in rows]))
    n_img = (n_seq + sequences_per_image - 1) // sequences_per_image
    feat = np.maximum(rng.standard_normal((n_img, R, F), dtype=np.float32), 0)
    feat[:, 0] = feat[:, 1:].mean(1)
    loc = rng.random((n_img, R, 5), dtype=np.float32)
    loc[:, 0] = [0, 0, 1, 1, 1]
    tgt = rng.standard_normal((n_img, R, C), dtype=np.float32)
    tgt = np.exp(tgt - tgt.max(-1, keepdims=True))
    tgt /= tgt.sum(-1, keepdims=True)
    img_of = np.arange(n_seq) // sequences_per_image
    label = np.where(rng.random((n_seq, R)) < 0.15, 1, -1)
    label[np.arange(n_seq), rng.integers(1, R, size=n_seq)] = 1     # at least one region predicted (:473-474)
    label[:, 0] = 0
    co = stack("co_attention_mask")[:, None, :].expand(n_seq, R, T)
    batch = dict(
        input_ids=stack("tokens"), token_type_ids=stack("segments"), token_position_ids=stack("positions"),
        masked_lm_labels=stack("labels"), lm_weight=stack("weights"),
        attention_mask=stack("txt_attention_mask").to(mask_dtype), co_attention_mask=co.to(mask_dtype).contiguous(),
        image_feat=torch.from_numpy(feat[img_of]), image_loc=torch.from_numpy(loc[img_of]),
        image_target=torch.from_numpy(tgt[img_of]), image_attention_mask=torch.ones(n_seq, R),
        image_label=torch.from_numpy(label.astype(np.int64)),
        next_sentence_label=torch.tensor([int(r["neg"]) for r in rows], dtype=torch.int64),
        nsp_weight=torch.tensor([[float(num_negative), 1.0]]),
        sep_indices=torch.zeros(n_seq, 25, dtype=torch.int64), sep_len=torch.ones(n_seq, dtype=torch.int64),
    )
    if device != "cpu":
        batch = {k: (v.to(device) if k != "nsp_weight" else v) for k, v in batch.items()}
    if compact:
        # the same batch without the dense masks and the per-sequence copies of the image tensors (row F3):
        # mask descriptors + one entry per image + an image index per sequence
        from .inputs import DialogMaskSpec
        batch["mask_spec"] = DialogMaskSpec([1 if r["mode"] == "gen" else 0 for r in rows], [r["L"] for r in rows],
                                            [r["n"] for r in rows])
        batch["image_index"] = torch.from_numpy(img_of.astype(np.int64))
        batch["image_feat_unique"] = torch.from_numpy(feat)
        batch["image_loc_unique"] = torch.from_numpy(loc)
        batch["image_target_unique"] = torch.from_numpy(tgt)
        if device != "cpu":
            for k in ("image_index", "image_feat_unique", "image_loc_unique", "image_target_unique"):
                batch[k] = batch[k].to(device)
    return batch


def make_scoring_batch(rounds=10, options=100, T=256, R=37, cfg=None, seed=1234, device="cpu", mask_dtype=torch.bool,
                       c_range=(40, 200), a_range=(2, 12)):
    """One image as val_lm.py:52-121 sees it: `rounds` dialog rounds x `options` candidate answers, generative mode, no random
    masking (labels = the answer copy).  The `options` sequences of a round share the image and the whole context
    [CLS] caption [SEP] q1 [SEP] a1 ... q_r [SEP] (tokens, segments, positions) and differ in the candidate answer only --
    dataloader/dataloader_visdial.py builds them that way.  Returns the same keys as make_batch (sequence b = round b // options,
    option b % options) plus `context_group` [rounds * options] = the round index, and `mask_spec`."""
    vocab = cfg.vocab_size if cfg is not None else 30522
    F = cfg.v_feature_size if cfg is not None else 2048
    rng = np.random.default_rng(seed)
    rows, group = [], []
    for r in range(rounds):
        ctx = random_utterances(rng, T, c_range=c_range, a_range=(a_range[1], a_range[1]))[:-1]     # context sized for the longest answer
        ctx_tok = [rng.integers(min(1000, vocab // 2), vocab, size=l) for l in ctx]
        seg0 = int(rng.integers(0, 2))
        for _ in range(options):
            a = int(rng.integers(a_range[0], a_range[1] + 1))
            ans = rng.integers(min(1000, vocab // 2), vocab, size=a)
            rows.append(build_sequence(ctx + [a], "gen", False, T=T, vocab=vocab, mask_prob=0.0, rng=rng, tokens=ctx_tok + [ans],
                                       start_segment=seg0))
            group.append(r)
    n_seq = len(rows)
    stack = lambda k: torch.from_numpy(np.stack([r[k] for r in rows]))
    feat = np.maximum(rng.standard_normal((1, R, F), dtype=np.float32), 0)
    feat[:, 0] = feat[:, 1:].mean(1)
    loc = rng.random((1, R, 5), dtype=np.float32)
    loc[:, 0] = [0, 0, 1, 1, 1]
    co = stack("co_attention_mask")[:, None, :].expand(n_seq, R, T)
    from .inputs import DialogMaskSpec
    batch = dict(
        input_ids=stack("tokens"), token_type_ids=stack("segments"), token_position_ids=stack("positions"),
        masked_lm_labels=stack("labels"), attention_mask=stack("txt_attention_mask").to(mask_dtype),
        co_attention_mask=co.to(mask_dtype).contiguous(),
        image_feat=torch.from_numpy(feat).expand(n_seq, R, F).contiguous(), image_loc=torch.from_numpy(loc).expand(n_seq, R, 5).contiguous(),
        image_attention_mask=torch.ones(n_seq, R), context_group=torch.tensor(group, dtype=torch.int64))
    if device != "cpu":
        batch = {k: v.to(device) for k, v in batch.items()}
    batch["mask_spec"] = DialogMaskSpec([1] * n_seq, [r["L"] for r in rows], [r["n"] for r in rows])
    return batch


def make_loader_batch(n_img=2, rounds=2, samples=3, T=256, R=37, cfg=None, seed=1234, **kw):
    """The same content in the DATALOADER's layout (what train.py's `forward` receives, train.py:30-112):
    text fields [n_img, rounds, samples, ...], image fields per image [n_img, R, ...] (train.py:413-432
    expands them; unimm_amd.trainer.expand_image_fields does the same)."""
    spi = rounds * samples
    b = make_batch(n_seq=n_img * spi, T=T, R=R, cfg=cfg, seed=seed, sequences_per_image=spi, **kw)
    shp = lambda t: t.reshape((n_img, rounds, samples) + tuple(t.shape[1:]))
    per_img = lambda t: t[::spi].contiguous()
    return dict(tokens=shp(b["input_ids"]), segments=shp(b["token_type_ids"]), positions=shp(b["token_position_ids"]),
                mask=shp(b["masked_lm_labels"]), weights=shp(b["lm_weight"]), txt_attention_mask=shp(b["attention_mask"]),
                co_attention_mask=shp(b["co_attention_mask"]), next_sentence_labels=shp(b["next_sentence_label"]),
                sep_indices=shp(b["sep_indices"]), hist_len=shp(b["sep_len"]) - 1,
                image_feat=per_img(b["image_feat"]), image_loc=per_img(b["image_loc"]), image_target=per_img(b["image_target"]),
                image_label=per_img(b["image_label"]), image_mask=per_img(b["image_attention_mask"])), b["nsp_weight"]
