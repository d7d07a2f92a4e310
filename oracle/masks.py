"""CPU oracle for the input-side contract -- TEST INFRASTRUCTURE.

Restates what `encode_input_gen` / `encode_input_dis` (utils/data_utils.py:139-288, :291-428)
emit for a list of utterances: token / segment / position ids, MLM labels, likelihood /
unlikelihood token weights and the generative / discriminative attention masks.  Pinned by
golden G5 (`tests/golden/masks_*.npz`, reference run with mask_prob=0 and with a scripted
numpy RNG) in `tests/test_oracle_golden.py`.

Only the structure is restated; randomness is passed in explicitly as `mask_draws`
(one uniform per utterance token, consumed in reference order) so that tests are deterministic.
"""
from __future__ import annotations

import numpy as np

CLS, SEP, MASK = 101, 102, 103


def _pad(vals, n, fill=0):
    out = np.full((1, n), fill, dtype=np.int64)
    vals = list(vals)[:n]
    out[0, :len(vals)] = vals
    return out


def _layout(utterances, start_segment, mask_prob, is_negative, mask_draws):
    """Shared prefix of both builders (utils/data_utils.py:158-198 / :307-346): [CLS] u1 [SEP] ...
    Returns per-position lists for the un-duplicated sequence."""
    tok, seg, pos, lab, wgt, seps = [CLS], [start_segment], [0], [0], [0], []
    cur_seg = start_segment
    draws = iter(mask_draws) if mask_draws is not None else None
    n_utt = len(utterances)
    for ui, utt in enumerate(utterances, start=1):
        last = ui == n_utt
        if last and len(utt) <= 1:
            picked = [0] * len(utt)                                    # :174-175
        else:
            picked = [1 if (draws is not None and next(draws) < mask_prob) else 0 for _ in utt]
        first_pos = len(pos)
        tok += list(utt) + [SEP]
        seg += [cur_seg] * (len(utt) + 1)
        lab += picked + [0]
        wgt += ([0] * len(utt) if (last and is_negative) else picked) + [0]   # :183-186
        pos += list(range(first_pos, first_pos + len(utt) + 1))
        seps.append(len(tok) - 1)
        if not last:
            cur_seg ^= 1
    return tok, seg, pos, lab, wgt, seps, cur_seg


def _finish(tok, lab, max_seq_len):
    """labels: -1 where not predicted, else the original token; inputs get [MASK] there
    (utils/data_utils.py:245-250; the 80/10/10 replacement branch :252-257 needs vocab_size and is
    not taken when vocab_size is None)."""
    tokens = _pad(tok, max_seq_len)
    picked = _pad(lab, max_seq_len)
    labels = np.where(picked == 1, tokens, -1)
    tokens = np.where(picked == 1, MASK, tokens)
    return tokens, labels


def encode_dis(utterances, start_segment=0, max_seq_len=256, max_sep_len=25, mask_prob=0.0,
               is_negative=0, mask_draws=None):
    """Discriminative regime (utils/data_utils.py:291-428): bidirectional mask on [0:L)x[0:L),
    co-attention mask 1 on [0:L)."""
    tok, seg, pos, lab, wgt, seps, _ = _layout(utterances, start_segment, mask_prob, is_negative, mask_draws)
    L = len(tok)
    txt = np.zeros((max_seq_len, max_seq_len), dtype=np.int64)
    co = np.zeros(max_seq_len, dtype=np.int64)
    txt[:L, :L] = 1
    co[:L] = 1
    if len(tok) > max_seq_len:
        seps[-1] = max_seq_len - 1
    tokens, labels = _finish(tok, lab, max_seq_len)
    return dict(tokens=tokens, segments=_pad(seg, max_seq_len), positions=_pad(pos, max_seq_len),
                sep_indices=_pad(seps, max_sep_len), labels=labels, weights=_pad(wgt, max_seq_len),
                txt_attention_mask=txt[None], co_attention_mask=co[None])


def encode_gen(utterances, start_segment=0, max_seq_len=256, max_sep_len=25, mask_prob=0.0,
               is_negative=0, weight=1, mask_draws=None):
    """Generative (autoregressive-MLM, two-stream) regime (utils/data_utils.py:139-288).
    With L = length incl. the answer + its [SEP], n = len(answer)+1, c = L - n:
      row 0 (CLS)            -> cols [0, L+n)                      (:202)
      rows [1, c)            -> cols [1, c)                        (:203)
      rows [c, L)  (answer)  -> cols [1, row]   (causal inclusive) (:204)
      rows [L, L+n) (copies) -> cols [1, row-n) (strictly earlier real tokens) + self (:151,:206)
      rows >= L+n            -> all zero                            (:207)
    co-attention mask 1 on [1, c) (:210).  The copy block is all [MASK] with the answer's
    position ids, labels = answer tokens + [SEP], weights +w / -w (:212-227)."""
    tok, seg, pos, lab, wgt, seps, cur_seg = _layout(utterances, start_segment, mask_prob, is_negative, mask_draws)
    ans = list(utterances[-1])
    n = len(ans) + 1
    L = len(tok)
    c = L - n
    T = max_seq_len
    ids = np.arange(T)
    txt = (ids[None, :] == ids[:, None])                     # identity start (:151)
    txt[0, :L + n] = True
    txt[1:c, 1:c] = True
    rows = np.arange(c, L)
    txt[c:L, 1:L] = ids[None, 1:L] <= rows[:, None]
    if L + n <= T:
        txt[L:L + n, 1:L] = ids[None, 1:L] < rows[:, None]
        txt[L + n:, :] = False
    else:                                                    # truncated copy block (:208-209)
        k = T - L
        txt[L:T, 1:L] = ids[None, 1:L] < rows[:k, None]
    co = np.zeros(T, dtype=np.int64)
    co[1:c] = 1
    # duplicate answer as the [MASK]-copy stream
    tok += ans + [SEP]
    seg += [cur_seg] * n
    lab += [1] * n
    wgt += [(-weight if is_negative else weight)] * n
    pos += pos[c:L]
    seps.append(seps[-1] + n)
    if len(tok) > T:
        seps[-1] = T - 1
    tokens, labels = _finish(tok, lab, T)
    return dict(tokens=tokens, segments=_pad(seg, T), positions=_pad(pos, T),
                sep_indices=_pad(seps, max_sep_len), labels=labels, weights=_pad(wgt, T),
                txt_attention_mask=txt[None], co_attention_mask=co[None])
