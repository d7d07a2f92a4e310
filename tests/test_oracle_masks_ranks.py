"""Oracle pinning, part 1: input-side contract (A28) and scores->ranks (A26) against the goldens
produced by the reference's own functions (oracle/make_goldens.py groups `masks`, `ranks`)."""
import json
import os

import numpy as np
import torch

from oracle import masks as OM
from oracle import vilbert_ref as R

KEYS = ("tokens", "segments", "positions", "sep_indices", "labels", "weights", "txt_attention_mask",
        "co_attention_mask")


def test_masks_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "masks.npz"))
    scripts = json.loads(str(g["scripts"]))
    n = 0
    for sname, utts in scripts.items():
        n_tok = sum(len(u) for u in utts)
        for mp in (0.0, 1.0):
            for neg in (0, 1):
                for fname, fn in (("gen", OM.encode_gen), ("dis", OM.encode_dis)):
                    draws = np.full(n_tok, 0.5)          # 0.5 < 1.0 -> all picked ; 0.5 < 0.0 -> none
                    got = fn(utts, start_segment=1, mask_prob=mp, is_negative=neg, mask_draws=draws)
                    for k in KEYS:
                        want = g[f"{sname}|{mp}|{neg}|{fname}|{k}"]
                        have = got[k]
                        assert have.shape == want.shape, (sname, mp, neg, fname, k)
                        assert np.array_equal(have.astype(np.int64), want.astype(np.int64)), (sname, mp, neg, fname, k)
                        n += 1
    assert n == 4 * 2 * 2 * 2 * 8


def test_ranks_match_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "ranks.npz"))
    got = R.scores_to_ranks(torch.from_numpy(g["scores"]))
    assert np.array_equal(got.numpy(), g["ranks"])
