"""Host-side boundary behaviour that needs no GPU: checkpoint key handling of `from_pretrained`
(models/vilbert_dialog.py:1232-1296), rank arithmetic with ties (utils/visdial_metrics.py:21-39), the engine's
weight-version tracking."""
import json
import logging
import os

import numpy as np
import pytest
import torch

from oracle import vilbert_ref as R
from unimm_amd import BertConfig, BertForMultiModalPreTraining
from unimm_amd import harness
from unimm_amd.modeling import load_pretrained_state_dict


def small_cfg(golden_dir):
    return BertConfig.from_dict(json.load(open(os.path.join(golden_dir, "small_config.json"))))


def test_from_pretrained_renames_gamma_beta_and_reports_keys(golden_dir, caplog):
    cfg = small_cfg(golden_dir)
    src = BertForMultiModalPreTraining(cfg)
    with torch.no_grad():
        for n, p in src.named_parameters():
            if "LayerNorm" in n:
                p.copy_(torch.randn_like(p))
    sd = {}
    for k, v in src.state_dict().items():
        if "LayerNorm" in k:                       # the key names of old BERT checkpoints (bert-base-uncased)
            k = k.replace("LayerNorm.weight", "LayerNorm.gamma").replace("LayerNorm.bias", "LayerNorm.beta")
        sd[k] = v.clone()
    dropped = "bert.encoder.c_layer.0.biOutput.q_dense1.weight"
    del sd[dropped]
    sd["bert.pooler.dense.weight"] = torch.zeros(4, 4)           # exists in bert-base, not in this model
    with caplog.at_level(logging.INFO, logger="unimm_amd.modeling"):
        model = BertForMultiModalPreTraining.from_pretrained("unused", cfg, state_dict=sd)
    assert model.pretrained_missing_keys == [dropped]
    assert model.pretrained_unexpected_keys == ["bert.pooler.dense.weight"]
    text = caplog.text
    assert "not initialized from pretrained model" in text and dropped in text
    assert "not used in BertForMultiModalPreTraining" in text and "bert.pooler.dense.weight" in text
    got = model.state_dict()
    for k, v in src.state_dict().items():
        if k != dropped:
            assert torch.equal(got[k], v), k               # the renamed LayerNorm tensors arrived too
    # a VisualDialogEncoder checkpoint (prefix `bert_pretrained.`, train.py:503-505) loads as well
    pref = {"bert_pretrained." + k: v for k, v in src.state_dict().items()}
    m2 = BertForMultiModalPreTraining.from_pretrained("unused", cfg, state_dict=pref)
    assert m2.pretrained_missing_keys == [] and m2.pretrained_unexpected_keys == []
    # shape mismatch: the reference raises (models/vilbert_dialog.py:1288-1294)
    bad = dict(src.state_dict())
    bad["bert.t_pooler.dense.weight"] = torch.zeros(3, 3)
    with pytest.raises(RuntimeError, match="size mismatch for bert.t_pooler.dense.weight"):
        BertForMultiModalPreTraining.from_pretrained("unused", cfg, state_dict=bad)
    # silent when default_gpu is False, as the reference
    missing, unexpected = load_pretrained_state_dict(BertForMultiModalPreTraining(cfg), bad, default_gpu=False)
    assert "bert.t_pooler.dense.weight" not in missing


def test_start_prefix_is_detected_for_a_bare_trunk():
    class Trunk(torch.nn.Module):                  # a model WITHOUT a `.bert` attribute (:1270-1274)
        def __init__(self):
            super().__init__()
            self.embeddings = torch.nn.Linear(3, 2)

    t = Trunk()
    sd = {"bert.embeddings.weight": torch.ones(2, 3), "bert.embeddings.bias": torch.ones(2), "cls.x": torch.zeros(1)}
    missing, unexpected = load_pretrained_state_dict(t, sd)
    assert missing == [] and unexpected == ["cls.x"]
    assert torch.equal(t.embeddings.weight, torch.ones(2, 3))


def test_scores_to_ranks_matches_reference_fixture_with_ties(golden_dir):
    g = np.load(os.path.join(golden_dir, "ranks.npz"))
    s = torch.from_numpy(g["scores"])
    assert torch.equal(harness.scores_to_ranks(s), torch.from_numpy(g["ranks"]))
    assert torch.equal(R.scores_to_ranks(s), torch.from_numpy(g["ranks"]))


def test_weight_version_sees_writes_through_the_parameters(golden_dir):
    """Parameters are `p.data = view` views with their own version counters; the engine's staleness test
    must follow them (and load_state_dict invalidates explicitly)."""
    from unimm_amd.arena import FlatArena
    from unimm_amd import params as PM
    cfg = small_cfg(golden_dir)
    model = BertForMultiModalPreTraining(cfg)
    eng = model.engine
    named = dict(model.named_parameters())
    eng.arena = FlatArena(named, PM.arena_groups(cfg))      # CPU arena: the bookkeeping is device-agnostic
    eng._plist = list(named.values())
    v0 = eng._weight_version()
    with torch.no_grad():
        named["bert.t_pooler.dense.bias"].copy_(torch.ones_like(named["bert.t_pooler.dense.bias"]))
    v1 = eng._weight_version()
    assert v1 != v0
    eng._w_version = v1
    model.load_state_dict(model.state_dict())
    assert eng._w_version is None                           # post-hook of load_state_dict
    assert eng.arena.is_current() and eng.arena.is_current_full()
    mid = eng.arena._probe[len(eng.arena._probe) // 2]         # first parameter of a bucket in the middle of the arena
    named[mid].data = named[mid].data.clone()                  # re-pointed (what a partial .to() / assign=True does)
    assert not eng.arena.is_current() and not eng.arena.is_current_full()


def test_dropout_masks_of_two_keys_are_not_index_permutations_of_one_pattern():
    """Host mirror of the device dropout word (unimm_amd/csrc/common.h: drop_word): keep rate 1 - p, and the mask of a
    second key is NOT the first key's mask read at index ^ k1 ^ k2 (it was, when the key only entered by XOR)."""
    from unimm_amd import dropout as DR
    k1 = DR.make_key(1234, 7, 3)
    k2 = k1 ^ 0x1234                                 # keys that differ in low bits only: the XOR distance stays inside the range
    _, thr, _ = DR.drop_arg(0.1, k1)
    n = 1 << 16
    a = DR.keep_mask(k1, thr, 2 * n)[0::2]           # field 0 of word indices 0 .. n-1
    b = DR.keep_mask(k2, thr, 2 * n)[0::2]
    assert abs(a.mean() - 0.9) < 5e-3 and abs(b.mean() - 0.9) < 5e-3
    d = (k1 ^ k2) & (n - 1)                          # low bits of the XOR distance: a permutation inside this index range
    idx = np.arange(n) ^ d
    agree = float((b == a[idx]).mean())
    assert agree < 0.9, agree                        # independent masks agree on ~0.82 of the positions, a permutation on all
    c = np.corrcoef(a.astype(np.float64), b.astype(np.float64))[0, 1]
    assert abs(c) < 2e-2, c
    # the hash's first stage is affine in the word index (w * M1 + key): a key that is d * M1 further on makes that stage
    # a d-word shift of the first key's -- the second entry of the key (XOR before the multiply) must break the relation
    for dwords in (1, 5, 1000):
        k3 = (k1 + dwords * 0x7FEB352D) & 0xFFFFFFFF
        e = DR.keep_mask(k3, thr, 2 * n)
        f = DR.keep_mask(k1, thr, 2 * (n + dwords))
        agree = float((e == f[2 * dwords:]).mean())
        assert agree < 0.85, (dwords, agree)


def test_dropout_mask_statistics_of_the_affine_stage_hash():
    """Keep rate, neighbour correlations (columns, rows) and row-count dispersion of the host mirror over several keys,
    including degenerate ones."""
    from unimm_amd import dropout as DR
    for key in (0, 1, 0xFFFFFFFF, DR.make_key(7, 0, 0), DR.make_key(7, 1, 0), DR.make_key(8, 0, 5)):
        _, thr, _ = DR.drop_arg(0.1, key)
        m = DR.keep_mask2d(key, thr, 2048, 256).astype(np.float64)
        rate = m.mean()
        assert abs(rate - 0.9) < 3e-3, (key, rate)
        d = m - rate
        v = d.var()
        for name, cc in (("cols", (d[:, :-1] * d[:, 1:]).mean() / v), ("rows", (d[:-1] * d[1:]).mean() / v),
                         ("cols2", (d[:, :-2] * d[:, 2:]).mean() / v), ("diag", (d[:-1, :-1] * d[1:, 1:]).mean() / v)):
            assert abs(cc) < 6e-3, (key, name, cc)
        disp = m.sum(1).var() / (256 * rate * (1 - rate))          # binomial dispersion of the per-row keep counts
        assert 0.85 < disp < 1.15, (key, disp)


def test_bert_config_constructor_forms_and_json_round_trip(tmp_path):
    """BertConfig(int | path) / from_dict / to_dict / to_json_string (models/vilbert_dialog.py:131-274)."""
    c = BertConfig(30522)
    assert c.vocab_size == 30522 and c.hidden_size == 768 and c.fusion_method == "mul" and c.with_coattention is True
    assert c.v_biattention_id == [0, 1] and c.t_biattention_id == [10, 11]
    c2 = BertConfig(30522, v_num_hidden_layers=6, v_biattention_id=[0, 1, 2], t_biattention_id=[9, 10, 11], bi_hidden_size=512)
    assert c2.bi_hidden_size == 512 and c2.v_biattention_id == [0, 1, 2]
    with pytest.raises(AssertionError):               # the reference's constructor asserts (:196-198)
        BertConfig(30522, v_biattention_id=[0, 1, 2])
    with pytest.raises(ValueError, match="First argument must be either a vocabulary size"):
        BertConfig(3.5)
    s = c2.to_json_string()
    assert s.endswith("\n") and json.loads(s) == c2.to_dict()
    assert list(json.loads(s)) == sorted(json.loads(s))          # sort_keys=True, indent=2 as the reference prints it
    p = tmp_path / "cfg.json"
    p.write_text(s)
    for back in (BertConfig.from_json_file(str(p)), BertConfig(str(p)), BertConfig.from_dict(json.loads(s))):
        assert back.to_dict() == c2.to_dict()
    assert repr(c2) == s
    d = c2.to_dict()
    d["v_biattention_id"].append(99)                            # to_dict is a deep copy
    assert c2.v_biattention_id == [0, 1, 2]
    # from_json_file = constructor defaults overwritten by the JSON keys (:257-262)
    shipped = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unimm_amd", "config",
                           "bert_base_6layer_6conect.json")
    full = BertConfig.from_json_file(shipped)
    assert full.vocab_size == 30522 and full.v_hidden_size == 1024 and full.bi_num_attention_heads == 8
    assert full.fusion_method == "mul" and full.predict_feature is False and full.fast_mode is False
