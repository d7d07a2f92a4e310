"""Parameter inventory of the two-stream model and its flat-arena layout.

`state_dict` keys and shapes are the reference's (535 keys, 250,090,109 unique
parameters at the full config; nn.Linear weights are [out, in]).  The ARENA order is this build's
own: parameters that one fused GEMM reads together (query/key/value of a block) are adjacent, and
the groups follow the order in which backward finishes them, so that every data-parallel gradient
bucket is one contiguous slice."""
from __future__ import annotations

from typing import Dict, List, Tuple

TIED_DECODER = "cls.predictions.decoder.weight"
WORD_EMB = "bert.embeddings.word_embeddings.weight"
# parameters that exist in checkpoints but never take part in forward (models/vilbert_dialog.py:319,
# :734-742): they keep grad None, exactly like the reference, and are skipped by the all-reduce
UNUSED_SUFFIXES = ("sep_embeddings.weight", "q_dense1.weight", "q_dense1.bias", "q_dense2.weight", "q_dense2.bias")


def _lin(out, name, o, i):
    out.append((name + ".weight", (o, i)))
    out.append((name + ".bias", (o,)))


def _ln(out, name, n):
    out.append((name + ".weight", (n,)))
    out.append((name + ".bias", (n,)))


def _qkv(out, prefix, names, o, i):
    """weights of the three projections back to back, then their biases: one [3*o, i] GEMM operand"""
    for n in names:
        out.append((prefix + n + ".weight", (o, i)))
    for n in names:
        out.append((prefix + n + ".bias", (o,)))


def arena_groups(cfg) -> List[Tuple[str, List[Tuple[str, tuple]]]]:
    """[(group name, [(param name, shape), ...]), ...] in arena order."""
    H, Hv, Hb = cfg.hidden_size, cfg.v_hidden_size, cfg.bi_hidden_size
    I, Iv = cfg.intermediate_size, cfg.v_intermediate_size
    groups = []

    g = []
    e = "bert.embeddings."
    g.append((e + "word_embeddings.weight", (cfg.vocab_size, H)))
    g.append((e + "position_embeddings.weight", (cfg.max_position_embeddings, H)))
    g.append((e + "token_type_embeddings.weight", (cfg.type_vocab_size, H)))
    g.append((e + "token_type_embeddings_extension.weight", (10, H)))
    g.append((e + "sep_embeddings.weight", (50, H)))
    _ln(g, e + "LayerNorm", H)
    groups.append(("text_embeddings", g))

    g = []
    v = "bert.v_embeddings."
    _lin(g, v + "image_embeddings", Hv, cfg.v_feature_size)
    _lin(g, v + "image_location_embeddings", Hv, 5)
    _ln(g, v + "LayerNorm", Hv)
    groups.append(("image_embeddings", g))

    def self_block(prefix, h, inter):
        g = []
        _qkv(g, prefix + "attention.self.", ("query", "key", "value"), h, h)
        _lin(g, prefix + "attention.output.dense", h, h)
        _ln(g, prefix + "attention.output.LayerNorm", h)
        _lin(g, prefix + "intermediate.dense", inter, h)
        _lin(g, prefix + "output.dense", h, inter)
        _ln(g, prefix + "output.LayerNorm", h)
        return g

    def conn_block(prefix):
        g = []
        _qkv(g, prefix + "biattention.", ("query1", "key1", "value1"), Hb, Hv)
        _qkv(g, prefix + "biattention.", ("query2", "key2", "value2"), Hb, H)
        o = prefix + "biOutput."
        _lin(g, o + "dense1", Hv, Hb); _ln(g, o + "LayerNorm1", Hv); _lin(g, o + "q_dense1", Hv, Hb)
        _lin(g, o + "dense2", H, Hb); _ln(g, o + "LayerNorm2", H); _lin(g, o + "q_dense2", H, Hb)
        _lin(g, prefix + "v_intermediate.dense", Iv, Hv)
        _lin(g, prefix + "v_output.dense", Hv, Iv); _ln(g, prefix + "v_output.LayerNorm", Hv)
        _lin(g, prefix + "t_intermediate.dense", I, H)
        _lin(g, prefix + "t_output.dense", H, I); _ln(g, prefix + "t_output.LayerNorm", H)
        return g

    for kind, idx in encoder_schedule(cfg):
        if kind == "t":
            groups.append((f"t{idx}", self_block(f"bert.encoder.layer.{idx}.", H, I)))
        elif kind == "v":
            groups.append((f"v{idx}", self_block(f"bert.encoder.v_layer.{idx}.", Hv, Iv)))
        else:
            groups.append((f"c{idx}", conn_block(f"bert.encoder.c_layer.{idx}.")))

    g = []
    _lin(g, "bert.t_pooler.dense", Hb, H)
    _lin(g, "bert.v_pooler.dense", Hb, Hv)
    g.append(("cls.predictions.bias", (cfg.vocab_size,)))
    _lin(g, "cls.predictions.transform.dense", H, H)
    _ln(g, "cls.predictions.transform.LayerNorm", H)
    _lin(g, "cls.bi_seq_relationship", 2, Hb)
    _lin(g, "cls.imagePredictions.transform.dense", Hv, Hv)
    _ln(g, "cls.imagePredictions.transform.LayerNorm", Hv)
    _lin(g, "cls.imagePredictions.decoder", cfg.v_target_size, Hv)
    groups.append(("heads", g))
    return groups


def encoder_schedule(cfg) -> List[Tuple[str, int]]:
    """Order in which BertEncoder.forward runs its blocks (models/vilbert_dialog.py:842-929):
    ('t', i) text layer, ('v', i) image layer, ('c', i) connection layer."""
    sched = []
    v_start = t_start = 0
    for count, (v_end, t_end) in enumerate(zip(cfg.v_biattention_id, cfg.t_biattention_id)):
        sched += [("v", i) for i in range(v_start, v_end)]
        sched += [("t", i) for i in range(t_start, t_end)]
        sched.append(("c", count))
        v_start, t_start = v_end, t_end
    sched += [("v", i) for i in range(v_start, cfg.v_num_hidden_layers)]
    sched += [("t", i) for i in range(t_start, cfg.num_hidden_layers)]
    return sched


def state_dict_names(cfg) -> Dict[str, tuple]:
    """name -> shape for every state_dict key (including the tied decoder alias)."""
    out = {}
    for _, g in arena_groups(cfg):
        for n, s in g:
            out[n] = s
    out[TIED_DECODER] = out[WORD_EMB]
    return out


def is_unused(name: str) -> bool:
    return name.endswith(UNUSED_SUFFIXES)


def no_grad_names(cfg):
    """Parameters that take part in forward but never receive a gradient under the encoder options of
    models/vilbert_dialog.py:842-929 -- the reference leaves their .grad None, so an optimizer never touches them (not even with
    weight decay): the first `fixed_t_layer` text layers and everything of the text embedding except the word table (which the
    tied decoder still trains), the same for `fixed_v_layer` on the image side, and every connection layer when
    `with_coattention` is off."""
    out = set()
    ft, fv = getattr(cfg, "fixed_t_layer", 0), getattr(cfg, "fixed_v_layer", 0)
    for gname, items in arena_groups(cfg):
        frozen = (gname.startswith("t") and gname[1:].isdigit() and int(gname[1:]) < ft) or \
                 (gname.startswith("v") and gname[1:].isdigit() and int(gname[1:]) < fv) or \
                 (gname.startswith("c") and gname[1:].isdigit() and not getattr(cfg, "with_coattention", True)) or \
                 (gname == "text_embeddings" and ft > 0) or (gname == "image_embeddings" and fv > 0)
        if frozen:
            out.update(n for n, _ in items if n != WORD_EMB)
    return out
