"""Oracle pinning, part 2: the CPU restatement (oracle/vilbert_ref.py) against goldens produced by the
reference's own modules (oracle/make_goldens.py).  fp32 vs fp32: tolerance 2e-5 abs on O(1) values."""
import json
import os

import numpy as np
import pytest
import torch

from oracle import vilbert_ref as R
from oracle.cases import block_inputs, embedding_inputs, loss_inputs

TOL = 2e-5


def T_(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def close(a, b, tol=TOL, what=""):
    a = a.detach().numpy() if torch.is_tensor(a) else np.asarray(a)
    err = np.abs(a.astype(np.float64) - np.asarray(b, dtype=np.float64)).max()
    assert err <= tol * max(1.0, float(np.abs(b).max())), f"{what}: max err {err}"


def oracle_kwargs(g, train=True, use_lm_weight=True):
    i = lambda k: T_(g["in::" + k])
    kw = dict(token_type_ids=i("token_type_ids"), position_ids=i("position_ids"), attention_mask=i("attention_mask"),
              image_attention_mask=i("image_attention_mask"), co_attention_mask=i("co_attention_mask"))
    if train:
        kw.update(masked_lm_labels=i("masked_lm_labels"), image_label=i("image_label"), image_target=i("image_target"),
                  next_sentence_label=i("next_sentence_label"), nsp_weight=i("nsp_weight"),
                  lm_weight=i("lm_weight") if use_lm_weight else None)
    return (i("input_ids"), i("image_feat"), i("image_loc")), kw


@pytest.fixture(scope="module")
def small(golden_dir):
    cfg = R.make_config(json.load(open(os.path.join(golden_dir, "small_config.json"))))
    return cfg, R.init_state_dict(cfg, seed=11)


@pytest.mark.parametrize("case", ["dis", "genpos", "genneg", "mixed"])
def test_small_end_to_end(golden_dir, small, case):
    cfg, sd = small
    g = np.load(os.path.join(golden_dir, f"small_{case}.npz"))
    args, kw = oracle_kwargs(g)
    with torch.no_grad():
        out = R.forward(sd, cfg, *args, **kw)
    for k in ("lm_loss", "img_loss", "nsp_loss", "nsp", "seq_out_t"):
        close(out[k], g[k], what=f"{case}/{k}")
    close(out["pred_t"].reshape(-1, cfg.vocab_size)[g["pred_rows"]], g["pred_t_rows"], what="pred_t rows")
    # inference branch + CE fallback (lm_weight=None)
    args, kw = oracle_kwargs(g, train=False)
    with torch.no_grad():
        inf = R.forward(sd, cfg, *args, **kw)
        args, kw = oracle_kwargs(g, use_lm_weight=False)
        ce = R.forward(sd, cfg, *args, **kw)
    assert "lm_loss" not in inf
    close(inf["pred_v"], g["inf_pred_v"], what="pred_v")
    close(inf["nsp"], g["inf_nsp"], what="inf nsp")
    close(ce["lm_loss"], g["lm_loss_ce"], what="lm CE fallback")


def test_small_gradients(golden_dir, small):
    cfg, sd = small
    g = np.load(os.path.join(golden_dir, "small_mixed.npz"))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    args, kw = oracle_kwargs(g)
    out = R.forward(leaves, cfg, *args, **kw)
    (out["lm_loss"] + out["img_loss"] + out["nsp_loss"]).sum().backward()
    names, norms = list(g["grad_names"]), g["grad_norms"]
    checked = 0
    for n, want in zip(names, norms):
        n = str(n)
        got = leaves[n].grad
        if want < 0:          # parameter never used by forward (sep_embeddings, q_dense1/2)
            assert got is None or float(got.abs().max()) == 0.0, n
            continue
        assert abs(float(got.norm()) - want) <= 1e-4 * max(want, 1e-3), (n, float(got.norm()), want)
        checked += 1
    assert checked > 100
    for k in g.files:
        if k.startswith("grad::"):
            close(leaves[k[6:]].grad, g[k], tol=1e-4, what=k)
    close(leaves["bert.embeddings.word_embeddings.weight"].grad[:64], g["grad_rows::word_embeddings"], tol=1e-4)


@pytest.fixture(scope="module")
def full_sd3():
    cfg = R.make_config(os.path.join(os.path.dirname(os.path.dirname(__file__)), "unimm_amd", "config",
                                     "bert_base_6layer_6conect.json"))
    return cfg, R.init_state_dict(cfg, seed=3)


def test_full_size_blocks(golden_dir, full_sd3):
    cfg, sd = full_sd3
    g = np.load(os.path.join(golden_dir, "block_layers.npz"))
    bi = block_inputs(int(g["seed"]))
    xt, xv, rows = T_(bi["xt"]), T_(bi["xv"]), bi["rows"]
    t_add = R.additive(T_(bi["tmask"]))[:, None]
    v_add = R.additive(T_(bi["vmask"]))[:, None, None, :]
    co_add = R.additive(T_(bi["co"])).unsqueeze(1)
    drop = R._Drop(None)
    with torch.no_grad():
        close(R.text_layer(sd, cfg, 3, xt, t_add, drop)[:, rows], g["text_layer3"], what="text layer")
        close(R.image_layer(sd, cfg, 2, xv, v_add, drop), g["image_layer2"], what="image layer")
        cv, ct = R.connection_layer(sd, cfg, 1, xv, v_add, xt, co_add, drop)
        close(cv, g["conn1_v"], what="connection v")
        close(ct[:, rows], g["conn1_t"], what="connection t")


def test_embeddings(golden_dir, full_sd3):
    cfg, sd = full_sd3
    g = np.load(os.path.join(golden_dir, "block_embeddings.npz"))
    ei = embedding_inputs(int(g["seed"]))
    drop = R._Drop(None)
    with torch.no_grad():
        et = R.text_embeddings(sd, cfg, T_(ei["ids"]), T_(ei["typ"]), T_(ei["pos"]), drop)
        ev = R.image_embeddings(sd, cfg, T_(ei["feat"]), T_(ei["loc"]), drop)
    close(et[:, ::8], g["text"], what="text embeddings")
    close(ev, g["image"], what="image embeddings")


def test_losses_and_their_gradients(golden_dir):
    g = np.load(os.path.join(golden_dir, "losses.npz"))
    li = loss_inputs(int(g["seed"]))
    assert abs(float(li["pred_t"].astype(np.float64).sum()) - float(g["pred_t_sum"])) < 1e-6
    pt = T_(li["pred_t"]).requires_grad_(True)
    pv = T_(li["pred_v"]).requires_grad_(True)
    ns = T_(li["nsp"]).requires_grad_(True)
    lm = R.mlm_ul_loss(pt, T_(li["labels"]), T_(li["weights"]))
    img = R.image_kl_loss(pv, T_(li["image_target"]), T_(li["image_label"]))
    nl = R.nsp_loss(ns, T_(li["next_sentence_label"]), torch.tensor([[5.0, 1.0]]))
    close(lm, g["lm_loss"], what="lm/ul loss")
    close(img, g["img_loss"], what="img loss")
    close(nl, g["nsp_loss"], what="nsp loss")
    (lm + img + nl).backward()
    V = pt.shape[-1]
    rows = g["d_rows"]
    close(pt.grad.reshape(-1, V)[rows][:, ::64], g["d_pred_t_rows"], tol=1e-6, what="dlogits")
    close(pt.grad.reshape(-1, V)[rows, li["labels"].reshape(-1)[rows]], g["d_pred_t_label"], tol=1e-6)
    close(pv.grad[:, :, ::16], g["d_pred_v"], tol=1e-6, what="dpred_v")
    close(ns.grad, g["d_nsp"], tol=1e-6, what="dnsp")
    with torch.no_grad():
        close(R.mlm_ul_loss(pt, T_(li["labels"]), None), g["lm_loss_ce"], what="CE fallback")
        close(R.nsp_loss(ns, T_(li["next_sentence_label"]), None), g["nsp_loss_unweighted"])


def test_full_config_b6(golden_dir):
    """BASELINE config 1: full model, 1 image x 6 sequences x 256 tokens x 37 regions."""
    g = np.load(os.path.join(golden_dir, "full_b6.npz"))
    cfg = R.make_config(os.path.join(os.path.dirname(os.path.dirname(__file__)), "unimm_amd", "config",
                                     "bert_base_6layer_6conect.json"))
    sd = R.init_state_dict(cfg, seed=5)
    i = lambda k: T_(g["in::" + k])
    n = g["in::input_ids"].shape[0]
    rep = lambda x: x.expand(n, *x.shape[1:])
    with torch.no_grad():
        out = R.forward(sd, cfg, i("input_ids"), rep(i("image_feat")), rep(i("image_loc")),
                        token_type_ids=i("token_type_ids"), position_ids=i("position_ids"),
                        attention_mask=i("attention_mask"), image_attention_mask=i("image_attention_mask"),
                        co_attention_mask=i("co_attention_mask").expand(n, 37, 256),
                        masked_lm_labels=i("masked_lm_labels"), image_label=i("image_label"),
                        image_target=rep(i("image_target")), next_sentence_label=i("next_sentence_label"),
                        nsp_weight=i("nsp_weight"), lm_weight=i("lm_weight"))
    for k in ("lm_loss", "img_loss", "nsp_loss", "nsp"):
        close(out[k], g[k], tol=5e-5, what=k)
    rows = g["rows"]
    close(out["pred_t"].reshape(-1, cfg.vocab_size)[rows][:, ::16], g["pred_t_rows"], tol=5e-5, what="pred_t")
    close(out["seq_out_t"].reshape(-1, 768)[rows], g["seq_out_t_rows"], tol=5e-5, what="seq_out_t")
    ll = R.sequence_log_likelihood(out["pred_t"], i("masked_lm_labels"))
    close(ll, g["seq_loglik"], tol=5e-5, what="sequence log-likelihood")


def test_full_config_b6_gradients(golden_dir):
    """G9: the oracle's autograd at the full config against the gradients the REFERENCE produced on the same batch
    (norms of every tensor + the sampled slices): pins the oracle as the checker of the full-size GPU backward tests."""
    from oracle.cases import grad_sample_index
    g = np.load(os.path.join(golden_dir, "full_b6.npz"))
    gg = np.load(os.path.join(golden_dir, "full_b6_grads.npz"))
    cfg = R.make_config(os.path.join(os.path.dirname(os.path.dirname(__file__)), "unimm_amd", "config",
                                     "bert_base_6layer_6conect.json"))
    sd = R.init_state_dict(cfg, seed=5)
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    i = lambda k: T_(g["in::" + k])
    n = g["in::input_ids"].shape[0]
    rep = lambda x: x.expand(n, *x.shape[1:])
    out = R.forward(leaves, cfg, i("input_ids"), rep(i("image_feat")), rep(i("image_loc")),
                    token_type_ids=i("token_type_ids"), position_ids=i("position_ids"),
                    attention_mask=i("attention_mask"), image_attention_mask=i("image_attention_mask"),
                    co_attention_mask=i("co_attention_mask").expand(n, 37, 256),
                    masked_lm_labels=i("masked_lm_labels"), image_label=i("image_label"),
                    image_target=rep(i("image_target")), next_sentence_label=i("next_sentence_label"),
                    nsp_weight=i("nsp_weight"), lm_weight=i("lm_weight"))
    (out["lm_loss"] + out["img_loss"] + out["nsp_loss"]).sum().backward()
    gmax = float(gg["grad_absmax"].max())
    checked = 0
    for name, want, am in zip(gg["grad_names"], gg["grad_norms"], gg["grad_absmax"]):
        name = str(name)
        got = leaves[name].grad
        if want < 0:
            assert got is None or float(got.abs().max()) == 0.0, name
            continue
        if am < 1e-6 * gmax:      # mathematically zero (key biases): both sides hold rounding noise only
            continue
        assert abs(float(got.double().norm()) - want) <= 2e-4 * want, (name, float(got.norm()), want)
        checked += 1
    assert checked > 450
    for k in gg.files:
        if not k.startswith("grad::"):
            continue
        name = k[6:]
        gr = leaves[name].grad
        if name.endswith("word_embeddings.weight"):
            got = gr[T_(gg["grad_rowidx::" + name])][:, ::4]
        elif gr.dim() == 1:
            got = gr[::4]
        else:
            got = gr[T_(grad_sample_index(tuple(gr.shape))[0])][:, ::4]
        if np.abs(gg[k]).max() < 1e-6 * gmax:
            continue
        err = np.abs(got.numpy() - gg[k]).max()
        assert err <= 2e-4 * np.abs(gg[k]).max(), (k, err)


def test_config_switches_sum_fusion_and_predict_feature(golden_dir):
    """fusion_method='sum' (models/vilbert_dialog.py:1062-1063) and predict_feature=True (MSE image loss, :1562-1566): the oracle's
    branches against the reference's own output (tests/golden/small_sumfeat.npz, oracle/make_goldens.py::gen_switches)."""
    cfgd = dict(json.load(open(os.path.join(golden_dir, "small_config.json"))), fusion_method="sum", predict_feature=True)
    cfg = R.make_config(cfgd)
    sd = R.init_state_dict(cfg, seed=11)
    g = np.load(os.path.join(golden_dir, "small_sumfeat.npz"))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    args, kw = oracle_kwargs(g)
    out = R.forward(leaves, cfg, *args, **kw)
    for k in ("lm_loss", "img_loss", "nsp_loss", "nsp"):
        close(out[k], g[k], what=k)
    (out["lm_loss"] + out["img_loss"] + out["nsp_loss"]).sum().backward()
    for n, want in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if want < 0:
            assert leaves[n].grad is None or float(leaves[n].grad.abs().max()) == 0.0, n
            continue
        got = float(leaves[n].grad.norm())
        assert abs(got - want) <= 2e-4 * max(want, 1e-3), (n, got, want)
    for k in g.files:
        if k.startswith("grad::"):
            close(leaves[k[6:]].grad, g[k], tol=2e-4, what=k)
    args, kw = oracle_kwargs(g, train=False)
    with torch.no_grad():
        inf = R.forward(sd, cfg, *args, **kw)
    close(inf["pred_v"], g["inf_pred_v"], what="pred_v")
    close(inf["nsp"], g["inf_nsp"], what="inf nsp")


@pytest.mark.parametrize("tag,extra", [("frozen", dict(fixed_t_layer=2)), ("nocoatt", dict(with_coattention=False))])
def test_encoder_options_frozen_text_layers_and_no_coattention(golden_dir, tag, extra):
    """fixed_t_layer (the first text layers run under no_grad, models/vilbert_dialog.py:864-869) and with_coattention=False
    (:901): the oracle's branches against the reference's own losses, scores, hidden states and gradients
    (tests/golden/small_frozen.npz / small_nocoatt.npz, oracle/make_goldens.py::gen_frozen) -- including WHICH parameters the
    reference leaves without a gradient (norm -1)."""
    cfgd = dict(json.load(open(os.path.join(golden_dir, "small_config.json"))), **extra)
    cfg = R.make_config(cfgd)
    sd = R.init_state_dict(cfg, seed=11)
    g = np.load(os.path.join(golden_dir, f"small_{tag}.npz"))
    leaves = {k: v.clone().requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    args, kw = oracle_kwargs(g)
    out = R.forward(leaves, cfg, *args, **kw)
    for k in ("lm_loss", "img_loss", "nsp_loss", "nsp"):
        close(out[k], g[k], what=k)
    (out["lm_loss"] + out["img_loss"] + out["nsp_loss"]).sum().backward()
    n_none = 0
    for n, want in zip([str(x) for x in g["grad_names"]], g["grad_norms"]):
        if want < 0:
            assert leaves[n].grad is None or float(leaves[n].grad.abs().max()) == 0.0, n
            n_none += 1
            continue
        got = float(leaves[n].grad.norm())
        assert abs(got - want) <= 2e-4 * max(want, 1e-3), (n, got, want)
    assert n_none > (30 if tag == "frozen" else 50)               # two frozen text layers + the text embeddings / both connection layers
    for k in g.files:
        if k.startswith("grad::"):
            close(leaves[k[6:]].grad, g[k], tol=2e-4, what=k)
    close(leaves[R.TIED[1]].grad[:64], g["grad_rows::word_embeddings"], tol=2e-4, what="word embedding rows")
