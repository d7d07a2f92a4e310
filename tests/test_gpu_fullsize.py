"""Full-size parity of the HIP path (bert_base_6layer_6conect: H=768 / Hv=1024 / 12+6+6 blocks, T=256, R=37).

  * the whole model's BACKWARD on BASELINE config 1 (B=6) against gradients produced by the REFERENCE's own
    modules (tests/golden/full_b6_grads.npz, oracle/make_goldens.py::gen_fullgrad): norms of all 534 tensors and
    sampled slices of one block of every type, the embeddings and the heads;
  * the engine's text / image / connection blocks and both embedding kernels at full width on the inputs of
    the reference fixtures block_layers.npz / block_embeddings.npz: forward against the fixture, backward
    against the oracle's autograd on the same inputs;
  * BASELINE config 2 (bs=240) at the full config as properties: finite losses, unpadded == padded,
    two streams == one stream, losses == the mean of the oracle's losses over B=6 chunks.

Every test prints the worst measured error of each tensor family (`pytest -s` / the captured report), so the
margin to the gates below is visible.  Gates: measured worst case + margin, never above the north_star's 1e-2
(bf16) on outputs; gradients are gated relative to the tensor's own scale (max |g|)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG_PATH = os.path.join(ROOT, "unimm_amd", "config", "bert_base_6layer_6conect.json")


def T_(x):
    return torch.from_numpy(np.ascontiguousarray(x))


def rel_to_scale(got, want):
    """max |got - want| / max |want|"""
    got = got.detach().float().cpu().numpy() if torch.is_tensor(got) else np.asarray(got)
    want = want.detach().float().cpu().numpy() if torch.is_tensor(want) else np.asarray(want)
    return float(np.abs(got.astype(np.float64) - want.astype(np.float64)).max() / max(float(np.abs(want).max()), 1e-30))


def build_full(seed, compute="bf16"):
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    model = BertForMultiModalPreTraining(BertConfig.from_json_file(CFG_PATH), compute_dtype=compute)
    ocfg = R.make_config(CFG_PATH)
    sd = R.init_state_dict(ocfg, seed=seed)
    model.load_state_dict(sd, strict=True)
    return model.cuda().eval(), ocfg, sd


def full_b6_call(g):
    i = lambda k: T_(g["in::" + k])
    n = g["in::input_ids"].shape[0]
    rep = lambda x: x.expand(n, *x.shape[1:])
    args = (i("input_ids"), rep(i("image_feat")), rep(i("image_loc")))
    kw = dict(token_type_ids=i("token_type_ids"), position_ids=i("position_ids"), attention_mask=i("attention_mask"),
              image_attention_mask=i("image_attention_mask"), co_attention_mask=i("co_attention_mask").expand(n, 37, 256),
              masked_lm_labels=i("masked_lm_labels"), image_label=i("image_label"), image_target=rep(i("image_target")),
              next_sentence_label=i("next_sentence_label"), nsp_weight=i("nsp_weight"), lm_weight=i("lm_weight"))
    return args, kw


# Gates of the full-size backward (set from the measured worst cases printed below, plus margin).
GRAD_NORM_GATE = 2e-2        # | ||g_hip|| - ||g_ref|| | / ||g_ref||, every tensor whose gradient is not numerically zero
#                              (round 3, poolers + NSP head in fp32: families 0.05-1.4 %; the 2-element NSP bias, which
#                              needed its own 5 % gate while the head ran on bf16 operands, is at 0.15 %)
GRAD_SLICE_L2_GATE = 1e-1    # ||g_hip - g_ref||_2 / ||g_ref||_2 on the sampled slices
GRAD_SLICE_MAX_GATE = 1.5e-1  # max |g_hip - g_ref| / max |g_ref| on the sampled slices
# The two poolers are ReLU(W h + b) on the first-token row of 6 sequences (models/vilbert_dialog.py:946-967): 6 x 1024
# units.  Round 2 blamed bf16 pooler arithmetic for their 8-16 % slice errors; round 3 moved the poolers, the fused product
# and the NSP head to exact fp32 (unimm_linear_f32, fp32 master weights, fp32 residual-stream input) forward and backward
# -- the gradient NORMS of these tensors went from 1-3 % to 0.15-0.4 % -- but the sampled-slice error did not move
# (10.7 % / 28.8 % before and after), because the switching units are decided by the pooler's INPUT: z = W h + b has
# std ~0.55 rms(h) (W ~ N(0, 0.02), 768 inputs), the encoder output h carries the bf16 noise of 24 blocks (measured 0.5 %
# of scale, test_gpu_model.py), so dz ~ 0.003 rms(h) and P(|z| < dz) ~ 0.4 % of the units = ~25 of 6,144 sit within the
# noise of zero.  Each of those gains or loses its WHOLE gradient, so ~2.4 % of the elements of a pooler gradient are off
# by O(1) of themselves: sqrt(0.024) = 15 % L2, worst element up to its full size.  That is a property of bf16 operands
# anywhere below the poolers, not of the poolers; only an fp32 encoder removes it.  The image side of the LAST connection
# block sits directly under the image pooler and inherits it (measured 11.8 % / 13 %; its text side 0.9 %).
TOP_BLOCK = "bert.encoder.c_layer.5"
POOLER_GATES = (2e-1, 3.5e-1)
# Measured (the test prints the table): heads 0.6-1.5 %, encoder blocks 4-6.8 % L2 (worst element up to 9.5 % of the
# tensor's largest on the image side, whose gradients average only 6 x 37 rows), embeddings 5.7 %, all with the tensor
# NORMS within 1.4 % (most within 0.5 %): errors orthogonal to the signal, i.e. noise.  That is the bf16 floor of this depth:
# every block perturbs the backward signal through its bf16 GEMM operands, bf16 attention probabilities and the bf16
# gradient stream between blocks (2^-9 per rounding), ~0.5-1 % per block and uncorrelated, ~sqrt(24) x 1 %.


def test_full_config_b6_backward_matches_reference_golden(golden_dir):
    """models/vilbert_dialog.py:1519-1624 + autograd, full config, BASELINE config 1."""
    from oracle.cases import grad_sample_index
    g = np.load(os.path.join(golden_dir, "full_b6.npz"))
    gg = np.load(os.path.join(golden_dir, "full_b6_grads.npz"))
    model, _, _ = build_full(seed=5)
    args, kw = full_b6_call(g)
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, _ = model(*args, **kw, _want_lm_scores=False)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    for name, got in (("lm_loss", lm), ("img_loss", img), ("nsp_loss", nsp_l)):
        assert abs(float(got.detach()) - float(gg[name].item())) <= 1e-2 * (1 + abs(float(gg[name].item()))), name
    params = dict(model.named_parameters())
    names = [str(n) for n in gg["grad_names"]]
    norms, amax = gg["grad_norms"], gg["grad_absmax"]
    gmax = float(amax.max())
    worst = {}
    checked = 0
    for n, want, am in zip(names, norms, amax):
        p = params[n]
        if want < 0:
            assert p.grad is None, f"{n} is never used by forward: grad must stay None"
            continue
        got = float(p.grad.double().norm())
        fam = n.split(".")[2] if n.startswith("bert.encoder") else n.split(".")[0] + "." + n.split(".")[1]
        if am < 1e-6 * gmax:
            # mathematically zero gradients (key biases: softmax is shift-invariant): only bounded, not compared
            assert got <= 1e-3 * float(norms.max()), (n, got)
            continue
        rel = abs(got - want) / want
        worst[fam] = max(worst.get(fam, 0.0), rel)
        assert rel <= GRAD_NORM_GATE, (n, got, want, rel)
        checked += 1
    assert checked > 450
    print("\nfull-config backward: worst relative error of a gradient norm, per tensor family")
    for k, v in sorted(worst.items()):
        print(f"  {k:32s} {v:.3e}")
    worst_s = {}
    nslices = 0
    for k in gg.files:
        if not k.startswith("grad::"):
            continue
        n = k[6:]
        want = gg[k]
        gr = params[n].grad
        if n.endswith("word_embeddings.weight"):
            got = gr[T_(gg["grad_rowidx::" + n]).cuda()][:, ::4]
        elif gr.dim() == 1:
            got = gr[::4]
        else:
            got = gr[T_(grad_sample_index(tuple(gr.shape))[0]).cuda()][:, ::4]
        if np.abs(want).max() < 1e-6 * gmax:
            continue
        r = rel_to_scale(got, want)
        gn = got.detach().double().cpu().numpy()
        l2 = float(np.linalg.norm(gn - want) / max(np.linalg.norm(want), 1e-30))
        fam = ".".join(n.split(".")[:4]) if n.startswith("bert.encoder") else ".".join(n.split(".")[:2])
        w = worst_s.setdefault(fam, [0.0, 0.0, ""])
        if l2 > w[0]:
            w[2] = n
        w[0], w[1] = max(w[0], l2), max(w[1], r)
        nslices += 1
    assert nslices > 100
    print("full-config backward: worst error on the sampled slices, per block:  ||err||2/||g||2   max|err|/max|g|")
    for k, v in sorted(worst_s.items()):
        print(f"  {k:40s} {v[0]:.3e}   {v[1]:.3e}   ({v[2].split('.', 4)[-1] if k.startswith('bert.encoder') else v[2]})")
    bad = {k: v for k, v in worst_s.items()
           if v[0] > (POOLER_GATES[0] if ("pooler" in k or k == TOP_BLOCK) else GRAD_SLICE_L2_GATE)
           or v[1] > (POOLER_GATES[1] if ("pooler" in k or k == TOP_BLOCK) else GRAD_SLICE_MAX_GATE)}
    assert not bad, bad


# ------------------------------------------------------------------------------------------------------
# single blocks at full width
# ------------------------------------------------------------------------------------------------------
@pytest.fixture(scope="module")
def full3():
    model, ocfg, sd = build_full(seed=3)
    eng = model.engine
    eng.ensure(torch.device("cuda", 0))
    eng.refresh_weights(force=True)
    eng.arena.attach_grads()
    return model, eng, ocfg, sd


def _leaves(sd, prefix):
    return {k: (v.clone().requires_grad_(True) if k.startswith(prefix) else v) for k, v in sd.items()}


def _check_param_grads(eng, leaves, prefix, gate, what):
    mine = {k: v for k, v in leaves.items() if k.startswith(prefix) and v.grad is not None and "q_dense" not in k}
    top = max(float(v.grad.abs().max()) for v in mine.values())
    worst = 0.0
    for k, v in mine.items():
        want = v.grad
        if float(want.abs().max()) < 1e-4 * top:      # mathematically zero gradients (key biases): rounding noise only
            assert float(eng.arena.grad(k).abs().max()) <= 1e-2 * top, (what, k)
            continue
        r = rel_to_scale(eng.arena.grad(k), want)
        worst = max(worst, r)
        assert r <= gate, (what, k, r)
    print(f"  {what}: worst parameter-gradient error (max|err|/max|g|) {worst:.3e}")


BLOCK_FWD_GATE = 1e-2     # |err| <= 1e-2 + 1e-2 |want| (north_star, bf16)
BLOCK_BWD_GATE = 2e-2     # max |err| / max |g|


def _close(got, want, what):
    got = got.detach().float().cpu().numpy()
    want = np.asarray(want, dtype=np.float64)
    diff = np.abs(got - want)
    excess = (diff - (BLOCK_FWD_GATE + BLOCK_FWD_GATE * np.abs(want))).max()
    print(f"  {what}: max |err| {diff.max():.3e} (values up to {np.abs(want).max():.2f})")
    assert excess <= 0, (what, float(diff.max()), float(excess))


def test_full_size_text_and_image_layer_fwd_bwd(golden_dir, full3):
    """BertLayer / BertImageLayer (models/vilbert_dialog.py:385-483, :514-612) at H=768x12 heads / Hv=1024x8 heads:
    forward against the reference fixture, backward against the oracle's autograd."""
    from oracle import vilbert_ref as R
    from oracle.cases import block_inputs
    from unimm_amd import lib as L
    model, eng, ocfg, sd = full3
    cfg = model.config
    g = np.load(os.path.join(golden_dir, "block_layers.npz"))
    bi = block_inputs(int(g["seed"]))
    rows = bi["rows"]
    dev = torch.device("cuda", 0)
    B, T, Rg = 2, 256, 37
    rng = np.random.Generator(np.random.PCG64(123))
    print()
    for kind in ("text", "image"):
        if kind == "text":
            key, pfx, x_np, mask_np, n, heads, li = "t3", "bert.encoder.layer.3.", bi["xt"], bi["tmask"], T, cfg.num_attention_heads, 3
        else:
            key, pfx, x_np, mask_np, n, heads, li = "v2", "bert.encoder.v_layer.2.", bi["xv"], bi["vmask"], Rg, cfg.v_num_attention_heads, 2
        Hd = x_np.shape[-1]
        eng.arena.zero_grads()
        eng._dev_masks = []
        mask = eng._pack_mask(T_(mask_np), dev, n)
        x32 = T_(x_np).reshape(B * n, Hd).to(dev)
        x16 = x32.to(torch.bfloat16)
        st = dict(train=False, tape=[])
        y32, y16 = eng._self_block(key, x32, x16, mask, B, n, heads, pfx, 0.1, 0.1, st)
        y32 = eng._dense32(y32).view(B, n, Hd)
        want = g["text_layer3"] if kind == "text" else g["image_layer2"]
        _close(y32[:, T_(rows).to(dev)] if kind == "text" else y32, want, f"{kind} layer forward")
        # backward: upstream gradient on the valid rows only (pad rows of the fixture masks never reach a loss)
        dy = rng.standard_normal((B, n, Hd)).astype(np.float32) * 0.05
        valid = (mask_np.any(-1) if mask_np.ndim == 3 else mask_np > 0)
        dy *= valid[:, :, None]
        dy_t = T_(dy)
        dx = st["tape"][-1][1](dy_t.reshape(B * n, Hd).to(dev).to(torch.bfloat16))
        eng._flush_wgrad(force=True)
        torch.cuda.synchronize()
        leaves = _leaves(sd, pfx)
        xr = T_(x_np).clone().requires_grad_(True)
        if kind == "text":
            add = R.additive(T_(mask_np))[:, None]
            yo = R.text_layer(leaves, ocfg, li, xr, add, R._Drop(None))
        else:
            add = R.additive(T_(mask_np))[:, None, None, :]
            yo = R.image_layer(leaves, ocfg, li, xr, add, R._Drop(None))
        (yo * dy_t.to(torch.bfloat16).float()).sum().backward()
        r = rel_to_scale(dx.view(B, n, Hd)[T_(valid).to(dev)], xr.grad[T_(valid)])
        print(f"  {kind} layer backward: input-gradient error (max|err|/max|g|) {r:.3e}")
        assert r <= BLOCK_BWD_GATE, (kind, r)
        _check_param_grads(eng, leaves, pfx, BLOCK_BWD_GATE, f"{kind} layer")


def test_full_size_connection_layer_fwd_bwd(golden_dir, full3):
    """BertConnectionLayer (models/vilbert_dialog.py:655-783): both co-attention directions, bi-output, both FFNs."""
    from oracle import vilbert_ref as R
    from oracle.cases import block_inputs
    model, eng, ocfg, sd = full3
    g = np.load(os.path.join(golden_dir, "block_layers.npz"))
    bi = block_inputs(int(g["seed"]))
    rows = bi["rows"]
    dev = torch.device("cuda", 0)
    B, T, Rg, H, Hv = 2, 256, 37, 768, 1024
    pfx = "bert.encoder.c_layer.1."
    eng.arena.zero_grads()
    eng._dev_masks = []
    vmask = eng._pack_mask(T_(bi["vmask"]), dev, Rg)
    comask = eng._pack_mask(T_(bi["co"]), dev, Rg)
    xv32 = T_(bi["xv"]).reshape(B * Rg, Hv).to(dev)
    xt32 = T_(bi["xt"]).reshape(B * T, H).to(dev)
    st = dict(train=False, tape=[])
    ov32, ov, ot32, ot = eng._conn_block("c1", 1, xv32, xv32.to(torch.bfloat16), xt32, xt32.to(torch.bfloat16), B, Rg, T,
                                         vmask, comask, st)
    eng._to_txt(ov32, ov)
    print()
    _close(eng._dense32(ov32).view(B, Rg, Hv), g["conn1_v"], "connection layer forward (image side)")
    _close(eng._dense32(ot32).view(B, T, H)[:, T_(rows).to(dev)], g["conn1_t"], "connection layer forward (text side)")
    rng = np.random.Generator(np.random.PCG64(321))
    dv = (rng.standard_normal((B, Rg, Hv)) * 0.05).astype(np.float32)
    dt = (rng.standard_normal((B, T, H)) * 0.05).astype(np.float32)
    dv_t, dt_t = T_(dv), T_(dt)
    gv = dv_t.reshape(B * Rg, Hv).to(dev).to(torch.bfloat16)
    gt = dt_t.reshape(B * T, H).to(dev).to(torch.bfloat16)
    eng._to_img(gv)
    dxv, dxt = st["tape"][-1][1](gv, gt)
    eng._flush_wgrad(force=True)
    eng._to_txt(dxv)
    torch.cuda.synchronize()
    leaves = _leaves(sd, pfx)
    xvr = T_(bi["xv"]).clone().requires_grad_(True)
    xtr = T_(bi["xt"]).clone().requires_grad_(True)
    v_add = R.additive(T_(bi["vmask"]))[:, None, None, :]
    co_add = R.additive(T_(bi["co"])).unsqueeze(1)
    cv, ct = R.connection_layer(leaves, ocfg, 1, xvr, v_add, xtr, co_add, R._Drop(None))
    ((cv * dv_t.to(torch.bfloat16).float()).sum() + (ct * dt_t.to(torch.bfloat16).float()).sum()).backward()
    rv = rel_to_scale(dxv.view(B, Rg, Hv), xvr.grad)
    rt = rel_to_scale(dxt.view(B, T, H), xtr.grad)
    print(f"  connection layer backward: input-gradient error image {rv:.3e} text {rt:.3e}")
    assert rv <= BLOCK_BWD_GATE and rt <= BLOCK_BWD_GATE, (rv, rt)
    _check_param_grads(eng, leaves, pfx, BLOCK_BWD_GATE, "connection layer")


def test_full_size_embeddings_fwd_bwd(golden_dir, full3):
    """BertEmbeddingsDialog incl. type ids >= 2 and BertImageEmbeddings (models/vilbert_dialog.py:326-356, :1487-1493)
    through the whole-model entry (they are fused into Engine._forward), on the inputs of block_embeddings.npz."""
    from oracle import vilbert_ref as R
    from oracle.cases import embedding_inputs
    from unimm_amd import lib as L
    model, eng, ocfg, sd = full3
    g = np.load(os.path.join(golden_dir, "block_embeddings.npz"))
    ei = embedding_inputs(int(g["seed"]))
    dev = torch.device("cuda", 0)
    B, T, Rg, H, Hv = 2, 256, 37, 768, 1024
    cfg = model.config
    eng.arena.zero_grads()
    # text: the embedding kernel directly (the model entry would also run the encoder)
    ids32, pos32, typ32 = (T_(ei[k].reshape(-1)).to(dev, dtype=torch.int32) for k in ("ids", "pos", "typ"))
    gmm, bta, ggm, gbt = eng.ln["emb_t"]
    xt = torch.empty((B * T, H), dtype=torch.bfloat16, device=dev)
    xt32 = torch.empty((B * T, H), dtype=torch.float32, device=dev)
    tabs = (eng.tab["word"], eng.tab["pos"], eng.tab["type"], eng.tab["ext"])
    L.embed_fwd(ids32, pos32, typ32, *tabs, gmm, bta, xt32, xt, B * T, H, cfg.type_vocab_size, drop=L.NO_DROP)
    print()
    _close(xt32.view(B, T, H)[:, ::8], g["text"], "text embeddings forward")
    rng = np.random.Generator(np.random.PCG64(77))
    dy = (rng.standard_normal((B, T, H)) * 0.05).astype(np.float32)
    dy_t = T_(dy)
    A = eng.arena
    e = "bert.embeddings."
    L.embed_bwd(ids32, pos32, typ32, *tabs, gmm, bta, dy_t.reshape(B * T, H).to(dev).to(torch.bfloat16),
                A.grad(e + "word_embeddings.weight"), A.grad(e + "position_embeddings.weight"),
                A.grad(e + "token_type_embeddings.weight"), A.grad(e + "token_type_embeddings_extension.weight"), ggm, gbt,
                eng.part[H], B * T, H, cfg.type_vocab_size, drop=L.NO_DROP)
    torch.cuda.synchronize()
    leaves = _leaves(sd, e)
    et = R.text_embeddings(leaves, ocfg, T_(ei["ids"]), T_(ei["typ"]), T_(ei["pos"]), R._Drop(None))
    (et * dy_t.to(torch.bfloat16).float()).sum().backward()
    _check_param_grads(eng, {k: v for k, v in leaves.items() if "sep_embeddings" not in k}, e, BLOCK_BWD_GATE, "text embeddings")
    # image: pack + one GEMM + LayerNorm, as Engine._forward runs them
    F = cfg.v_feature_size
    feat = T_(ei["feat"]).reshape(B * Rg, F).to(dev)
    loc = T_(ei["loc"]).reshape(B * Rg, 5).to(dev)
    packed = torch.empty((B * Rg, eng.vemb_k), dtype=torch.bfloat16, device=dev)
    L.pack_image(feat, loc, packed, B * Rg, F, eng.vemb_k)
    prev = torch.empty((B * Rg, Hv), dtype=torch.float32, device=dev)
    L.gemm_nt(packed, eng.vemb_w, prev, bias=eng.vemb_b, M=B * Rg, N=Hv, K=eng.vemb_k)
    xv32, xv, mv, rv = eng._layernorm(prev, "emb_v", True)
    _close(xv32.view(B, Rg, Hv), g["image"], "image embeddings forward")


# ------------------------------------------------------------------------------------------------------
# BASELINE config 2 at the full config: properties at bs=240
# ------------------------------------------------------------------------------------------------------
def test_bs240_full_config_properties(golden_dir):
    """240 sequences x 256 tokens x 37 regions, the bench workload (dropout off so that runs are comparable):
    losses finite; the unpadded schedule == the padded one; two streams == one stream (bit-identical losses);
    every parameter gradient finite; and the losses of two B=6 chunks == the CPU oracle's on the same rows."""
    from oracle import vilbert_ref as R
    from unimm_amd import synth
    model, ocfg, sd = build_full(seed=5)
    eng = model.engine
    b = synth.make_batch(n_seq=240, T=256, R=37, cfg=model.config, seed=1234, sequences_per_image=6, device="cuda")

    def kwargs(sl=slice(None), dev=None):
        mv = (lambda t: t[sl].to(dev)) if dev else (lambda t: t[sl])
        return (mv(b["input_ids"]), mv(b["image_feat"]), mv(b["image_loc"])), dict(
            token_type_ids=mv(b["token_type_ids"]), position_ids=mv(b["token_position_ids"]), attention_mask=mv(b["attention_mask"]),
            co_attention_mask=mv(b["co_attention_mask"]), image_attention_mask=mv(b["image_attention_mask"]),
            masked_lm_labels=mv(b["masked_lm_labels"]), image_label=mv(b["image_label"]), image_target=mv(b["image_target"]),
            next_sentence_label=mv(b["next_sentence_label"]), nsp_weight=b["nsp_weight"].to(dev) if dev else b["nsp_weight"],
            lm_weight=mv(b["lm_weight"]))

    def run(unpad, dual):
        eng.unpad, eng.dual_stream = unpad, dual
        model.zero_grad(set_to_none=True)
        args, kw = kwargs()
        lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
        (lm + img + nsp_l).sum().backward()
        torch.cuda.synchronize()
        return torch.stack([lm, img, nsp_l]).flatten().detach().clone(), nsp.detach().clone(), eng.arena.grad_flat.clone()

    was = (eng.unpad, eng.dual_stream)
    try:
        base = run(True, True)
        padded = run(False, True)
        single = run(True, False)
    finally:
        eng.unpad, eng.dual_stream = was
    assert torch.isfinite(base[0]).all() and torch.isfinite(base[2]).all()
    print(f"\nbs=240 losses (lm, img, nsp): {[round(float(x), 4) for x in base[0]]}")
    d_loss = float((base[0] - padded[0]).abs().max())
    d_nsp = float((base[1] - padded[1]).abs().max())
    d_grad = float((base[2] - padded[2]).abs().max() / padded[2].abs().max())
    print(f"  unpadded vs padded: losses {d_loss:.2e}, nsp {d_nsp:.2e}, gradients {d_grad:.2e} of max|g|")
    assert d_loss <= 2e-3 and d_nsp <= 2e-3 and d_grad <= 1e-2
    assert torch.equal(base[0], single[0]) and torch.equal(base[1], single[1])      # same kernels, same words
    d2 = float((base[2] - single[2]).abs().max() / single[2].abs().max())
    print(f"  two streams vs one: losses bit-identical, gradients {d2:.2e} of max|g| (atomics order)")
    assert d2 <= 1e-4
    # two B=6 chunks against the oracle (mean-of-chunk losses is what the data-parallel split computes)
    leaves = dict(sd)
    tol = 1e-2                                   # the bf16 engine's class tolerance (north_star: 1e-2 bf16), allclose form
    for c in (0, 17):
        sl = slice(6 * c, 6 * c + 6)
        args, kw = kwargs(sl)
        with torch.no_grad():
            got = model(*args, **kw, _want_lm_scores=False)
            cargs, ckw = kwargs(sl, dev="cpu")
            want = R.forward(leaves, ocfg, *cargs, **ckw)
        for name, gi in (("lm_loss", 0), ("img_loss", 1), ("nsp_loss", 2)):
            e = abs(float(got[gi]) - float(want[name]))
            print(f"  chunk {c}: {name} hip {float(got[gi]):.5f} oracle {float(want[name]):.5f}  (|err| {e:.2e}, gate {tol:.0e} allclose)")
            assert e <= tol * (1 + abs(float(want[name]))), (c, name, e)
        err = (got[5].cpu() - want["nsp"]).abs()
        assert bool((err <= tol + tol * want["nsp"].abs()).all()), (c, "nsp", float(err.max()))      # every NSP logit, allclose form
        print(f"  chunk {c}: NSP logits max |err| {float(err.max()):.2e}")


# ------------------------------------------------------------------------------------------------------
# BASELINE configs[3] and [4] at the full config: the two workloads that were bench-only until round 3
# ------------------------------------------------------------------------------------------------------
def _call_kwargs(b, sl=slice(None), dev=None, labels=True):
    mv = (lambda t: t[sl].to(dev)) if dev else (lambda t: t[sl])
    kw = dict(token_type_ids=mv(b["token_type_ids"]), position_ids=mv(b["token_position_ids"]), attention_mask=mv(b["attention_mask"]),
              co_attention_mask=mv(b["co_attention_mask"]), image_attention_mask=mv(b["image_attention_mask"]))
    if labels:
        kw.update(masked_lm_labels=mv(b["masked_lm_labels"]), image_label=mv(b["image_label"]), image_target=mv(b["image_target"]),
                  next_sentence_label=mv(b["next_sentence_label"]),
                  nsp_weight=b["nsp_weight"].to(dev) if dev else b["nsp_weight"], lm_weight=mv(b["lm_weight"]))
    return (mv(b["input_ids"]), mv(b["image_feat"]), mv(b["image_loc"])), kw


# Oracle gates of the two workloads, per arithmetic class: |err| <= tol + tol * |want| (north_star: 1e-2 bf16 / 1e-3 fp32).
# configs[3] is the reference's fp32 script (dense_annotation_finetuning.py:253 runs without autocast), so its own class is fp32x3.
CLASS_TOL = {"bf16": 1e-2, "fp32x3": 1e-3}


@pytest.mark.parametrize("compute", ["bf16", "fp32x3"])
def test_dense_finetune_b100_full_config_properties(compute):
    """BASELINE configs[3] (dense_annotation_finetuning.py:253-296): one micro-step of 100 discriminative sequences,
    2 per image, objective = NeuralNDCG^T over the 100 options + LM loss + 0 x NSP, full config, dropout off:
    finite; unpadded == padded; two streams == one; the ranking term equals the PyTorch (CPU-formulation) path of
    unimm_amd.ranking on the same scores; losses and NSP logits of two 6-row chunks == the CPU oracle in allclose form at the
    class tolerance (bf16 engine 1e-2; fp32x3 engine -- the reference's own arithmetic for this script -- 1e-3)."""
    from oracle import vilbert_ref as R
    from unimm_amd import ranking, synth
    tol = CLASS_TOL[compute]
    model, ocfg, sd = build_full(seed=5, compute=compute)
    eng = model.engine
    n = 100
    b = synth.make_batch(n_seq=n, T=256, R=37, cfg=model.config, seed=4321, sequences_per_image=2, device="cuda", modes=["dis"] * n)
    g = torch.Generator().manual_seed(77)
    relevance = torch.tensor([0, 0, 0, 0, 0.2, 0.4, 0.6, 1.0])[torch.randint(0, 8, (1, n), generator=g)].cuda()

    def run(unpad, dual):
        eng.unpad, eng.dual_stream = unpad, dual
        model.zero_grad(set_to_none=True)
        args, kw = _call_kwargs(b)
        lm, img, nsp_l, _, _, nsp = model(*args, **kw, _want_lm_scores=False)
        loss, parts = ranking.dense_finetune_loss(nsp, b["next_sentence_label"], relevance, lm, 0.0, num_options=n)
        loss.backward()
        torch.cuda.synchronize()
        return (torch.stack([loss.detach().reshape(()), lm.detach().reshape(()), parts["target"].detach().reshape(())]).clone(),
                nsp.detach().clone(), eng.arena.grad_flat.clone())

    was = (eng.unpad, eng.dual_stream)
    try:
        base = run(True, True)
        padded = run(False, True)
        single = run(True, False)
    finally:
        eng.unpad, eng.dual_stream = was
    assert torch.isfinite(base[0]).all() and torch.isfinite(base[2]).all()
    print(f"\ndense b=100 (loss, lm, NeuralNDCG^T term): {[round(float(x), 4) for x in base[0]]}")
    d_loss = float((base[0] - padded[0]).abs().max())
    d_nsp = float((base[1] - padded[1]).abs().max())
    d_grad = float((base[2] - padded[2]).abs().max() / padded[2].abs().max())
    print(f"  unpadded vs padded: losses {d_loss:.2e}, nsp {d_nsp:.2e}, gradients {d_grad:.2e} of max|g|")
    assert d_loss <= 2e-3 and d_nsp <= 2e-3 and d_grad <= 1e-2
    if compute == "fp32x3":                                            # fp32 arithmetic: padding rows change nothing beyond rounding
        assert d_loss <= 2e-5 and d_nsp <= 2e-5 and d_grad <= 1e-4, (d_loss, d_nsp, d_grad)
    assert torch.equal(base[0], single[0]) and torch.equal(base[1], single[1])
    d2 = float((base[2] - single[2]).abs().max() / single[2].abs().max())
    print(f"  two streams vs one: losses bit-identical, gradients {d2:.2e} of max|g| (atomics order)")
    assert d2 <= 1e-4
    # the fused NeuralNDCG^T kernel against the PyTorch formulation (the one tests/test_ranking_cpu.py pins to the
    # reference's utils/rank_loss.py) on this step's scores, evaluated on the CPU
    p_answer = torch.softmax(base[1].float().view(1, n, 2), dim=-1)[:, :, 0].cpu()
    want = ranking.neuralNDCG_transposed_torch(p_answer, relevance.float().cpu())
    e = abs(float(base[0][2]) - float(want))
    print(f"  NeuralNDCG^T: kernel {float(base[0][2]):.6f} torch/CPU {float(want):.6f}")
    assert e <= 1e-4 * max(1.0, abs(float(want))), e
    for c in (0, 11):
        sl = slice(6 * c, 6 * c + 6)
        args, kw = _call_kwargs(b, sl)
        with torch.no_grad():
            got = model(*args, **kw, _want_lm_scores=False)
            cargs, ckw = _call_kwargs(b, sl, dev="cpu")
            want = R.forward(dict(sd), ocfg, *cargs, **ckw)
        for name, gi in (("lm_loss", 0), ("img_loss", 1), ("nsp_loss", 2)):
            e = abs(float(got[gi]) - float(want[name]))
            print(f"  chunk {c}: {name} hip {float(got[gi]):.5f} oracle {float(want[name]):.5f}  (|err| {e:.2e}, gate {tol:.0e} allclose)")
            assert e <= tol * (1 + abs(float(want[name]))), (c, name, e)
        err = (got[5].cpu() - want["nsp"]).abs()
        assert bool((err <= tol + tol * want["nsp"].abs()).all()), (c, "nsp", float(err.max()))      # every NSP logit, allclose form
        print(f"  chunk {c}: NSP logits max |err| {float(err.max()):.2e}")


@pytest.mark.parametrize("compute", ["bf16", "fp32x3"])
def test_generative_scoring_chunk250_full_config_properties(compute):
    """BASELINE configs[4] (val_lm.py:121-149): one chunk of 250 generative candidate sequences, full config: the
    sequence log-likelihoods (decoded on the labelled rows only) are finite, the unpadded schedule == the padded one,
    12 sampled sequences == the oracle's dense-logits cross entropy summed per sequence, and the ranks the scores
    induce agree with the oracle's wherever the oracle's margin between two candidates exceeds the tolerance
    (bf16 engine: 1e-2 of the largest |log-likelihood|; fp32x3 engine: allclose form at 1e-3 and IDENTICAL ranks)."""
    from oracle import vilbert_ref as R
    from unimm_amd import synth
    from unimm_amd.harness import scores_to_ranks
    model, ocfg, sd = build_full(seed=5, compute=compute)
    eng = model.engine
    n = 250
    b = synth.make_batch(n_seq=n, T=256, R=37, cfg=model.config, seed=999, sequences_per_image=250, device="cuda",
                         modes=["gen"] * n, mask_prob=0.0)
    args, kw = _call_kwargs(b, labels=False)
    res = {}
    was = eng.unpad
    try:
        for unpad in (True, False):
            eng.unpad = unpad
            scores, _ = model.sequence_log_likelihood(*args, b["masked_lm_labels"], **kw)
            torch.cuda.synchronize()
            res[unpad] = scores.clone()
    finally:
        eng.unpad = was
    got = res[True]
    assert torch.isfinite(got).all() and (got < 0).all()
    d = float((res[True] - res[False]).abs().max())
    print(f"\nscoring chunk of 250: log-likelihoods {float(got.min()):.2f} .. {float(got.max()):.2f}; unpadded vs padded {d:.2e}")
    assert d <= 2e-3 * float(got.abs().max())
    pick = list(range(0, n, 21))[:12]
    want = []
    for i in pick:                                   # the oracle, one sequence at a time (dense logits: [1, 256, 30522])
        cargs, ckw = _call_kwargs(b, slice(i, i + 1), dev="cpu", labels=False)
        with torch.no_grad():
            o = R.forward(dict(sd), ocfg, *cargs, **ckw)
            lab = b["masked_lm_labels"][i:i + 1].cpu()
            nll = torch.nn.functional.cross_entropy(o["pred_t"].view(-1, o["pred_t"].shape[-1]), lab.view(-1), ignore_index=-1,
                                                    reduction="none").view(1, -1)
        want.append(float(-nll.sum()))
    want = torch.tensor(want)
    sel = got[pick].cpu()
    tol = 1e-2 * float(want.abs().max())
    print(f"  12 sequences vs oracle: max |err| {float((sel - want).abs().max()):.3e} (values {float(want.min()):.2f} .. {float(want.max()):.2f}, gate {tol:.3e})")
    assert float((sel - want).abs().max()) <= tol
    if compute == "fp32x3":
        assert bool(((sel - want).abs() <= 1e-3 + 1e-3 * want.abs()).all()), (sel - want).abs().max()      # allclose form, every sequence
        assert scores_to_ranks(sel.view(1, 1, -1)).view(-1).tolist() == scores_to_ranks(want.view(1, 1, -1)).view(-1).tolist()
    # ranks among the 12 sampled candidates: every pair the oracle separates by more than 2 x tol keeps its order
    r_got = scores_to_ranks(sel.view(1, 1, -1)).view(-1)
    r_want = scores_to_ranks(want.view(1, 1, -1)).view(-1)
    for a_ in range(12):
        for c_ in range(12):
            if want[a_] - want[c_] > 2 * tol:
                assert r_got[a_] < r_got[c_], (a_, c_, float(want[a_]), float(want[c_]), float(sel[a_]), float(sel[c_]))
    assert sorted(r_got.tolist()) == list(range(1, 13)) and sorted(r_want.tolist()) == list(range(1, 13))


def test_generative_scoring_shared_context_full_config():
    """val_lm.py:52-121 scores 100 candidates per dialog round that share image, history and question.  With
    `shared_context=<round index>` the context rows and the image stream are computed once per round
    (unimm_amd/scoring.py: under the generative mask they never see the candidate, utils/data_utils.py:199-210).  One
    chunk of 250 = 2.5 rounds at the full config: scores == the per-sequence path within 2e-3 of the largest |score|, NSP
    logits likewise, ranks inside every round identical wherever two candidates are further apart than that tolerance
    (and >= 97 % identical outright: candidates closer than the bf16 noise of the two schedules, ~0.03 on scores of
    -30 .. -140, may trade places); 12 sampled sequences == the oracle at 1e-2; mask descriptors == dense masks; a sequence whose context does NOT match its group comes back as NaN and nothing else changes."""
    from oracle import vilbert_ref as R
    from unimm_amd import synth
    from unimm_amd.harness import scores_to_ranks
    model, ocfg, sd = build_full(seed=5)
    full = synth.make_scoring_batch(rounds=3, options=100, cfg=model.config, seed=2024, device="cuda")
    n = 250
    spec_full = full.pop("mask_spec")
    b = {k: v[:n] for k, v in full.items()}
    grp = b["context_group"]
    args = (b["input_ids"], b["image_feat"], b["image_loc"], b["masked_lm_labels"])
    kw = dict(token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"], attention_mask=b["attention_mask"],
              co_attention_mask=b["co_attention_mask"], image_attention_mask=b["image_attention_mask"])
    base, nsp0 = model.sequence_log_likelihood(*args, **kw)
    got, nsp1 = model.sequence_log_likelihood(*args, shared_context=grp, **kw)
    torch.cuda.synchronize()
    assert torch.isfinite(got).all() and (got < 0).all()
    scale = float(base.abs().max())
    d = float((got - base).abs().max())
    dn = float((nsp1 - nsp0).abs().max())
    print(f"\nshared-context scoring, 250 sequences = 2.5 rounds: scores {float(got.min()):.2f} .. {float(got.max()):.2f}; "
          f"vs the per-sequence path {d:.3e} ({d / scale:.2e} of scale), NSP logits {dn:.3e}")
    assert d <= 2e-3 * scale and dn <= 2e-2 * (1 + float(nsp0.abs().max()))
    tol = 2e-3 * scale
    same_rank = total = 0
    for r0, r1 in ((0, 100), (100, 200), (200, 250)):
        ra = scores_to_ranks(base[r0:r1].view(1, 1, -1)).view(-1)
        rb = scores_to_ranks(got[r0:r1].view(1, 1, -1)).view(-1)
        same_rank += int((ra == rb).sum())
        total += r1 - r0
        s0 = base[r0:r1]
        far = (s0[:, None] - s0[None, :]) > tol                       # pairs the per-sequence path separates clearly
        assert bool((rb[:, None] < rb[None, :])[far].all())
    print(f"  ranks identical for {same_rank} of {total} candidates")
    assert same_rank >= 0.97 * total
    # mask descriptors instead of dense masks: the same packed words, the same result
    from unimm_amd.inputs import DialogMaskSpec
    spec = DialogMaskSpec(spec_full.mode[:n], spec_full.length[:n], spec_full.answer[:n])
    kw2 = dict(kw, attention_mask=spec, co_attention_mask=None)
    got2, _ = model.sequence_log_likelihood(*args, shared_context=grp, **kw2)
    assert float((got - got2).abs().max()) <= 1e-4               # the same packed words; the per-sequence sums are fp32 atomics (order)
    # the oracle, one sequence at a time
    pick = list(range(3, n, 21))[:12]
    want = []
    for i in pick:
        with torch.no_grad():
            o = R.forward(dict(sd), ocfg, b["input_ids"][i:i + 1].cpu(), b["image_feat"][i:i + 1].cpu(), b["image_loc"][i:i + 1].cpu(),
                          token_type_ids=b["token_type_ids"][i:i + 1].cpu(), position_ids=b["token_position_ids"][i:i + 1].cpu(),
                          attention_mask=b["attention_mask"][i:i + 1].cpu(), co_attention_mask=b["co_attention_mask"][i:i + 1].cpu(),
                          image_attention_mask=b["image_attention_mask"][i:i + 1].cpu())
            lab = b["masked_lm_labels"][i:i + 1].cpu()
            nll = torch.nn.functional.cross_entropy(o["pred_t"].view(-1, o["pred_t"].shape[-1]), lab.view(-1), ignore_index=-1,
                                                    reduction="none")
        want.append(float(-nll.sum()))
    want = torch.tensor(want)
    e = float((got[pick].cpu() - want).abs().max())
    print(f"  12 sequences vs oracle: max |err| {e:.3e} (values {float(want.min()):.2f} .. {float(want.max()):.2f})")
    assert e <= 1e-2 * float(want.abs().max())
    # a context that is not shared is reported, not silently scored against the wrong rows
    ids_bad = b["input_ids"].clone()
    ids_bad[137, 5] = ids_bad[137, 5] + 1
    bad, _ = model.sequence_log_likelihood(ids_bad, *args[1:], shared_context=grp, **kw)
    assert torch.isnan(bad[137]) and int(torch.isnan(bad).sum()) == 1
    keep = torch.ones(n, dtype=torch.bool, device=bad.device)
    keep[137] = False
    assert float((bad[keep] - got[keep]).abs().max()) <= 1e-4
    # CPU tensors handed over as val_lm.py does (val_lm.py:86-121): staged inside the call, the same scores on both paths (up to
    # the order of the per-sequence sums, as between two runs on device tensors)
    cargs = tuple(a.cpu() for a in args)
    ckw = {k: v.cpu() for k, v in kw.items()}
    got_h, nsp_h = model.sequence_log_likelihood(*cargs, shared_context=grp.cpu(), **ckw)
    base_h, _ = model.sequence_log_likelihood(*cargs, **ckw)
    torch.cuda.synchronize()
    assert float((got_h - got).abs().max()) <= 1e-4 and float((base_h - base).abs().max()) <= 1e-4
    assert float((nsp_h - nsp1).abs().max()) <= 1e-5
    # ... and so is a dense mask that is not the generative one in its context block, a deviating co-attention mask, or a member
    # that carries another image / image location than its group (these were assumed, not read)
    def only_nan_at(scores, i):
        return bool(torch.isnan(scores[i])) and int(torch.isnan(scores).sum()) == 1
    am_bad = kw["attention_mask"].clone()
    am_bad[41, 3, 2] = 0                                              # a context row that does not attend a context column
    assert only_nan_at(model.sequence_log_likelihood(*args, shared_context=grp, **{**kw, "attention_mask": am_bad})[0], 41)
    am_bad = kw["attention_mask"].clone()
    am_bad[42, 2, 0] = 1                                              # a context row that attends column 0
    assert only_nan_at(model.sequence_log_likelihood(*args, shared_context=grp, **{**kw, "attention_mask": am_bad})[0], 42)
    co_bad = kw["co_attention_mask"].clone()
    co_bad[43, 5, 1] = 0
    assert only_nan_at(model.sequence_log_likelihood(*args, shared_context=grp, **{**kw, "co_attention_mask": co_bad})[0], 43)
    loc_bad = args[2].clone()
    loc_bad[44, 7, 2] += 0.25
    assert only_nan_at(model.sequence_log_likelihood(args[0], args[1], loc_bad, *args[3:], shared_context=grp, **kw)[0], 44)
    with pytest.raises(ValueError):                                   # groups of different context lengths are refused up front
        model.sequence_log_likelihood(*args, shared_context=torch.zeros(n, dtype=torch.int64), **kw)


def test_full_config_pooler_gradients_tight_on_units_that_cannot_switch(golden_dir):
    """the bf16 gates on the two poolers (20 % L2 / 35 % worst element) cannot catch a pooler gradient that
    is wrong by a constant, because ~2.4 % of the ReLU units of the 6 pooled rows switch on bf16 noise and each of those gains or
    loses its whole gradient.  But WHICH units can switch is known: those whose pre-activation z = W h + b comes near zero for
    one of the sequences.  For all other units the reference's gradient (tests/golden/full_b6_grads.npz: every 4th bias element =
    256 units per pooler, 16 weight rows) must be reproduced to bf16 accuracy: here within 4 % of the largest safe element, with
    a least-squares scale between the two within 1.5 % of 1 -- a gradient wrong by a constant, a sign or a missing contributor fails."""
    from oracle import vilbert_ref as R
    from oracle.cases import grad_sample_index
    g = np.load(os.path.join(golden_dir, "full_b6.npz"))
    gg = np.load(os.path.join(golden_dir, "full_b6_grads.npz"))
    model, ocfg, sd = build_full(seed=5)
    args, kw = full_b6_call(g)
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, _ = model(*args, **kw, _want_lm_scores=False)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    with torch.no_grad():                                      # the oracle's pre-activations of the pooled rows (fp32, CPU)
        xt, xv, _, _ = R.trunk(sd, ocfg, *args, **{k: kw[k] for k in ("token_type_ids", "position_ids", "attention_mask",
                                                                       "image_attention_mask", "co_attention_mask")})
    params = dict(model.named_parameters())
    print()
    for side, x in (("t", xt), ("v", xv)):
        name = f"bert.{side}_pooler.dense"
        z = torch.nn.functional.linear(x[:, 0], sd[name + ".weight"], sd[name + ".bias"])          # [6, 1024]
        margin = 0.05 * float(z.pow(2).mean().sqrt())                                            # 10 x the bf16 noise of z (test docstring above)
        safe_unit = (z.abs() > margin).all(0).numpy()
        # bias gradient: every 4th unit
        want_b, got_b = gg["grad::" + name + ".bias"], params[name + ".bias"].grad[::4].double().cpu().numpy()
        sb = safe_unit[::4] & (np.abs(want_b) > 0)
        # weight gradient: 16 sampled rows x every 4th column
        rows = grad_sample_index(tuple(params[name + ".weight"].shape))[0]
        want_w = gg["grad::" + name + ".weight"]
        got_w = params[name + ".weight"].grad[T_(rows).cuda()][:, ::4].double().cpu().numpy()
        sw = safe_unit[rows]
        # (the 6 sequences share one image and most of their dialog: about half of the units are off for all of them, with a
        #  zero gradient in the reference -- those are compared too, through the weight rows, but carry no scale)
        assert sb.sum() >= 40 and sw.sum() >= 6, (side, int(sb.sum()), int(sw.sum()))
        for what, got, want in (("bias", got_b[sb], want_b[sb]), ("weight rows", got_w[sw].ravel(), want_w[sw].ravel())):
            err = float(np.abs(got - want).max() / np.abs(want).max())
            slope = float((got * want).sum() / (want * want).sum())
            print(f"  {name} {what}: {got.size} elements of units that cannot switch: max |err| / max |g| {err:.3e}, least-squares scale {slope:.4f}")
            assert err <= 4e-2 and abs(slope - 1.0) <= 1.5e-2, (name, what, err, slope)    # measured 1.1-2.3e-2 / 0.1-0.4e-2
