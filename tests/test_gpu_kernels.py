"""GPU parity of the attention / row / loss kernels (through the C ABI) against fp32 PyTorch
references of the same op on the same bf16-rounded inputs, and against the oracle's loss functions.
Tolerances: bf16 outputs 2^-7 relative to the tensor scale; fp32 loss scalars 1e-4."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"


def bf(x):
    return x.to(torch.bfloat16)


def relerr(a, b):
    return ((a.float() - b.float()).abs().max() / b.float().abs().max().clamp_min(1e-6)).item()


# ----------------------------------------------------------------------------------------------
# attention
# ----------------------------------------------------------------------------------------------
def _attn_case(B, H, Tq, Tk, D, dense_mask, seed, p_drop=0.0):
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(seed)
    HD = H * D
    qkv_q = bf(torch.randn((B * Tq, 3 * HD), generator=g, device=DEV))
    qkv_k = qkv_q if Tq == Tk else bf(torch.randn((B * Tk, 3 * HD), generator=g, device=DEV))
    q, k, v = qkv_q[:, :HD], qkv_k[:, HD:2 * HD], qkv_k[:, 2 * HD:]
    if dense_mask:
        m = (torch.rand((B, Tq, Tk), generator=g, device=DEV) < 0.6)
        m[:, :, 0] = True
        m[0, Tq // 2:] = False                  # fully masked rows (pad rows of the generative mask)
        mq, mb = (Tk + 31) // 32, Tq * ((Tk + 31) // 32)
    else:
        m = (torch.rand((B, 1, Tk), generator=g, device=DEV) < 0.8)
        m[:, :, 0] = True
        mq, mb = 0, (Tk + 31) // 32
    packed = lib.mask_pack(m.to(torch.int64) if seed % 2 else m)
    scale = 1.0 / math.sqrt(D)
    drop = DR.drop_arg(p_drop, DR.make_key(7, 1, seed))
    out = torch.zeros((B * Tq, HD), device=DEV, dtype=torch.bfloat16)
    lse = torch.zeros((B, H, Tq), device=DEV)
    lib.attn_fwd(q, k, v, out, lse, packed, B, H, Tq, Tk, D, scale, mq, mb, drop)

    # fp32 reference with autograd
    qf = q.float().reshape(B, Tq, H, D).permute(0, 2, 1, 3).detach().requires_grad_(True)
    kf = k.float().reshape(B, Tk, H, D).permute(0, 2, 1, 3).detach().requires_grad_(True)
    vf = v.float().reshape(B, Tk, H, D).permute(0, 2, 1, 3).detach().requires_grad_(True)
    s = qf @ kf.transpose(-1, -2) * scale + ((1.0 - m.float()) * -10000.0)[:, None]
    pr = torch.softmax(s, -1)
    if p_drop > 0:
        keep = DR.keep_mask_nd(drop[0], drop[1], (B, H, Tq, Tk))
        pr = pr * torch.from_numpy(keep).to(DEV) * drop[2]
    ref = (pr @ vf).permute(0, 2, 1, 3).reshape(B * Tq, HD)
    torch.cuda.synchronize()
    assert relerr(out, ref) < 2 ** -6, relerr(out, ref)
    ref_lse = torch.logsumexp(s, -1)
    assert (lse - ref_lse).abs().max().item() < 2e-2

    # backward
    dout = bf(torch.randn((B * Tq, HD), generator=g, device=DEV))
    ref.backward(dout.float())
    dq_buf = torch.zeros_like(qkv_q)
    dk_buf = dq_buf if Tq == Tk else torch.zeros_like(qkv_k)
    delta = torch.zeros((B, H, Tq), device=DEV)
    lib.attn_bwd(q, k, v, out, dout, lse, delta, dq_buf[:, :HD], dk_buf[:, HD:2 * HD], dk_buf[:, 2 * HD:], packed,
                 B, H, Tq, Tk, D, scale, mq, mb, drop)
    torch.cuda.synchronize()
    rq = qf.grad.permute(0, 2, 1, 3).reshape(B * Tq, HD)
    rk = kf.grad.permute(0, 2, 1, 3).reshape(B * Tk, HD)
    rv = vf.grad.permute(0, 2, 1, 3).reshape(B * Tk, HD)
    errs = {"out": relerr(out, ref), "lse": (lse - ref_lse).abs().max().item(), "dq": relerr(dq_buf[:, :HD], rq),
            "dk": relerr(dk_buf[:, HD:2 * HD], rk), "dv": relerr(dk_buf[:, 2 * HD:], rv)}
    print(f"attention B={B} H={H} Tq={Tq} Tk={Tk} D={D} dense={dense_mask} p={p_drop}: " +
          "  ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["dq"] < 2 ** -5 and errs["dk"] < 2 ** -5 and errs["dv"] < 2 ** -5, errs
    return errs


@pytest.mark.parametrize("B,H,Tq,Tk,D,dense", [
    (2, 2, 256, 256, 64, True),      # text self-attention, dense per-sequence mask
    (3, 2, 64, 64, 64, True),        # small-config text (tests/golden/small_config.json)
    (2, 2, 37, 37, 128, False),      # visual self-attention, key-padding mask
    (2, 2, 256, 37, 128, False),     # co-attention: text queries, image keys
    (2, 2, 37, 256, 128, True),      # co-attention: image queries, text keys (co-mask)
    (2, 2, 64, 37, 128, False), (2, 2, 37, 64, 128, True), (1, 1, 100, 200, 64, True),
])
def test_attention_fwd_bwd(B, H, Tq, Tk, D, dense):
    _attn_case(B, H, Tq, Tk, D, dense, seed=Tq + Tk + D)


@pytest.mark.parametrize("B,H,Tq,Tk,D,dense", [(2, 2, 256, 256, 64, True), (2, 2, 37, 256, 128, True),
                                               (2, 2, 256, 37, 128, False)])
def test_attention_dropout_replays_in_backward(B, H, Tq, Tk, D, dense):
    _attn_case(B, H, Tq, Tk, D, dense, seed=3 + Tq, p_drop=0.1)


# ----------------------------------------------------------------------------------------------
# row kernels
# ----------------------------------------------------------------------------------------------
@pytest.mark.parametrize("shape,dtype", [((3, 256, 256), torch.int64), ((3, 256, 256), torch.bool),
                                          ((5, 37), torch.float32), ((2, 37, 256), torch.int32), ((4, 100), torch.bool)])
def test_mask_pack(shape, dtype):
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(1)
    m = (torch.rand(shape, generator=g, device=DEV) < 0.5)
    words = lib.mask_pack(m.to(dtype)).cpu().numpy().view(np.uint32)
    t = shape[-1]
    bits = np.zeros(shape[:-1] + (((t + 31) // 32) * 32,), dtype=np.uint8)
    bits[..., :t] = m.cpu().numpy()
    want = np.packbits(bits.reshape(shape[:-1] + (-1, 32)), axis=-1, bitorder="little").view(np.uint32)[..., 0]
    assert np.array_equal(words, want)


@pytest.mark.parametrize("M,H", [(1000, 768), (333, 1024), (64, 128), (77, 256)])
def test_layernorm_fwd_bwd(M, H):
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(M)
    x = torch.randn((M, H), generator=g, device=DEV) * 2 + 0.5
    gamma = torch.randn(H, generator=g, device=DEV) * 0.2 + 1
    beta = torch.randn(H, generator=g, device=DEV) * 0.1
    y = torch.empty((M, H), device=DEV, dtype=torch.bfloat16)
    y32 = torch.empty((M, H), device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    lib.layernorm_fwd(x, gamma, beta, y32, y, mean, rstd, M, H)
    xr = x.clone().requires_grad_(True)
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-12)
    torch.cuda.synchronize()
    assert relerr(y, ref) < 2 ** -7 and relerr(y32, ref) < 1e-5
    dy = bf(torch.randn((M, H), generator=g, device=DEV))
    drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
    ref.backward(dy.float())
    dx, dxd = torch.empty_like(y), torch.empty_like(y)
    dg, db, dbias = torch.ones(H, device=DEV), torch.ones(H, device=DEV), torch.ones(H, device=DEV)
    part = torch.empty(lib.colpartials_bytes(H) // 4, device=DEV)
    lib.layernorm_bwd(dy, x, mean, rstd, gamma, dx, dxd, dg, db, dbias, part, M, H, drop=drop)
    torch.cuda.synchronize()
    assert relerr(dx, xr.grad) < 2 ** -6
    keep = torch.from_numpy(DR.keep_mask2d(drop[0], drop[1], M, H)).to(DEV)
    want_dxd = xr.grad * keep * drop[2]
    assert relerr(dxd, want_dxd) < 2 ** -6
    assert relerr(dg - 1, gr.grad) < 5e-3 and relerr(db - 1, br.grad) < 5e-3      # accumulate (+=) semantics
    assert relerr(dbias - 1, want_dxd.sum(0)) < 5e-3


def test_embeddings_fwd_bwd():
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(4)
    M, H, V = 700, 768, 2000
    tabs = [torch.randn((n, H), generator=g, device=DEV) * 0.05 for n in (V, 512, 2, 10)]
    ids = torch.randint(0, V, (M,), generator=g, device=DEV, dtype=torch.int32)
    ids[::7] = 103
    pos = torch.randint(0, 512, (M,), generator=g, device=DEV, dtype=torch.int32)
    typ = torch.randint(0, 12, (M,), generator=g, device=DEV, dtype=torch.int32)
    typ[::3] = 0
    typ[1::3] = 1
    gamma = torch.randn(H, generator=g, device=DEV) * 0.2 + 1
    beta = torch.randn(H, generator=g, device=DEV) * 0.1
    y = torch.empty((M, H), device=DEV, dtype=torch.bfloat16)
    y32 = torch.empty((M, H), device=DEV)
    lib.embed_fwd(ids, pos, typ, *tabs, gamma, beta, y32, y, M, H)
    leaves = [t.clone().requires_grad_(True) for t in tabs]
    gr, br = gamma.clone().requires_grad_(True), beta.clone().requires_grad_(True)
    il, pl, tl = ids.long(), pos.long(), typ.long()
    tvec = torch.where((tl >= 2)[:, None], leaves[3][(tl - 2).clamp_min(0)], leaves[2][tl.clamp_max(1)])
    ref = torch.nn.functional.layer_norm(leaves[0][il] + leaves[1][pl] + tvec, (H,), gr, br, 1e-12)
    torch.cuda.synchronize()
    assert relerr(y, ref) < 2 ** -7 and relerr(y32, ref) < 1e-5
    dy = bf(torch.randn((M, H), generator=g, device=DEV))
    ref.backward(dy.float())
    grads = [torch.zeros((n, H), device=DEV) for n in (V, 512, 2, 10)]
    dg, db = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    part = torch.empty(lib.colpartials_bytes(H) // 4, device=DEV)
    lib.embed_bwd(ids, pos, typ, *tabs, gamma, beta, dy, grads[0], grads[1], grads[2], grads[3], dg, db, part, M, H)
    torch.cuda.synchronize()
    for got, leaf in zip(grads, leaves):
        assert relerr(got, leaf.grad) < 2e-3
    assert relerr(dg, gr.grad) < 2e-3 and relerr(db, br.grad) < 2e-3


def test_colsum_casts_pack():
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(9)
    dy = bf(torch.randn((1234, 1608), generator=g, device=DEV))
    db = torch.ones(1601, device=DEV)
    lib.colsum(dy, db, 1234, 1601)
    torch.cuda.synchronize()
    assert relerr(db - 1, dy[:, :1601].float().sum(0)) < 1e-3
    w = torch.randn((300, 200), generator=g, device=DEV)
    flat = torch.empty(300 * 200, device=DEV, dtype=torch.bfloat16)
    lib.cast_f32_bf16(w, flat)
    wt = torch.full((200, 320), 7.0, device=DEV, dtype=torch.bfloat16)
    lib.transpose_cast(w, wt, 300, 200, 320)
    torch.cuda.synchronize()
    assert torch.equal(flat.view(300, 200), bf(w))
    assert torch.equal(wt[:, :300], bf(w).t()) and (wt[:, 300:] == 0).all()
    feat = torch.randn((74, 2048), generator=g, device=DEV)
    loc = torch.rand((74, 5), generator=g, device=DEV)
    packed = torch.full((74, 2112), 3.0, device=DEV, dtype=torch.bfloat16)
    lib.pack_image(feat, loc, packed, 74, 2048, 2112)
    torch.cuda.synchronize()
    assert torch.equal(packed[:, :2048], bf(feat)) and torch.equal(packed[:, 2048:2053], bf(loc))
    assert (packed[:, 2053:] == 0).all()


# ----------------------------------------------------------------------------------------------
# losses (reference: the oracle's functions, which are pinned to the reference's goldens)
# ----------------------------------------------------------------------------------------------
def test_lm_ul_loss_fwd_bwd_against_oracle():
    from oracle import vilbert_ref as R
    from oracle.cases import loss_inputs
    from unimm_amd import lib
    li = loss_inputs()
    V = li["pred_t"].shape[-1]
    ld = (V + 7) // 8 * 8
    z = torch.from_numpy(li["pred_t"]).reshape(-1, V)
    labels, weights = torch.from_numpy(li["labels"]).reshape(-1), torch.from_numpy(li["weights"]).reshape(-1)
    sel = torch.nonzero(weights != 0)[:, 0]
    n = sel.numel()
    logits = torch.zeros((n, ld), device=DEV)
    logits[:, :V] = z[sel].to(DEV)
    lab, wgt = labels[sel].int().to(DEV), weights[sel].int().to(DEV)
    rowloss, rownll, lse = (torch.empty(n, device=DEV) for _ in range(3))
    lib.lm_loss_fwd(logits, lab, wgt, rowloss, rownll, lse, n, V)
    loss = torch.empty(1, device=DEV)
    lib.reduce_sum(rowloss, n, loss, 1.0 / n)
    zt = torch.from_numpy(li["pred_t"]).requires_grad_(True)
    want = R.mlm_ul_loss(zt, torch.from_numpy(li["labels"]), torch.from_numpy(li["weights"]))
    want.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - want.item()) < 1e-4 * max(1.0, abs(want.item()))
    gdev = torch.ones(1, device=DEV)
    dl = torch.full((n, ld), 9.0, device=DEV, dtype=torch.bfloat16)
    lib.lm_loss_bwd(logits, lab, wgt, lse, gdev, 1.0 / n, dl, n, V)
    torch.cuda.synchronize()
    ref = zt.grad.reshape(-1, V)[sel]
    assert (dl[:, :V].float().cpu() - ref).abs().max() < 2 ** -7 * ref.abs().max() + 1e-6
    assert (dl[:, V:] == 0).all()
    # per-row nll = generative scoring term (val_lm.py:131-136)
    ref_nll = torch.nn.functional.cross_entropy(z[sel], labels[sel], reduction="none")
    assert (rownll.cpu() - ref_nll).abs().max() < 1e-3


def test_kl_and_nsp_losses_against_oracle():
    from oracle import vilbert_ref as R
    from oracle.cases import loss_inputs
    from unimm_amd import lib
    li = loss_inputs()
    pv = torch.from_numpy(li["pred_v"])
    B, Rg, C = pv.shape
    rows = B * Rg
    pred = torch.zeros((rows, 1604), device=DEV)
    pred[:, :C] = pv.reshape(rows, C).to(DEV)
    tgt = torch.from_numpy(li["image_target"]).reshape(rows, C).contiguous().to(DEV)
    lab = torch.from_numpy(li["image_label"]).reshape(rows).int().to(DEV)
    rowloss, lse, loss = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV), torch.empty(1, device=DEV)
    lib.kl_loss_fwd(pred, tgt, lab, rowloss, lse, rows, C)
    nsel = int((li["image_label"] == 1).sum())
    lib.reduce_sum(rowloss, rows, loss, 1.0 / nsel)
    pvt = pv.clone().requires_grad_(True)
    want = R.image_kl_loss(pvt, torch.from_numpy(li["image_target"]), torch.from_numpy(li["image_label"]))
    want.backward()
    torch.cuda.synchronize()
    assert abs(loss.item() - want.item()) < 1e-4 * max(1.0, abs(want.item()))
    dp = torch.full((rows, 1664), 5.0, device=DEV, dtype=torch.bfloat16)
    lib.kl_loss_bwd(pred, tgt, lab, lse, torch.ones(1, device=DEV), 1.0 / nsel, dp, rows, C)
    torch.cuda.synchronize()
    ref = pvt.grad.reshape(rows, C)
    assert (dp[:, :C].float().cpu() - ref).abs().max() < 2 ** -7 * ref.abs().max() + 1e-7
    assert (dp[:, C:] == 0).all()
    # NSP
    g = torch.Generator().manual_seed(0)
    nb = 37
    z = torch.randn((nb, 2), generator=g)
    y = (torch.rand(nb, generator=g) < 0.8).long()
    zt = z.clone().requires_grad_(True)
    want = R.nsp_loss(zt, y, torch.tensor([[5.0, 1.0]]))
    want.backward()
    zd = torch.zeros((nb, 4), device=DEV)
    zd[:, :2] = z.to(DEV)
    loss = torch.empty(1, device=DEV)
    lib.nsp_loss_fwd(zd, y.int().to(DEV), 1.0, 0.2, loss, nb)
    dz = torch.full((nb, 4), 3.0, device=DEV)
    lib.nsp_loss_bwd(zd, y.int().to(DEV), 1.0, 0.2, torch.ones(1, device=DEV), dz, nb)
    torch.cuda.synchronize()
    assert abs(loss.item() - want.item()) < 1e-5
    assert (dz[:, :2].cpu() - zt.grad).abs().max() < 1e-6
    assert (dz[:, 2:] == 0).all()
    extra = torch.randn((nb, 2), generator=g).to(DEV)          # a gradient arriving through the returned scores
    dz2 = torch.empty((nb, 2), device=DEV)
    lib.nsp_loss_bwd(zd, y.int().to(DEV), 1.0, 0.2, torch.ones(1, device=DEV), dz2, nb, extra=extra)
    torch.cuda.synchronize()
    assert (dz2 - (dz[:, :2] + extra)).abs().max() < 1e-6


@pytest.mark.parametrize("M,N,K", [(240, 1024, 768), (240, 1024, 1024), (6, 2, 1024), (37, 50, 20), (100, 2, 1024)])
def test_linear_f32_three_forms_against_fp64(M, N, K):
    """unimm_linear_f32 (v_mfma_f32_16x16x4_f32, exact fp32): y = relu(x W^T + b), dx = dy W, dW += dy^T x and
    db += colsum(dy) against fp64 torch, ragged tiles, the 16-byte path and the strided path."""
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(M + N + K)
    x = torch.randn((M, K), generator=g, device=DEV)
    w = torch.randn((N, K), generator=g, device=DEV) * 0.05
    b = torch.randn(N, generator=g, device=DEV)
    ldo = (N + 3) // 4 * 4
    y = torch.full((M, ldo), 9.0, device=DEV)
    lib.linear_f32(x, w, y, M, N, K, (K, 1), (1, K), bias=b, relu=True)
    ref = torch.relu(x.double() @ w.double().t() + b.double())
    assert (y[:, :N].double() - ref).abs().max().item() <= 2e-6 * max(1.0, ref.abs().max().item())
    assert (y[:, N:] == 9.0).all()
    dy = torch.randn((M, N), generator=g, device=DEV)
    dx = torch.empty((M, K), device=DEV)
    lib.linear_f32(dy, w, dx, M, K, N, (N, 1), (K, 1))
    refdx = dy.double() @ w.double()
    assert (dx.double() - refdx).abs().max().item() <= 2e-6 * max(1.0, refdx.abs().max().item())
    dw = torch.ones((N, K), device=DEV)                          # accumulates on top of what is there
    db = torch.ones(N, device=DEV)
    lib.linear_f32(dy, x, dw, N, K, M, (1, N), (K, 1), accumulate=True, rowsum=db)
    torch.cuda.synchronize()
    refdw = 1.0 + dy.double().t() @ x.double()
    assert (dw.double() - refdw).abs().max().item() <= 4e-6 * max(1.0, refdw.abs().max().item())
    refdb = 1.0 + dy.double().sum(0)
    assert (db.double() - refdb).abs().max().item() <= 4e-6 * max(1.0, refdb.abs().max().item())


def test_mul_dropout_and_rows_add_fp32():
    """fused pooled vector (models/vilbert_dialog.py:1064-1065) in fp32, its backward with the pooler ReLU gates folded
    in, and the fp32 -> bf16 row accumulation of the pooler input gradient."""
    from unimm_amd import lib
    from unimm_amd import dropout as DR
    g = torch.Generator(device=DEV).manual_seed(5)
    n = 6 * 1024
    a = torch.relu(torch.randn(n, generator=g, device=DEV))
    b = torch.relu(torch.randn(n, generator=g, device=DEV))
    key = DR.make_key(1, 2, 3)
    drop = DR.drop_arg(0.1, key)
    keep = torch.from_numpy(DR.keep_mask2d(key, drop[1], 1, n)).to(DEV).reshape(-1).float() * drop[2]
    out = torch.empty(n, device=DEV)
    lib.mul_dropout(a, b, out, n, drop)
    assert torch.equal(out, a * b * keep)
    do = torch.randn(n, generator=g, device=DEV)
    da, db_ = torch.empty(n, device=DEV), torch.empty(n, device=DEV)
    lib.mul_dropout_bwd(a, b, do, da, db_, n, drop)
    assert torch.equal(da, torch.where(a > 0, do * keep * b, torch.zeros_like(a)))
    assert torch.equal(db_, torch.where(b > 0, do * keep * a, torch.zeros_like(a)))
    H = 768
    dst = bf(torch.randn((50, H), generator=g, device=DEV))
    want = dst.float().clone()
    idx = torch.tensor([3, 17, 49, 0], dtype=torch.int32, device=DEV)
    src = torch.randn((4, H), generator=g, device=DEV)
    want[idx.long()] += src
    lib.rows_add_f32(dst, idx, src, 4, H)
    torch.cuda.synchronize()
    assert torch.equal(dst, bf(want))


def test_attention_variable_length_matches_padded():
    """Packed (unpadded) q/k/v with per-sequence offsets and lengths == the padded layout on the valid rows."""
    from unimm_amd import lib
    B, H, T, D = 3, 2, 256, 64
    HD = H * D
    g = torch.Generator(device=DEV).manual_seed(11)
    lens = [200, 37, 129]
    qkv = bf(torch.randn((B * T, 3 * HD), generator=g, device=DEV))
    m = torch.zeros((B, T, T), dtype=torch.bool, device=DEV)
    for b, l in enumerate(lens):
        m[b, :l, :l] = torch.rand((l, l), generator=g, device=DEV) < 0.7
        m[b, :l, 0] = True
    packed = lib.mask_pack(m)
    nw = T // 32
    scale = D ** -0.5
    out_p = torch.zeros((B * T, HD), device=DEV, dtype=torch.bfloat16)
    lse_p = torch.zeros((B, H, T), device=DEV)
    lib.attn_fwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out_p, lse_p, packed, B, H, T, T, D, scale, nw, T * nw)
    rows = torch.cat([torch.arange(b * T, b * T + l, device=DEV) for b, l in enumerate(lens)])
    off = torch.tensor([0, lens[0], lens[0] + lens[1]], dtype=torch.int32, device=DEV)
    ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
    qkv_v = qkv[rows].contiguous()
    out_v = torch.zeros((rows.numel(), HD), device=DEV, dtype=torch.bfloat16)
    lse_v = torch.zeros((B, H, T), device=DEV)
    lib.attn_fwd(qkv_v[:, :HD], qkv_v[:, HD:2 * HD], qkv_v[:, 2 * HD:], out_v, lse_v, packed, B, H, T, T, D, scale, nw, T * nw,
                 qvar=(off, ln), kvar=(off, ln))
    torch.cuda.synchronize()
    assert torch.equal(out_v, out_p[rows])
    dout = bf(torch.randn((B * T, HD), generator=g, device=DEV))
    keep = torch.zeros(B * T, dtype=torch.bool, device=DEV)
    keep[rows] = True
    dout[~keep] = 0                                   # padding rows carry no gradient
    dqkv_p, dqkv_v = torch.zeros_like(qkv), torch.zeros_like(qkv_v)
    delta = torch.zeros((B, H, T), device=DEV)
    lib.attn_bwd(qkv[:, :HD], qkv[:, HD:2 * HD], qkv[:, 2 * HD:], out_p, dout, lse_p, delta, dqkv_p[:, :HD],
                 dqkv_p[:, HD:2 * HD], dqkv_p[:, 2 * HD:], packed, B, H, T, T, D, scale, nw, T * nw)
    dout_v = dout[rows].contiguous()
    lib.attn_bwd(qkv_v[:, :HD], qkv_v[:, HD:2 * HD], qkv_v[:, 2 * HD:], out_v, dout_v, lse_v, delta, dqkv_v[:, :HD],
                 dqkv_v[:, HD:2 * HD], dqkv_v[:, 2 * HD:], packed, B, H, T, T, D, scale, nw, T * nw,
                 qvar=(off, ln), kvar=(off, ln))
    torch.cuda.synchronize()
    assert relerr(dqkv_v, dqkv_p[rows]) < 1e-2


@pytest.mark.parametrize("D,Tq,Tk", [(64, 256, 256), (128, 37, 256), (128, 256, 37)])
def test_attention_item_order_changes_nothing_but_the_schedule(D, Tq, Tk):
    """unimm_attn_args.order (the sequences longest first, from unimm_plan_build): forward outputs, log-sum-exps and all three
    gradients are the same bits whatever order the workgroups take the (sequence, head) items in -- every kernel form (the
    one-kernel text backward, the dQ + dK/dV pair of the 37-region directions), with dropout."""
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    B, H = 7, 3
    HD = H * D
    g = torch.Generator(device=DEV).manual_seed(D + Tq)
    lens = [200, 37, 129, 256, 64, 1, 255]
    var_q, var_k = Tq == 256, Tk == 256
    ql = lens if var_q else [Tq] * B
    kl = lens if var_k else [Tk] * B
    nq, nk = sum(ql), sum(kl)
    q = bf(torch.randn((nq, HD), generator=g, device=DEV))
    kv = bf(torch.randn((nk, 2 * HD), generator=g, device=DEV))
    m = torch.zeros((B, Tq, Tk), dtype=torch.bool, device=DEV)
    for b in range(B):
        m[b, :ql[b], :kl[b]] = torch.rand((ql[b], kl[b]), generator=g, device=DEV) < 0.7
        m[b, :ql[b], 0] = True
    packed = lib.mask_pack(m)
    nw = packed.shape[-1]
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    cum = lambda v: i32([sum(v[:i]) for i in range(B)])
    drop = DR.drop_arg(0.1, DR.make_key(2, 7, 1))
    dout = bf(torch.randn((nq, HD), generator=g, device=DEV))
    res = []
    for order in (None, i32(sorted(range(B), key=lambda b: (-lens[b], b))), i32([3, 0, 6, 5, 1, 2, 4])):
        qv = (cum(ql), i32(ql), None, order) if var_q else None
        kvv = (cum(kl), i32(kl), None, order) if var_k else None
        out = torch.zeros((nq, HD), device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros((B, H, Tq), device=DEV)
        lib.attn_fwd(q, kv[:, :HD], kv[:, HD:], out, lse, packed, B, H, Tq, Tk, D, D ** -0.5, Tk // 32 * 0 + nw, Tq * nw, drop,
                     qvar=qv, kvar=kvv)
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        delta = torch.zeros((B, H, Tq), device=DEV)
        lib.attn_bwd(q, kv[:, :HD], kv[:, HD:], out, dout, lse, delta, dq, dkv[:, :HD], dkv[:, HD:], packed, B, H, Tq, Tk, D,
                     D ** -0.5, nw, Tq * nw, drop, qvar=qv, kvar=kvv)
        torch.cuda.synchronize()
        res.append((out, lse, dq, dkv))
    for r in res[1:]:
        for a, b in zip(res[0], r):
            assert torch.equal(a.view(torch.int16) if a.dtype == torch.bfloat16 else a, b.view(torch.int16) if b.dtype == torch.bfloat16 else b)
    assert float(res[0][2].float().abs().max()) > 0


@pytest.mark.parametrize("Tq,Tk", [(37, 256), (256, 37)])
def test_coattention_variable_length_matches_padded(Tq, Tk):
    """The two co-attention directions (D = 128; 37 regions on one side, packed text rows of per-sequence length on the other --
    the one-kernel backward forms attn_bwd_fewq128 / attn_bwd_fewk128) against the same problem in the padded layout: forward
    bit-equal on the valid rows, gradients to bf16 accuracy, with dropout (the counters index padded positions)."""
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    B, H, D = 5, 2, 128
    HD = H * D
    g = torch.Generator(device=DEV).manual_seed(Tq)
    lens = [200, 33, 129, 256, 64]
    text_q = Tq == 256
    ql, kl = (lens if text_q else [Tq] * B), ([Tk] * B if text_q else lens)
    qp = bf(torch.randn((B * Tq, HD), generator=g, device=DEV))
    kvp = bf(torch.randn((B * Tk, 2 * HD), generator=g, device=DEV))
    m = torch.zeros((B, Tq, Tk), dtype=torch.bool, device=DEV)
    for b in range(B):
        m[b, :ql[b], :kl[b]] = torch.rand((ql[b], kl[b]), generator=g, device=DEV) < 0.7
        m[b, :ql[b], 0] = True
    packed = lib.mask_pack(m)
    nw = packed.shape[-1]
    drop = DR.drop_arg(0.1, DR.make_key(4, 1, 9))
    sc = D ** -0.5
    qrows = torch.cat([torch.arange(b * Tq, b * Tq + l, device=DEV) for b, l in enumerate(ql)])
    krows = torch.cat([torch.arange(b * Tk, b * Tk + l, device=DEV) for b, l in enumerate(kl)])
    dout_p = bf(torch.randn((B * Tq, HD), generator=g, device=DEV))
    keepq = torch.zeros(B * Tq, dtype=torch.bool, device=DEV)
    keepq[qrows] = True
    dout_p[~keepq] = 0
    i32 = lambda v: torch.tensor(v, dtype=torch.int32, device=DEV)
    cum = lambda v: i32([sum(v[:i]) for i in range(B)])

    def run(q, kv, dout, qv, kvv):
        out = torch.zeros((q.shape[0], HD), device=DEV, dtype=torch.bfloat16)
        lse = torch.zeros((B, H, Tq), device=DEV)
        lib.attn_fwd(q, kv[:, :HD], kv[:, HD:], out, lse, packed, B, H, Tq, Tk, D, sc, nw, Tq * nw, drop, qvar=qv, kvar=kvv)
        dq, dkv = torch.zeros_like(q), torch.zeros_like(kv)
        delta = torch.zeros((B, H, Tq), device=DEV)
        lib.attn_bwd(q, kv[:, :HD], kv[:, HD:], out, dout, lse, delta, dq, dkv[:, :HD], dkv[:, HD:], packed, B, H, Tq, Tk, D, sc,
                     nw, Tq * nw, drop, qvar=qv, kvar=kvv)
        torch.cuda.synchronize()
        return out, dq, dkv

    out_p, dq_p, dkv_p = run(qp, kvp, dout_p, None, None)
    qv = (cum(ql), i32(ql)) if text_q else None
    kvv = None if text_q else (cum(kl), i32(kl))
    out_v, dq_v, dkv_v = run(qp[qrows].contiguous(), kvp[krows].contiguous(), dout_p[qrows].contiguous(), qv, kvv)
    assert torch.equal(out_v, out_p[qrows])
    assert relerr(dq_v, dq_p[qrows]) < 1e-2 and relerr(dkv_v, dkv_p[krows]) < 1e-2


def test_mask_synth_matches_the_oracle_encoders_bit_for_bit():
    """unimm_mask_synth against oracle/masks.py (itself pinned to the reference's encode_input_gen / _dis by golden
    G5): packed text and co-attention words for generative and discriminative sequences, incl. one-token answers,
    a copy block cut by the sequence end, and other T."""
    from oracle import masks as OM
    from unimm_amd import lib
    from unimm_amd.inputs import DialogMaskSpec
    for T in (256, 64, 96):
        cases = []
        rng = np.random.default_rng(T)
        for _ in range(40):
            n_utt = int(rng.integers(1, 6))
            a = int(rng.integers(1, 9))
            lens = [int(rng.integers(1, max(2, (T - 3 * a) // (2 * n_utt)))) for _ in range(n_utt)] + [a]
            cases.append((int(rng.integers(0, 2)), lens))
        cases += [(1, [3, 1]), (1, [1, 1]), (0, [2, 2]), (1, [T - 14, 9]), (0, [T - 3])]     # [T-14, 9]: copy block cut at T
        mode, Ls, ns, txt, co = [], [], [], [], []
        for m, lens in cases:
            utts = [list(range(1000, 1000 + l)) for l in lens]
            enc = (OM.encode_gen if m else OM.encode_dis)(utts, max_seq_len=T)
            L = 1 + sum(l + 1 for l in lens)
            if L > T:
                continue
            mode.append(m); Ls.append(L); ns.append(lens[-1] + 1)
            txt.append(np.asarray(enc["txt_attention_mask"][0]) != 0); co.append(np.asarray(enc["co_attention_mask"][0]) != 0)
        spec = DialogMaskSpec(mode, Ls, ns)
        tw, cw = lib.mask_synth(*spec.to_device(DEV), T)
        want_t = lib.mask_pack(torch.from_numpy(np.stack(txt)).to(DEV))
        want_c = lib.mask_pack(torch.from_numpy(np.stack(co)).to(DEV))
        torch.cuda.synchronize()
        assert torch.equal(tw, want_t) and torch.equal(cw, want_c)
        dt, dc = spec.dense(T)                                   # the host-side dense form agrees too
        assert torch.equal(dt, torch.from_numpy(np.stack(txt))) and torch.equal(dc, torch.from_numpy(np.stack(co)))


def test_plan_kernels_match_the_dense_formulation():
    """unimm_plan_lengths / unimm_plan_build against the definition on the dense masks (what the engine used to compute
    with eager ops): valid prefix lengths, row maps, and the decoded rows in torch.nonzero order."""
    import struct
    from unimm_amd import lib as L
    g = torch.Generator().manual_seed(5)
    for B, T, R, two_d, use_w in [(7, 256, 37, False, True), (5, 64, 37, False, False), (6, 100, 9, True, True), (3, 256, 256, False, True)]:
        lens_true = torch.randint(1, T + 1, (B,), generator=g)
        if two_d:
            am = (torch.arange(T)[None, :] < lens_true[:, None]).to(torch.uint8)
        else:
            am = torch.zeros(B, T, T, dtype=torch.uint8)
            for b in range(B):
                n = int(lens_true[b])
                am[b, :n, :n] = (torch.rand(n, n, generator=g) < 0.5).to(torch.uint8)
                am[b, torch.randint(0, n, (1,), generator=g), n - 1] = 1       # someone attends the last valid token
        cm = torch.zeros(B, R, T, dtype=torch.uint8)
        for b in range(B):
            cm[b, :, :max(1, int(lens_true[b]) - 3)] = (torch.rand(R, max(1, int(lens_true[b]) - 3), generator=g) < 0.3).to(torch.uint8)
        labels = torch.full((B, T), -1, dtype=torch.int64)
        weights = torch.zeros(B, T, dtype=torch.int64)
        for b in range(B):
            pos = torch.randperm(T, generator=g)[:5]
            labels[b, pos] = torch.randint(0, 1000, (5,), generator=g)
            weights[b, pos[:4]] = torch.tensor([1, -1, 1, -1])                  # one labelled row with weight 0 (context mask of a negative)
        labels[0, T - 1] = 7
        weights[0, T - 1] = 1                                                   # a label past every mask extends the prefix
        nspw = torch.tensor([[5.0, 1.0]], device="cuda")
        amd, cmd = am.cuda(), cm.cuda()
        tw, cw = L.mask_pack(amd), L.mask_pack(cmd)
        nw = tw.shape[-1]
        tmask = (tw, 0, nw) if two_d else (tw, nw, T * nw)
        lab32, w32 = labels.cuda().int(), (weights.cuda().int() if use_w else None)
        il = torch.randint(-1, 2, (B, R), generator=g)
        header = L.plan_lengths(tmask, (cw, nw, R * nw), R, lab32, w32, nspw.reshape(-1), B, T, image_label=il.cuda().int())
        hh = header.tolist()
        assert hh[2 * B + 2:3 * B + 2] == il.eq(1).sum(1).tolist()
        # definition on the dense tensors
        valid = am.ne(0) if two_d else (am.ne(0).any(1) | am.ne(0).any(2))
        valid = valid | cm.ne(0).any(1) | labels.ne(-1)
        if use_w:
            valid = valid | weights.ne(0)
        want_len = (valid.int() * torch.arange(1, T + 1)).amax(1).clamp_min(1)
        if not two_d:
            # a valid token whose own mask row is empty attends all T keys uniformly in the reference (:1418), padding
            # included: such a sequence runs at its full length
            forced = (valid & ~am.ne(0).any(2)).any(1)
            want_len = torch.where(forced, torch.full_like(want_len, T), want_len)
        assert hh[:B] == want_len.tolist()
        selm = weights.ne(0) if use_w else labels.ne(-1)
        assert hh[B:2 * B] == selm.sum(1).tolist()
        assert struct.unpack("<2f", struct.pack("<2i", hh[2 * B], hh[2 * B + 1])) == (5.0, 1.0)
        Mv, n = sum(hh[:B]), sum(hh[B:2 * B])
        built = L.plan_build(header, lab32, w32, B, T, Mv, n)
        off = torch.cat([torch.zeros(1, dtype=torch.int64), want_len.cumsum(0)[:-1]])
        assert built["off"].cpu().tolist() == off.tolist() and built["lens"].cpu().tolist() == want_len.tolist()
        rows = torch.cat([torch.arange(b * T, b * T + int(l)) for b, l in enumerate(want_len)])
        assert torch.equal(built["rows"].cpu(), rows)
        inv = torch.full((B * T,), -1, dtype=torch.int64)
        inv[rows] = torch.arange(Mv)
        assert torch.equal(built["inv"].cpu(), inv)
        pos = torch.nonzero(selm.reshape(-1))[:, 0]
        assert built["lm_pos"].cpu().tolist() == pos.tolist()
        assert built["lm_idx"].cpu().tolist() == inv[pos].tolist()
        assert built["lm_label"].cpu().tolist() == labels.reshape(-1)[pos].tolist()
        assert built["lm_weight"].cpu().tolist() == (weights.reshape(-1)[pos].tolist() if use_w else [1] * n)
        # the item order of the attention launches: by length, longest first, ties in batch order
        assert built["order"].cpu().tolist() == sorted(range(B), key=lambda b: (-int(want_len[b]), b))
    # nothing labelled, no masks beyond labels=None: counts are zero and the lm group may be omitted
    header = L.plan_lengths((tw, nw, T * nw), None, 0, None, None, None, B, T)
    assert sum(header.tolist()[B:2 * B]) == 0
    b2 = L.plan_build(header, None, None, B, T, sum(header.tolist()[:B]), 0)
    assert b2["lm_pos"] is None and b2["rows"].numel() == sum(header.tolist()[:B])


@pytest.mark.parametrize("R,C,lds", [(37, 1000, 1008), (627, 30522, 30528), (64, 64, 64), (1, 9, 16)])
def test_transpose_bf16_exact(R, C, lds):
    """unimm_transpose_bf16: dst[c, r] = src[r, c] bit for bit, surplus destination columns zero, source padding ignored."""
    from unimm_amd import lib as L
    g = torch.Generator(device="cuda").manual_seed(R * 131 + C)
    src = torch.randn((R, lds), generator=g, device="cuda").to(torch.bfloat16)
    ldd = (R + 63) // 64 * 64
    dst = torch.full((C, ldd), 7.0, dtype=torch.bfloat16, device="cuda")
    L.transpose_bf16(src[:, :C], dst, R, C)
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :R], src[:, :C].t())
    assert (dst[:, R:] == 0).all()


def test_combine_losses_equals_the_written_out_arithmetic():
    """harness.combine_losses (one autograd node for c_lm * lm.mean() + c_nsp * nsp.mean() + c_img * img.mean(), train.py:164-168)."""
    from unimm_amd import harness
    vals = [torch.tensor([v], device=DEV, requires_grad=True) for v in (3.25, 0.7, 0.55)]
    ref = [v.detach().clone().requires_grad_(True) for v in vals]
    c = (1.0, 0.5, 2.0)
    got = harness.combine_losses(vals[0], vals[1], vals[2], *c)
    want = c[0] * ref[0].mean() + c[1] * ref[1].mean() + c[2] * ref[2].mean()
    (got * 3.0).backward()
    (want * 3.0).backward()
    assert got.shape == want.shape and abs(float(got) - float(want)) < 1e-6
    for a, b in zip(vals, ref):
        assert a.grad.shape == b.grad.shape and torch.allclose(a.grad, b.grad)
    cpu = [torch.tensor([1.0, 2.0], requires_grad=True) for _ in range(3)]          # several replicas / CPU: the plain arithmetic
    assert abs(float(harness.combine_losses(*cpu, 1.0, 1.0, 1.0)) - 4.5) < 1e-6


def test_device_prefetcher_reads_fresh_host_buffers_and_keeps_pinned_memory_bounded():
    """unimm_amd.inputs.DevicePrefetcher with a loader that REUSES one host buffer and mutates it every step (and hands over
    fresh tensors besides): every batch must arrive with the values it had when it was drawn, and the staging memory must
    stay at `ring` pinned sets however many batches pass (round-4 advisor finding: the per-tensor pinned cache grew without
    bound and re-sent the first sighting of a reused buffer).  cache_pinned=True keeps the one-pin-per-tensor form for cycled,
    immutable batches."""
    from unimm_amd.inputs import DevicePrefetcher
    shared = torch.zeros(1024, dtype=torch.int64)

    def loader(n):
        for i in range(n):
            shared.fill_(i)                                    # the loader's reused buffer
            yield {"reused": shared, "fresh": torch.full((257,), float(i)), "meta": i}

    pf = DevicePrefetcher(loader(12), "cuda", ring=3)
    seen = 0
    for i, b in enumerate(pf):
        assert b["meta"] == i and b["reused"].is_cuda and b["fresh"].is_cuda
        assert int(b["reused"][0]) == i and int(b["reused"][-1]) == i, (i, b["reused"][:2])
        assert float(b["fresh"][3]) == float(i)
        seen += 1
    assert seen == 12
    assert len(pf._pinned) == 0 and len(pf._ring) == 3
    assert sum(len(s["bufs"]) for s in pf._ring) <= 3 * 2      # two pageable tensors per batch, three staging sets
    hb = [{"x": torch.full((64,), float(k))} for k in range(2)]
    import itertools
    pf2 = DevicePrefetcher(itertools.islice(itertools.cycle(hb), 8), "cuda", cache_pinned=True)
    vals = [float(b["x"][0]) for b in pf2]
    assert vals == [0.0, 1.0] * 4 and len(pf2._pinned) == 2


@pytest.mark.parametrize("p_drop", [0.0, 0.1])
def test_attention_few_queries_forward_packed_keys_match_padded(p_drop):
    """attn_fwd_fewq128_kernel (round 6: the regions-attend-text forward with the KEYS dealt to the waves, D = 128, 37 queries x up
    to 256 keys): packed (variable-length) keys with per-sequence offsets and lengths == the padded layout bit for bit, with and
    without dropout, with a per-query co-attention mask and with a key-padding mask; both agree with the fp32 reference."""
    from unimm_amd import lib
    from unimm_amd import dropout as DR
    B, H, Tq, T, D = 4, 2, 37, 256, 128
    HD = H * D
    g = torch.Generator(device=DEV).manual_seed(23)
    lens = [256, 37, 130, 1]
    q = bf(torch.randn((B * Tq, HD), generator=g, device=DEV))
    kv = bf(torch.randn((B * T, 2 * HD), generator=g, device=DEV))
    scale = D ** -0.5
    drop = DR.drop_arg(p_drop, DR.make_key(5, 2, 99)) if p_drop > 0 else lib.NO_DROP
    for dense in (True, False):
        m = torch.zeros((B, Tq if dense else 1, T), dtype=torch.bool, device=DEV)
        for b, l in enumerate(lens):
            m[b, :, :l] = torch.rand((m.shape[1], l), generator=g, device=DEV) < 0.7
            m[b, :, 0] = True
        packed = lib.mask_pack(m)
        nw = T // 32
        mq, mb = (nw, Tq * nw) if dense else (0, nw)
        out_p = torch.zeros((B * Tq, HD), device=DEV, dtype=torch.bfloat16)
        lse_p = torch.zeros((B, H, Tq), device=DEV)
        lib.attn_fwd(q, kv[:, :HD], kv[:, HD:], out_p, lse_p, packed, B, H, Tq, T, D, scale, mq, mb, drop)
        rows = torch.cat([torch.arange(b * T, b * T + l, device=DEV) for b, l in enumerate(lens)])
        off = torch.tensor(np.concatenate([[0], np.cumsum(lens)[:-1]]), dtype=torch.int32, device=DEV)
        ln = torch.tensor(lens, dtype=torch.int32, device=DEV)
        kv_v = kv[rows].contiguous()
        out_v = torch.zeros_like(out_p)
        lse_v = torch.zeros_like(lse_p)
        lib.attn_fwd(q, kv_v[:, :HD], kv_v[:, HD:], out_v, lse_v, packed, B, H, Tq, T, D, scale, mq, mb, drop, kvar=(off, ln))
        torch.cuda.synchronize()
        assert torch.equal(out_v, out_p) and torch.equal(lse_v, lse_p), (dense, p_drop)
        qf = q.float().reshape(B, Tq, H, D).permute(0, 2, 1, 3)
        kf = kv[:, :HD].float().reshape(B, T, H, D).permute(0, 2, 1, 3)
        vf = kv[:, HD:].float().reshape(B, T, H, D).permute(0, 2, 1, 3)
        s = qf @ kf.transpose(-1, -2) * scale + ((1.0 - m.float()) * -10000.0)[:, None]
        pr = torch.softmax(s, -1)
        if p_drop > 0:
            pr = pr * torch.from_numpy(DR.keep_mask_nd(drop[0], drop[1], (B, H, Tq, T))).to(DEV) * drop[2]
        ref = (pr @ vf).permute(0, 2, 1, 3).reshape(B * Tq, HD)
        assert relerr(out_p, ref) < 2 ** -6, relerr(out_p, ref)
        assert (lse_p - torch.logsumexp(s, -1)).abs().max().item() < 2e-2
