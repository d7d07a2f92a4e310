"""GPU parity of the fp32-accuracy mode's kernels (csrc/x3ops.hip, through the C ABI): the split-operand GEMM (the bf16
MFMA kernels over [hi | lo | hi] x [hi | hi | lo] planes) against fp64, the fp32 attention cores against an fp64 reference
with autograd (masks, -10000 additive, dropout replay, variable length), LayerNorm / loss backward on fp32 gradients.
Gate: 1e-3 is north_star's fp32 tolerance for model OUTPUTS; single kernels must sit far inside it (<= 5e-5 of scale)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def rel(a, b):
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))


def rup(x, m):
    return (x + m - 1) // m * m


def split(a, wtype=False, op=None, b=None, want32=False):
    from unimm_amd import lib
    rows, cols = a.shape
    cp = rup(cols, 64)
    out3 = torch.full((rows, 3 * cp), float("nan"), dtype=torch.bfloat16, device=DEV)
    out32 = torch.empty((rows, cols), device=DEV) if want32 else None
    lib.x3_split(a, out3=out3, out32=out32, op=lib.X3_COPY if op is None else op, b=b, wtype=wtype)
    return (out3, out32) if want32 else out3


@pytest.mark.parametrize("M,N,K", [(300, 768, 768), (257, 3072, 768), (130, 768, 3072), (64, 30522, 768), (100, 1024, 2112)])
def test_split_gemm_nt_matches_fp64(M, N, K):
    """X3 . W3^T on unimm_gemm_nt = fp32-grade x . w^T (+ bias)."""
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(M + N)
    x = torch.randn((M, K), generator=g, device=DEV)
    w = torch.randn((N, K), generator=g, device=DEV) * 0.05
    bias = torch.randn(N, generator=g, device=DEV)
    x3, w3 = split(x), split(w, wtype=True)
    assert torch.isfinite(x3.float()).all()                       # padding columns were written (zeros)
    ldo = rup(N, 4)
    out = torch.empty((M, ldo), device=DEV)
    lib.gemm_nt(x3, w3, out, bias=bias, M=M, N=N, K=x3.shape[1])
    want = x.double() @ w.double().t() + bias.double()
    torch.cuda.synchronize()
    e3 = rel(out[:, :N], want)
    e1 = rel((x.bfloat16().float() @ w.bfloat16().float().t() + bias), want)
    print(f"\nsplit GEMM {M}x{N}x{K}: fp32x3 {e3:.2e}   (bf16 operands {e1:.2e}, torch fp32 {rel(x @ w.t() + bias, want):.2e})")
    assert e3 < 2e-5, e3


def test_split_weight_gradient_as_three_grouped_problems():
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(3)
    M, N, K = 1500, 768, 1024
    dy = torch.randn((M, N), generator=g, device=DEV)
    x = torch.randn((M, K), generator=g, device=DEV)
    dy3, x3 = split(dy), split(x)
    dw = torch.zeros((N, K), device=DEV)
    db = torch.zeros(N, device=DEV)
    lib.gemm_tn_grouped([(dy3[:, :N], x3[:, :K], dw, M, N, K, db), (dy3[:, N:2 * N], x3[:, :K], dw, M, N, K, db),
                         (dy3[:, :N], x3[:, K:2 * K], dw, M, N, K, None)])
    torch.cuda.synchronize()
    want = dy.double().t() @ x.double()
    assert rel(dw, want) < 2e-5, rel(dw, want)
    assert rel(db, dy.double().sum(0)) < 2e-5


def test_transposed_weight_split_and_elementwise_ops():
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(4)
    R, C = 770, 300
    w = torch.randn((R, C), generator=g, device=DEV)
    Rp = rup(R, 64)
    wt3 = torch.zeros((C, 3 * Rp), dtype=torch.bfloat16, device=DEV)
    lib.x3_split_wt(w, wt3, R, C, Rp)
    hi = w.bfloat16()
    lo = (w - hi.float()).bfloat16()
    torch.cuda.synchronize()
    assert torch.equal(wt3[:, :R], hi.t()) and torch.equal(wt3[:, Rp:Rp + R], hi.t()) and torch.equal(wt3[:, 2 * Rp:2 * Rp + R], lo.t())
    assert float(wt3[:, R:Rp].abs().max()) == 0
    a = torch.randn((37, 1601), generator=g, device=DEV)            # ragged width: element-wise tail path
    b = torch.randn((37, 1601), generator=g, device=DEV)
    for op, ref in ((lib.X3_COPY, a), (lib.X3_ADD, a + b), (lib.X3_GELU, torch.nn.functional.gelu(a)),
                    (lib.X3_MUL_DGELU, None)):
        o3, o32 = split(a, op=op, b=b, want32=True)
        if ref is None:
            bb = b.double().requires_grad_(True)
            torch.nn.functional.gelu(bb).sum().backward()
            ref = (a.double() * bb.grad).float()
        cp = o3.shape[1] // 3
        torch.cuda.synchronize()
        assert rel(o32, ref) < 2e-6, (op, rel(o32, ref))
        rec = o3[:, :1601].float() + o3[:, cp:cp + 1601].float()
        assert rel(rec, ref) < 2e-5 and torch.equal(o3[:, :cp], o3[:, 2 * cp:])
        assert float(o3[:, 1601:cp].abs().max()) == 0 and float(o3[:, cp + 1601:2 * cp].abs().max()) == 0


def _attn_case(B, H, Tq, Tk, D, dense_mask, seed, p_drop=0.0, varlen=False):
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(seed)
    HD = H * D
    qkv_q = torch.randn((B * Tq, 3 * HD), generator=g, device=DEV)
    qkv_k = qkv_q if Tq == Tk else torch.randn((B * Tk, 3 * HD), generator=g, device=DEV)
    q, k, v = qkv_q[:, :HD], qkv_k[:, HD:2 * HD], qkv_k[:, 2 * HD:]
    if dense_mask:
        m = (torch.rand((B, Tq, Tk), generator=g, device=DEV) < 0.6)
        m[:, :, 0] = True
        if not varlen:
            m[0, Tq // 2:] = False               # fully masked rows: softmax over the raw scores, as the reference (:1418)
        mq, mb = (Tk + 31) // 32, Tq * ((Tk + 31) // 32)
    else:
        m = (torch.rand((B, 1, Tk), generator=g, device=DEV) < 0.8)
        m[:, :, 0] = True
        mq, mb = 0, (Tk + 31) // 32
    qvar = kvar = None
    ql = torch.full((B,), Tq, dtype=torch.int32)
    kl = torch.full((B,), Tk, dtype=torch.int32)
    if varlen:                                   # valid prefixes; keys past the prefix are masked for every query
        ql = torch.randint(max(1, Tq // 3), Tq + 1, (B,), generator=torch.Generator().manual_seed(seed)).to(torch.int32)
        kl = ql.clone() if Tq == Tk else torch.randint(max(1, Tk // 3), Tk + 1, (B,), generator=torch.Generator().manual_seed(seed + 1)).to(torch.int32)
        for b in range(B):
            m[b, :, int(kl[b]):] = False
    packed = lib.mask_pack(m)
    scale = 1.0 / math.sqrt(D)
    drop = DR.drop_arg(p_drop, DR.make_key(7, 1, seed))

    def pack_rows(x, T, lens):                   # padded [B*T, .] -> packed valid rows
        return torch.cat([x[b * T:b * T + int(lens[b])] for b in range(B)], 0).contiguous()

    if varlen:
        qoff = torch.cat([torch.zeros(1, dtype=torch.int32), ql.cumsum(0)[:-1].to(torch.int32)])
        koff = torch.cat([torch.zeros(1, dtype=torch.int32), kl.cumsum(0)[:-1].to(torch.int32)])
        qvar, kvar = (qoff.to(DEV), ql.to(DEV)), (koff.to(DEV), kl.to(DEV))
        qp, kp, vp = pack_rows(q, Tq, ql), pack_rows(k, Tk, kl), pack_rows(v, Tk, kl)
    else:
        qp, kp, vp = q, k, v
    out = torch.zeros((qp.shape[0], HD), device=DEV)
    lse = torch.zeros((B, H, Tq), device=DEV)
    lib.x3_attn_fwd(qp, kp, vp, out, lse, packed, B, H, Tq, Tk, D, scale, mq, mb, drop, qvar=qvar, kvar=kvar)

    qf = q.double().reshape(B, Tq, H, D).permute(0, 2, 1, 3).detach().requires_grad_(True)
    kf = k.double().reshape(B, Tk, H, D).permute(0, 2, 1, 3).detach().requires_grad_(True)
    vf = v.double().reshape(B, Tk, H, D).permute(0, 2, 1, 3).detach().requires_grad_(True)
    s = qf @ kf.transpose(-1, -2) * scale + ((1.0 - m.double()) * -10000.0)[:, None]
    pr = torch.softmax(s, -1)
    if p_drop > 0:
        keep = DR.keep_mask_nd(drop[0], drop[1], (B, H, Tq, Tk))
        pr = pr * torch.from_numpy(keep).to(DEV) * drop[2]
    ref = (pr @ vf).permute(0, 2, 1, 3).reshape(B * Tq, HD)
    ref_lse = torch.logsumexp(s, -1)
    torch.cuda.synchronize()
    valid_q = torch.cat([torch.arange(b * Tq, b * Tq + int(ql[b])) for b in range(B)]).to(DEV)
    valid_k = torch.cat([torch.arange(b * Tk, b * Tk + int(kl[b])) for b in range(B)]).to(DEV)
    got = out if not varlen else torch.zeros((B * Tq, HD), device=DEV).index_copy_(0, valid_q, out)
    errs = {"out": rel(got[valid_q], ref[valid_q])}
    lmask = torch.zeros((B, Tq), dtype=torch.bool, device=DEV)
    for b in range(B):
        lmask[b, :int(ql[b])] = True
    # Fully masked query rows (pad rows of the generative mask) are softmax(raw scores - 10000): at magnitude 1e4 an fp32
    # score has an ulp of 1e-3, in the reference's fp32 arithmetic as much as here ("garbage-but-finite", never
    # reach a loss).  They are held to 2e-3 against fp64; rows that attend something to fp32 accuracy.
    live = (m.any(-1) if dense_mask else torch.ones((B, Tq), dtype=torch.bool, device=DEV)) & lmask
    live_rows = live.reshape(-1).nonzero().flatten()
    errs["out"] = rel(got[live_rows], ref[live_rows])
    errs["lse"] = float((lse.double() - ref_lse).abs().permute(0, 2, 1)[live].max())
    dead = (~live & lmask).reshape(-1).nonzero().flatten()
    if dead.numel():
        e_dead = rel(got[dead], ref[dead])
        assert e_dead < 2e-3, e_dead
        errs["out(masked rows)"] = e_dead

    dout = torch.randn((B * Tq, HD), generator=g, device=DEV)
    dsel = torch.zeros_like(dout)
    dsel[live_rows] = dout[live_rows]
    dout = dsel                                   # padding / fully masked rows carry no gradient (they reach no loss)
    ref.backward(dout.double())
    doutp = pack_rows(dout, Tq, ql) if varlen else dout
    dq = torch.full_like(qp, float("nan"))
    dk, dv = torch.full_like(kp, float("nan")), torch.full_like(vp, float("nan"))
    delta = torch.zeros((B, H, Tq), device=DEV)
    lib.x3_attn_bwd(qp, kp, vp, out, doutp, lse, delta, dq, dk, dv, packed, B, H, Tq, Tk, D, scale, mq, mb, drop, qvar=qvar, kvar=kvar)
    torch.cuda.synchronize()
    rq = qf.grad.permute(0, 2, 1, 3).reshape(B * Tq, HD)[valid_q]
    rk = kf.grad.permute(0, 2, 1, 3).reshape(B * Tk, HD)[valid_k]
    rv = vf.grad.permute(0, 2, 1, 3).reshape(B * Tk, HD)[valid_k]
    errs.update(dq=rel(dq, rq), dk=rel(dk, rk), dv=rel(dv, rv))
    print(f"\nfp32 attention B={B} H={H} Tq={Tq} Tk={Tk} D={D} dense={dense_mask} p={p_drop} varlen={varlen}: " +
          "  ".join(f"{k} {v:.2e}" for k, v in errs.items()))
    assert errs["out"] < 2e-5 and errs["lse"] < 1e-4, errs
    assert errs["dq"] < 5e-5 and errs["dk"] < 5e-5 and errs["dv"] < 5e-5, errs


@pytest.mark.parametrize("B,H,Tq,Tk,D,dense", [
    (2, 2, 256, 256, 64, True), (3, 2, 64, 64, 64, True), (2, 2, 37, 37, 128, False), (2, 2, 256, 37, 128, False),
    (2, 2, 37, 256, 128, True), (1, 1, 100, 200, 64, True), (2, 3, 130, 70, 128, True),
    (2, 1, 65, 17, 64, True), (1, 2, 16, 129, 128, True), (3, 1, 1, 3, 64, False),      # tile / chunk edges of the matrix kernels
])
def test_fp32_attention_fwd_bwd(B, H, Tq, Tk, D, dense):
    _attn_case(B, H, Tq, Tk, D, dense, seed=Tq + Tk + D)


@pytest.mark.parametrize("B,H,Tq,Tk,D,dense", [(2, 2, 256, 256, 64, True), (2, 2, 37, 256, 128, True), (2, 2, 256, 37, 128, False)])
def test_fp32_attention_dropout_and_variable_length(B, H, Tq, Tk, D, dense):
    _attn_case(B, H, Tq, Tk, D, dense, seed=3 + Tq, p_drop=0.1)
    if Tq == Tk:
        _attn_case(B, H, Tq, Tk, D, dense, seed=5 + Tq, p_drop=0.1, varlen=True)


@pytest.mark.parametrize("B,H,Tq,Tk,D,dense", [(2, 2, 256, 256, 64, True), (2, 3, 130, 70, 128, True)])
def test_fp32_attention_vector_kernels_still_agree(B, H, Tq, Tk, D, dense):
    """unimm_x3_attn_set_impl(0): the vector-ALU kernels of the first version (kept for A/B runs) pass the same gates."""
    from unimm_amd import lib
    lib.x3_attn_set_impl(0)
    try:
        _attn_case(B, H, Tq, Tk, D, dense, seed=11 + Tq, p_drop=0.1)
        if Tq == Tk:
            _attn_case(B, H, Tq, Tk, D, dense, seed=13 + Tq, p_drop=0.1, varlen=True)
    finally:
        lib.x3_attn_set_impl(1)


def _unsplit(x3, cols):
    """x-type split operand [rows, 3 cp] -> (hi + lo as fp32, the three planes)"""
    cp = x3.shape[1] // 3
    hi, lo, hi2 = x3[:, :cp].float(), x3[:, cp:2 * cp].float(), x3[:, 2 * cp:].float()
    return (hi + lo)[:, :cols], hi, lo, hi2


@pytest.mark.parametrize("B,H,Tq,Tk,D", [(2, 2, 200, 200, 64), (2, 2, 37, 150, 128)])
def test_fp32_attention_split_outputs_equal_split_of_fp32_outputs(B, H, Tq, Tk, D):
    """The planes the attention kernels write themselves (context forward; dQ / dK / dV backward, into column slices of a wider
    gradient buffer) are bit for bit unimm_x3_split of their fp32 results."""
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(Tq)
    HD = H * D
    q = torch.randn((B * Tq, HD), generator=g, device=DEV)
    k = torch.randn((B * Tk, HD), generator=g, device=DEV)
    v = torch.randn((B * Tk, HD), generator=g, device=DEV)
    m = torch.rand((B, Tq, Tk), generator=g, device=DEV) < 0.7
    m[:, :, 0] = True
    packed = lib.mask_pack(m)
    mq, mb = (Tk + 31) // 32, Tq * ((Tk + 31) // 32)
    drop = DR.drop_arg(0.1, DR.make_key(3, 1, Tq))
    scale = 1.0 / math.sqrt(D)
    out, out_b = torch.empty((B * Tq, HD), device=DEV), torch.empty((B * Tq, HD), device=DEV)
    lse = torch.empty((B, H, Tq), device=DEV)
    out3 = torch.zeros((B * Tq, 3 * HD), dtype=torch.bfloat16, device=DEV)
    lib.x3_attn_fwd(q, k, v, out, lse, packed, B, H, Tq, Tk, D, scale, mq, mb, drop, out3=out3)
    lib.x3_attn_fwd(q, k, v, out_b, lse, packed, B, H, Tq, Tk, D, scale, mq, mb, drop)
    want3 = torch.empty_like(out3)
    lib.x3_split(out_b, out3=want3)
    torch.cuda.synchronize()
    assert torch.equal(out, out_b) and torch.equal(out3, want3)
    dout = torch.randn((B * Tq, HD), generator=g, device=DEV)
    delta = torch.empty_like(lse)
    dq, dk, dv = torch.empty_like(q), torch.empty_like(k), torch.empty_like(v)
    lib.x3_attn_bwd(q, k, v, out, dout, lse, delta, dq, dk, dv, packed, B, H, Tq, Tk, D, scale, mq, mb, drop)
    # gradient buffers as the engine lays them out: [rows, 3 N] with N = 3 HD, dq / dk / dv in column slices of plane 0
    N = 3 * HD
    gq = torch.zeros((B * Tq, 3 * N), dtype=torch.bfloat16, device=DEV)
    gk = torch.zeros((B * Tk, 3 * N), dtype=torch.bfloat16, device=DEV)
    lib.x3_attn_bwd(q, k, v, out, dout, lse, delta, gq[:, :HD], gk[:, HD:2 * HD], gk[:, 2 * HD:3 * HD], packed, B, H, Tq, Tk, D, scale,
                    mq, mb, drop, planes=(N, gq.stride(0)))
    torch.cuda.synchronize()
    for t32, buf, c0 in ((dq, gq, 0), (dk, gk, HD), (dv, gk, 2 * HD)):
        want = torch.empty((t32.shape[0], 3 * HD), dtype=torch.bfloat16, device=DEV)
        lib.x3_split(t32, out3=want)
        torch.cuda.synchronize()
        for pl in range(3):
            assert torch.equal(buf[:, pl * N + c0:pl * N + c0 + HD], want[:, pl * HD:(pl + 1) * HD]), (c0, pl)
    assert float(gq[:, HD:N].abs().max()) == 0.0            # nothing written outside the slices


@pytest.mark.parametrize("M,H", [(1000, 768), (77, 1024)])
def test_fp32_layernorm_forward_split_output(M, H):
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(M)
    x = torch.randn((M, H), generator=g, device=DEV) * 3 - 0.7
    gamma = torch.randn(H, generator=g, device=DEV) * 0.2 + 1
    beta = torch.randn(H, generator=g, device=DEV) * 0.1
    for drop in (DR.drop_arg(0.1, DR.make_key(2, 5, M)), lib.NO_DROP):
        y_a, y_b = torch.empty((M, H), device=DEV), torch.empty((M, H), device=DEV)
        m_a, r_a, m_b, r_b = (torch.empty(M, device=DEV) for _ in range(4))
        y3 = torch.empty((M, 3 * H), dtype=torch.bfloat16, device=DEV)
        lib.x3_layernorm_fwd(x, gamma, beta, y_a, y3, m_a, r_a, M, H, drop=drop)
        lib.layernorm_fwd(x, gamma, beta, y_b, None, m_b, r_b, M, H, drop=drop)
        want3 = torch.empty_like(y3)
        lib.x3_split(y_b, out3=want3)
        torch.cuda.synchronize()
        assert torch.equal(y_a, y_b) and torch.equal(m_a, m_b) and torch.equal(r_a, r_b) and torch.equal(y3, want3)
    ref = torch.nn.functional.layer_norm(x.double(), (H,), gamma.double(), beta.double(), 1e-12)
    assert rel(_unsplit(y3, H)[0], ref) < 3e-5 and rel(y_a, ref) < 2e-6         # (last pass: no dropout)


@pytest.mark.parametrize("M,H", [(1000, 768), (333, 1024), (64, 128)])
def test_fp32_layernorm_backward(M, H):
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(M)
    x = torch.randn((M, H), generator=g, device=DEV) * 2 + 0.5
    gamma = torch.randn(H, generator=g, device=DEV) * 0.2 + 1
    beta = torch.randn(H, generator=g, device=DEV) * 0.1
    y32 = torch.empty((M, H), device=DEV)
    mean, rstd = torch.empty(M, device=DEV), torch.empty(M, device=DEV)
    lib.layernorm_fwd(x, gamma, beta, y32, None, mean, rstd, M, H)
    xr = x.double().requires_grad_(True)
    gr, br = gamma.double().requires_grad_(True), beta.double().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (H,), gr, br, 1e-12)
    dy = torch.randn((M, H), generator=g, device=DEV)
    ref.backward(dy.double())
    drop = DR.drop_arg(0.1, DR.make_key(1, 2, 3))
    dx32 = torch.empty((M, H), device=DEV)
    dxd3 = torch.empty((M, 3 * H), dtype=torch.bfloat16, device=DEV)
    part = torch.empty(lib.colpartials_bytes(H) // 4, device=DEV)
    blocks = lib.x3_layernorm_bwd_partials(dy, x, mean, rstd, gamma, dx32, dxd3, part, M, H, drop=drop)
    dg, dbt, dbias = torch.zeros(H, device=DEV), torch.zeros(H, device=DEV), torch.zeros(H, device=DEV)
    lib.colpartials_finish_grouped([(part, blocks, H, [dg, dbt, dbias])])
    torch.cuda.synchronize()
    assert rel(y32, ref) < 2e-6
    assert rel(dx32, xr.grad) < 2e-5, rel(dx32, xr.grad)
    assert rel(dg, gr.grad) < 2e-5 and rel(dbt, br.grad) < 2e-5
    keep = torch.from_numpy(DR.keep_mask2d(drop[0], drop[1], M, H)).to(DEV)
    want_d = xr.grad * keep * drop[2]
    rec = dxd3[:, :H].float() + dxd3[:, H:2 * H].float()
    assert rel(rec, want_d) < 5e-5 and torch.equal(dxd3[:, :H], dxd3[:, 2 * H:])
    assert rel(dbias, want_d.sum(0)) < 5e-5


def test_fp32_loss_backward_as_split_operands():
    from oracle import vilbert_ref as R
    from unimm_amd import lib
    g = torch.Generator(device=DEV).manual_seed(9)
    n, V = 70, 30522
    Vp = rup(V, 64)
    logits = torch.randn((n, Vp), generator=g, device=DEV) * 2
    labels = torch.randint(0, V, (n,), generator=g, device=DEV, dtype=torch.int32)
    weights = torch.tensor([1, -1, 2, 0, 1, -1, 1], device=DEV, dtype=torch.int32).repeat(10)
    labels[3] = -1
    logits[1, labels[1]] = 30.0                      # p_y -> 1: the clamp regime of the unlikelihood term (zero gradient)
    rowloss, rownll, lse = (torch.empty(n, device=DEV) for _ in range(3))
    lib.lm_loss_fwd(logits, labels, weights, rowloss, rownll, lse, n, V)
    gup = torch.tensor([0.7], device=DEV)
    nz = int((weights != 0).sum())
    out3 = torch.empty((n, 3 * Vp), dtype=torch.bfloat16, device=DEV)
    lib.x3_lm_loss_bwd(logits, labels, weights, lse, gup, 1.0 / nz, out3, n, V)
    z = logits[:, :V].double().cpu().requires_grad_(True)
    lab = labels.long().cpu().clone()
    w = weights.long().cpu()
    loss = R.mlm_ul_loss(z.unsqueeze(0), lab.unsqueeze(0), w.unsqueeze(0))
    (loss * 0.7).backward()
    torch.cuda.synchronize()
    rec = (out3[:, :V].float() + out3[:, Vp:Vp + V].float()).cpu().double()
    assert rel(rec, z.grad) < 5e-5, rel(rec, z.grad)
    assert float(out3[:, V:Vp].abs().max()) == 0 and torch.equal(out3[:, :Vp], out3[:, 2 * Vp:])
    # region KL
    rows, C = 74, 1601
    Cp = rup(C, 64)
    pred = torch.randn((rows, rup(C, 4)), generator=g, device=DEV)
    tgt = torch.softmax(torch.randn((rows, C), generator=g, device=DEV), -1)
    label = torch.randint(-1, 2, (rows,), generator=g, device=DEV, dtype=torch.int32)
    rl, lse2 = torch.empty(rows, device=DEV), torch.empty(rows, device=DEV)
    lib.kl_loss_fwd(pred, tgt, label, rl, lse2, rows, C)
    o3 = torch.empty((rows, 3 * Cp), dtype=torch.bfloat16, device=DEV)
    nlab = max(1, int((label == 1).sum()))
    lib.x3_kl_loss_bwd(pred, tgt, label, lse2, gup, 1.0 / nlab, o3, rows, C)
    pz = pred[:, :C].double().requires_grad_(True)
    kl = (torch.nn.functional.kl_div(torch.log_softmax(pz, -1), tgt.double(), reduction="none") * (label == 1).double()[:, None]).sum() / nlab
    (kl * 0.7).backward()
    torch.cuda.synchronize()
    rec = o3[:, :C].float() + o3[:, Cp:Cp + C].float()
    assert rel(rec, pz.grad) < 5e-5, rel(rec, pz.grad)
