"""GPU parity of the bf16 MFMA GEMM kernels (through the C ABI) against fp32 torch matmul on the
same bf16-rounded inputs.  Tolerance: fp32 accumulation of bf16 products -> 2e-3 relative to the
row scale for fp32 outputs, bf16 rounding (2^-8) for bf16 outputs."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def _rand(shape, gen, scale=1.0):
    return (torch.randn(shape, generator=gen, device="cuda") * scale).to(torch.bfloat16)


def _gelu(x):
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 384, 128), (300, 200, 192), (37 * 5, 1601, 256),
                                   (1000, 1000, 768), (61, 30522, 128), (4096, 2304, 768)])
def test_gemm_nt_bias_f32_and_bf16(M, N, K):
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(M * 7 + N)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda")
    ref = x.float() @ w.float().t() + bias
    ldo = (N + 7) // 8 * 8
    out32 = torch.full((M, ldo), float("nan"), device="cuda")
    lib.gemm_nt(x, w, out32, bias=bias, N=N)
    torch.cuda.synchronize()
    err = (out32[:, :N] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    if ldo > N:
        assert torch.isnan(out32[:, N:]).all()          # pad columns untouched
    out16 = torch.zeros((M, ldo), device="cuda", dtype=torch.bfloat16)
    lib.gemm_nt(x, w, out16, bias=bias, N=N)
    torch.cuda.synchronize()
    assert (out16[:, :N].float() - ref).abs().max().item() <= 2 ** -7 * max(1.0, ref.abs().max().item())


def test_gemm_nt_asymmetric_identity():
    """A = I against an asymmetric B catches a transposed C layout (cdna guide 3)."""
    from unimm_amd import lib
    n = 128
    x = torch.eye(n, device="cuda", dtype=torch.bfloat16)
    w = (torch.arange(n * n, device="cuda").reshape(n, n) % 251).to(torch.bfloat16)
    out = torch.empty((n, n), device="cuda")
    lib.gemm_nt(x, w, out)
    torch.cuda.synchronize()
    assert torch.equal(out, w.float().t())


def test_gemm_nt_strided_views_and_epilogues():
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(5)
    M, N, K = 384, 256, 128
    big = _rand((M, 3 * K), g)
    x = big[:, K:2 * K]                     # row stride 3K
    w = _rand((N, K), g, 0.1)
    bias = torch.randn(N, generator=g, device="cuda")
    aux = _rand((M, N), g)
    base = x.float() @ w.float().t()
    # GELU with both outputs
    h = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    u = torch.empty_like(h)
    lib.gemm_nt(x, w, h, bias=bias, epilogue=lib.EPI_BIAS_GELU, out2=u)
    torch.cuda.synchronize()
    assert (u.float() - (base + bias)).abs().max() <= 2 ** -7 * (base + bias).abs().max()
    assert (h.float() - _gelu(base + bias)).abs().max() <= 2 ** -7 * (base + bias).abs().max()
    # fp32 residual stream (no dropout)
    o = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    o32 = torch.empty((M, N), device="cuda")
    res32 = torch.randn((M, N), generator=g, device="cuda")
    lib.gemm_nt(x, w, o32, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=res32)
    ref = base + bias + res32
    torch.cuda.synchronize()
    assert (o32 - ref).abs().max() <= 2e-3 * ref.abs().max()
    with pytest.raises(lib.UnimmHipError):      # residual epilogue only writes fp32
        lib.gemm_nt(x, w, o, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=res32)
    # relu
    lib.gemm_nt(x, w, o, bias=bias, epilogue=lib.EPI_BIAS_RELU)
    torch.cuda.synchronize()
    assert (o.float() - torch.relu(base + bias)).abs().max() <= 2 ** -7 * ref.abs().max()
    # dgelu: out = acc * gelu'(aux)
    a32 = aux.float().requires_grad_(True)
    _gelu(a32).sum().backward()
    lib.gemm_nt(x, w, o, epilogue=lib.EPI_DGELU, aux=aux)
    torch.cuda.synchronize()
    ref = base * a32.grad
    assert (o.float() - ref).abs().max() <= 2 ** -7 * max(1.0, ref.abs().max().item())
    # mul, and the fused GELU + GELU' epilogue of the training FFN
    lib.gemm_nt(x, w, o, epilogue=lib.EPI_MUL, aux=aux)
    torch.cuda.synchronize()
    ref = base * aux.float()
    assert (o.float() - ref).abs().max() <= 2 ** -7 * max(1.0, ref.abs().max().item())
    dg = torch.empty_like(h)
    lib.gemm_nt(x, w, h, bias=bias, epilogue=lib.EPI_BIAS_GELU_DG, out2=dg)
    torch.cuda.synchronize()
    u32 = (base + bias).requires_grad_(True)
    _gelu(u32).sum().backward()
    assert (h.float() - _gelu(base + bias)).abs().max() <= 2 ** -7 * (base + bias).abs().max()
    assert (dg.float() - u32.grad).abs().max() <= 2 ** -7
    # add
    lib.gemm_nt(x, w, o, epilogue=lib.EPI_ADD, aux=aux)
    torch.cuda.synchronize()
    ref = base + aux.float()
    assert (o.float() - ref).abs().max() <= 2 ** -7 * ref.abs().max()


@pytest.mark.parametrize("M,N,K", [(64, 128, 128), (640, 256, 384), (1000, 200, 72), (37 * 30, 1024, 1024),
                                   (5000, 768, 768), (333, 1601, 256), (700, 1000, 128)])
def test_gemm_tn(M, N, K):
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    ldy, ldx = (N + 7) // 8 * 8, (K + 7) // 8 * 8
    dy = torch.zeros((M, ldy), device="cuda", dtype=torch.bfloat16)
    x = torch.zeros((M, ldx), device="cuda", dtype=torch.bfloat16)
    dy[:, :N] = _rand((M, N), g)
    x[:, :K] = _rand((M, K), g)
    init = torch.randn((N, K), generator=g, device="cuda")
    dw = init.clone()
    db = torch.ones(N, device="cuda")
    lib.gemm_tn(dy, x, dw, N=N, K=K, dbias=db)
    torch.cuda.synchronize()
    ref = init + dy[:, :N].float().t() @ x[:, :K].float()
    err = (dw - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    ref_b = 1 + dy[:, :N].float().sum(0)
    assert (db - ref_b).abs().max().item() <= 2e-3 * max(1.0, ref_b.abs().max().item())


@pytest.mark.parametrize("cfg", [1, 3, 6, 7, 8, 12, 14, 15])
@pytest.mark.parametrize("M,N,K", [(300, 200, 64), (1000, 1000, 768), (515, 2304, 128), (4096, 768, 3072), (257, 257 * 3, 320)])
def test_gemm_nt_every_tile_configuration(cfg, M, N, K):
    """All block-tile / ring configurations of unimm_gemm_nt give the same result (the automatic choice only
    covers some of them at a given shape): ragged M and N, K from 2 to 96 ring steps, fp32 residual epilogue."""
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(M + 3 * N + K)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda")
    ldo = (N + 7) // 8 * 8
    resid = torch.randn((M, ldo), generator=g, device="cuda")
    ref = x.float() @ w.float().t() + bias + resid[:, :N]
    out = torch.full((M, ldo), float("nan"), device="cuda")
    lib.gemm_nt(x, w, out, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=resid, N=N, tile=cfg)
    torch.cuda.synchronize()
    err = (out[:, :N] - ref).abs().max().item()
    assert err <= 2e-3 * max(1.0, ref.abs().max().item()), err
    if ldo > N:
        assert torch.isnan(out[:, N:]).all()


@pytest.mark.parametrize("cfg", [101, 103, 108, 112, 114, 201, 203, 208, 212, 215])
def test_gemm_nt_persistent_workgroups(cfg):
    """x1xx = persistent workgroups (one per CU slot walking several tiles), x2xx = one workgroup per tile: same
    result on a grid of several rounds, ragged M edge, GELU epilogue with its second output."""
    from unimm_amd import lib
    M, N, K = 8300, 3072, 768
    g = torch.Generator(device="cuda").manual_seed(cfg)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda")
    u = x.float() @ w.float().t() + bias
    out = torch.zeros((M, N), device="cuda", dtype=torch.bfloat16)
    out2 = torch.zeros_like(out)
    for _ in range(2):                      # twice: a persistent workgroup must leave no state behind
        lib.gemm_nt(x, w, out, bias=bias, epilogue=lib.EPI_BIAS_GELU_DG, out2=out2, tile=cfg)
    torch.cuda.synchronize()
    ref = _gelu(u)
    assert (out.float() - ref).abs().max().item() <= 2 ** -6 * max(1.0, ref.abs().max().item())
    cdf = 0.5 * (1 + torch.erf(u / math.sqrt(2.0)))
    dref = cdf + u * torch.exp(-0.5 * u * u) / math.sqrt(2 * math.pi)
    assert (out2.float() - dref).abs().max().item() <= 2 ** -6


@pytest.mark.parametrize("K", [64, 192, 768, 3072])
def test_gemm_nt_ping_pong_loop_is_race_free_and_bit_equal_to_the_ring_loop(K):
    """Tile configuration 8 (two half-workgroups one barrier apart, hand-counted LDS-DMA completion) accumulates every
    output element in the same K order as configuration 3 (lock-step ring): the results must be bit-identical, on a
    grid of several persistent rounds, every one of 8 repetitions (a staging race shows up as a sporadic mismatch)."""
    from unimm_amd import lib
    M, N = 31162, 768
    g = torch.Generator(device="cuda").manual_seed(K)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda")
    ref = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    lib.gemm_nt(x, w, ref, bias=bias, epilogue=lib.EPI_BIAS, tile=3)
    for rep in range(8):
        out = torch.zeros_like(ref)
        lib.gemm_nt(x, w, out, bias=bias, epilogue=lib.EPI_BIAS, tile=8)
        torch.cuda.synchronize()
        bad = (out.view(torch.int16) != ref.view(torch.int16)).sum().item()
        assert bad == 0, (rep, bad)


@pytest.mark.parametrize("K", [64, 128, 192, 768, 3072])
def test_gemm_nt_three_slot_x_ring_is_race_free_and_bit_equal_to_the_two_slot_ring(K):
    """Tile configuration 12 (192x256, the X operand on a ring of three slots: X(t+2) in flight while step t computes, one
    counted vmcnt per step) accumulates every output element in the same K order as configuration 6 (both operands on two
    slots): bit-identical results, on a grid of several persistent rounds with a ragged M edge, every one of 8 repetitions, from
    1 to 48 K steps (a staging race -- a slot refilled before its last reader, a fragment read before its DMA landed -- shows
    up as a sporadic mismatch), with the heaviest epilogue in between to vary the timing."""
    from unimm_amd import lib
    M, N = 31162, 768
    g = torch.Generator(device="cuda").manual_seed(K)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda")
    ref = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    lib.gemm_nt(x, w, ref, bias=bias, epilogue=lib.EPI_BIAS, tile=6)
    resid = torch.randn((M, N), generator=g, device="cuda")
    ref32, out32 = torch.empty((M, N), device="cuda"), torch.empty((M, N), device="cuda")
    lib.gemm_nt(x, w, ref32, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=resid, tile=6)
    for rep in range(8):
        out = torch.zeros_like(ref)
        lib.gemm_nt(x, w, out, bias=bias, epilogue=lib.EPI_BIAS, tile=12)
        lib.gemm_nt(x, w, out32, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=resid, tile=112 if rep % 2 else 212)
        torch.cuda.synchronize()
        bad = (out.view(torch.int16) != ref.view(torch.int16)).sum().item()
        assert bad == 0, (rep, bad)
        assert torch.equal(out32, ref32), rep


@pytest.mark.parametrize("M,N,K", [(515, 768, 128), (4096, 1024, 1024), (300, 200, 64)])
def test_gemm_nt_residual_from_lazy_layernorm(M, N, K):
    """aux_mean/rstd/gamma/beta: the residual operand is LayerNorm(aux) evaluated in the epilogue; must equal
    the same launch fed with the materialised fp32 LayerNorm output."""
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(M + N)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda")
    ldo = (N + 7) // 8 * 8
    h = torch.randn((M, ldo), generator=g, device="cuda") * 3 + 1
    gamma, beta = torch.randn(N, generator=g, device="cuda"), torch.randn(N, generator=g, device="cuda")
    mean = h[:, :N].mean(1)
    rstd = torch.rsqrt(h[:, :N].var(1, unbiased=False) + 1e-12)
    y = torch.zeros_like(h)
    y[:, :N] = (h[:, :N] - mean[:, None]) * rstd[:, None] * gamma + beta
    drop = (12345, int(0.1 * 2 ** 32), 1.0 / 0.9)
    a = torch.full((M, ldo), float("nan"), device="cuda")
    b = torch.full((M, ldo), float("nan"), device="cuda")
    lib.gemm_nt(x, w, a, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=y, N=N, drop=drop)
    lib.gemm_nt(x, w, b, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=h, N=N, drop=drop, aux_ln=(mean, rstd, gamma, beta))
    torch.cuda.synchronize()
    assert (a[:, :N] - b[:, :N]).abs().max().item() <= 1e-5 * max(1.0, a[:, :N].abs().max().item())
    with pytest.raises(lib.UnimmHipError):      # only with the residual epilogue
        lib.gemm_nt(x, w, b, bias=bias, epilogue=lib.EPI_BIAS, N=N, aux_ln=(mean, rstd, gamma, beta))


def test_gemm_tn_grouped_matches_single_launches():
    """One grouped call over mixed problems (big-tile and small-tile classes, different M, ragged M tails,
    ragged N/K, a bias gradient on some, two problems accumulating into ONE dw as the tied decoder /
    embedding weight does) == the same problems one by one == fp32 torch."""
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(77)
    shapes = [(9000, 768, 768, True), (9000, 2304, 768, True), (4133, 1024, 1024, False), (9000, 768, 3072, False),
              (500, 200, 136, True), (37 * 7, 1601, 1024, True), (240, 2, 768, False), (5000, 512, 256, False),
              (4096, 256, 512, True), (7777, 320, 264, False), (6000, 1024, 768, True), (4100, 768, 1024, False),
              (8192, 3072, 768, True), (64, 64, 64, False)]            # 14 > 12: the big class needs two launches
    probs, refs = [], []
    for (M, N, K, wb) in shapes:
        dy = torch.zeros((M, (N + 7) // 8 * 8), device="cuda", dtype=torch.bfloat16)
        x = torch.zeros((M, (K + 7) // 8 * 8), device="cuda", dtype=torch.bfloat16)
        dy[:, :N] = _rand((M, N), g, 0.5)
        x[:, :K] = _rand((M, K), g, 0.5)
        dw = torch.randn((N, K), generator=g, device="cuda")
        db = torch.randn(N, generator=g, device="cuda") if wb else None
        refs.append((dw + dy[:, :N].float().t() @ x[:, :K].float(), None if db is None else db + dy[:, :N].float().sum(0)))
        probs.append((dy, x, dw, M, N, K, db))
    # shared accumulator: problems 0 and 1' both add into dw of problem 0
    dy2, x2 = _rand((4500, 768), g, 0.5), _rand((4500, 768), g, 0.5)
    probs.append((dy2, x2, probs[0][2], 4500, 768, 768, None))
    refs[0] = (refs[0][0] + dy2.float().t() @ x2.float(), refs[0][1])
    singles = [(dy, x, dw.clone(), M, N, K, None if db is None else db.clone()) for (dy, x, dw, M, N, K, db) in probs]
    singles[-1] = singles[-1][:2] + (singles[0][2],) + singles[-1][3:]
    lib.gemm_tn_grouped(probs)
    for (dy, x, dw, M, N, K, db) in singles:
        lib.gemm_tn(dy, x, dw, M=M, N=N, K=K, dbias=db)
    torch.cuda.synchronize()
    for (pr, sg, (rw, rb)) in zip(probs, singles, refs):
        tol = 2e-3 * max(1.0, rw.abs().max().item())
        assert (pr[2] - rw).abs().max().item() <= tol
        assert (sg[2] - rw).abs().max().item() <= tol
        if rb is not None:
            assert (pr[6] - rb).abs().max().item() <= 2e-3 * max(1.0, rb.abs().max().item())
    with pytest.raises(lib.UnimmHipError):                   # one bad problem rejects the whole group, nothing launched
        lib.gemm_tn_grouped([probs[0], (probs[1][0][:, 1:], probs[1][1], probs[1][2], 9000, 2303, 768, None)])


def test_gemm_rejects_bad_arguments():
    from unimm_amd import lib
    x = torch.zeros((128, 100), device="cuda", dtype=torch.bfloat16)      # K % 64 != 0
    w = torch.zeros((128, 100), device="cuda", dtype=torch.bfloat16)
    out = torch.zeros((128, 128), device="cuda")
    with pytest.raises(lib.UnimmHipError):
        lib.gemm_nt(x, w, out)
    with pytest.raises(lib.UnimmHipError):
        lib.gemm_nt(x.cpu(), w, out)


@pytest.mark.parametrize("cfg", [1, 3, 6, 7, 8, 12, 14, 15])
@pytest.mark.parametrize("epi", ["add", "mul", "bias16", "resid_drop", "resid_ln"])
def test_gemm_nt_epilogue_layout_against_torch(cfg, epi):
    """Every epilogue family on every instantiated tile (the automatic rule picks 6 for N = 768 / 2304 and 8 for
    N = 3072 on the hot path), ragged M / N with strides that allow and forbid the 16-byte path (and the
    epilogue-operand prefetch clamped at the edges), against torch on the same bf16 inputs; the dropout mask against
    the host mirror of the counter-based generator; "resid_ln" = the residual is a lazily evaluated LayerNorm."""
    from unimm_amd import lib
    from unimm_amd import dropout as DR
    g = torch.Generator(device="cuda").manual_seed(cfg * 10 + len(epi))
    for (M, N, K, ldo) in [(777, 1000, 192, 1000), (777, 1000, 192, 1004), (300, 250, 64, 256)]:
        x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
        base = x.float() @ w.float().t()
        bias = torch.randn(N, generator=g, device="cuda")
        if epi in ("add", "mul"):
            aux = torch.zeros((M, ldo), device="cuda", dtype=torch.bfloat16)
            aux[:, :N] = _rand((M, N), g)
            out = torch.full((M, ldo), 7.0, device="cuda", dtype=torch.bfloat16)
            lib.gemm_nt(x, w, out, epilogue=lib.EPI_ADD if epi == "add" else lib.EPI_MUL, aux=aux, N=N, tile=cfg)
            ref = base + aux[:, :N].float() if epi == "add" else base * aux[:, :N].float()
            tol = 2 ** -7
        elif epi == "bias16":
            out = torch.full((M, ldo), 7.0, device="cuda", dtype=torch.bfloat16)
            lib.gemm_nt(x, w, out, bias=bias, N=N, tile=cfg)
            ref, tol = base + bias, 2 ** -7
        else:
            aux = torch.randn((M, ldo), generator=g, device="cuda") * 2 + 0.5
            out = torch.full((M, ldo), 7.0, device="cuda")
            key = DR.make_key(3, 9, 77)
            drop = DR.drop_arg(0.1, key)
            resid, aux_ln = aux[:, :N], None
            if epi == "resid_ln":
                gamma, beta = torch.randn(N, generator=g, device="cuda"), torch.randn(N, generator=g, device="cuda")
                mean = aux[:, :N].mean(1)
                rstd = torch.rsqrt(aux[:, :N].var(1, unbiased=False) + 1e-12)
                resid, aux_ln = (aux[:, :N] - mean[:, None]) * rstd[:, None] * gamma + beta, (mean, rstd, gamma, beta)
            lib.gemm_nt(x, w, out, bias=bias, epilogue=lib.EPI_BIAS_DROP_RESID, aux=aux, N=N, drop=drop, aux_ln=aux_ln, tile=cfg)
            keep = torch.from_numpy(DR.keep_mask2d(key, drop[1], M, N)).cuda()
            ref, tol = (base + bias) * keep * drop[2] + resid, 2e-3
        torch.cuda.synchronize()
        err = (out[:, :N].float() - ref).abs().max().item()
        assert err <= tol * max(1.0, ref.abs().max().item()), (M, N, K, ldo, err)
        if ldo > N:
            assert (out[:, N:].float() == 7.0).all()                 # pad columns untouched


def test_gemm_nt_rejects_uninstantiated_tiles():
    """Tile codes 2, 4, 5 (the BK = 32 rings that are no longer built) and anything else above 10 fail at the call, not later."""
    from unimm_amd import lib
    x = torch.zeros((256, 64), device="cuda", dtype=torch.bfloat16)
    w = torch.zeros((256, 64), device="cuda", dtype=torch.bfloat16)
    out = torch.zeros((256, 256), device="cuda", dtype=torch.bfloat16)
    for bad in (2, 4, 5, 11, 13, 16, 308, -1):
        with pytest.raises(lib.UnimmHipError):
            lib.gemm_nt(x, w, out, tile=bad)


def test_gemm_tn_workspace_reducer_matches_atomic_path():
    """unimm_gemm_tn_grouped_ws: partial tiles through slabs + one last-arriver reducer per tile == the fp32-atomic path,
    on a group that mixes tile classes and ragged shapes, accumulating (+=) over repeated launches (the arrival counters
    must be back at zero after every launch), with the chip-sharing hint on and off, and with a workspace that is too
    small (falls back to atomics)."""
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(31)
    M = 9000
    shapes = [(768, 768), (2304, 768), (768, 3072), (1000, 520), (200, 72)]
    probs_a, probs_b, refs = [], [], []
    for (N, K) in shapes:
        ldy, ldx = (N + 7) // 8 * 8, (K + 7) // 8 * 8
        dy = torch.zeros((M, ldy), device="cuda", dtype=torch.bfloat16)
        x = torch.zeros((M, ldx), device="cuda", dtype=torch.bfloat16)
        dy[:, :N] = _rand((M, N), g)
        x[:, :K] = _rand((M, K), g)
        dwa, dwb = torch.zeros((N, K), device="cuda"), torch.zeros((N, K), device="cuda")
        dba, dbb = torch.zeros(N, device="cuda"), torch.zeros(N, device="cuda")
        probs_a.append((dy, x, dwa, M, N, K, dba))
        probs_b.append((dy, x, dwb, M, N, K, dbb))
        refs.append(dy[:, :N].float().t() @ x[:, :K].float())
    ws = torch.zeros(256 << 20, dtype=torch.uint8, device="cuda")
    for rep, shared in enumerate((True, False, True)):
        lib.gemm_tn_grouped(probs_a, shared=shared, ws=None)
        lib.gemm_tn_grouped(probs_b, shared=shared, ws=ws)
        torch.cuda.synchronize()
        for (dy, x, dwa, _, N, K, dba), (_, _, dwb, _, _, _, dbb), ref in zip(probs_a, probs_b, refs):
            scale = max(1.0, ref.abs().max().item()) * (rep + 1)
            assert (dwa - dwb).abs().max().item() <= 1e-4 * scale, (N, K, rep)
            assert (dwb - (rep + 1) * ref).abs().max().item() <= 3e-3 * scale, (N, K, rep)
            assert (dba - dbb).abs().max().item() <= 1e-3 * max(1.0, dba.abs().max().item())
        # the arrival counters (the first bytes of the workspace) are zero again
        assert int(ws[:16384].view(torch.int32).abs().sum()) == 0
    small = torch.zeros(1 << 20, dtype=torch.uint8, device="cuda")          # too small for the slabs: atomic fallback
    lib.gemm_tn_grouped(probs_b, shared=True, ws=small)
    torch.cuda.synchronize()
    assert (probs_b[0][2] - 4 * refs[0]).abs().max().item() <= 3e-3 * 4 * max(1.0, refs[0].abs().max().item())


@pytest.mark.parametrize("M,N,K,tile", [(3900, 768, 3072, 7), (3900, 768, 2304, 7), (3900, 768, 3072, 1), (1110, 1024, 3072, 7),
                                        (777, 200, 1024, 7), (3900, 3072, 768, 7)])
def test_gemm_nt_split_k_equals_unsplit(M, N, K, tile):
    """Split-K (unimm_gemm_nt_args.splitk: several workgroups per output tile, partial tiles meet in the caller's workspace, the
    last arriver runs the fused epilogue) against the unsplit launch, every epilogue of the small-batch regime, repeated launches
    on one workspace (the counters must come back to zero), forced 2 / 3 / 4 splits and the library's own choice."""
    from unimm_amd import dropout as DR
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    x, w = _rand((M, K), g), _rand((N, K), g, 0.05)
    bias = torch.randn(N, generator=g, device="cuda")
    aux16 = _rand((M, N), g)
    aux32 = torch.randn((M, N), generator=g, device="cuda")
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    drop = DR.drop_arg(0.1, DR.make_key(3, 1, 4))
    cases = [(lib.EPI_BIAS, torch.bfloat16, None, None), (lib.EPI_ADD, torch.bfloat16, aux16, None),
             (lib.EPI_MUL, torch.bfloat16, aux16, None), (lib.EPI_BIAS_DROP_RESID, torch.float32, aux32, drop),
             (lib.EPI_BIAS_GELU_DG, torch.bfloat16, None, None)]
    for epi, dt, aux, dr in cases:
        def run(splitk):
            out = torch.full((M, N), float("nan"), device="cuda").to(dt)
            out2 = torch.full((M, N), float("nan"), device="cuda").to(torch.bfloat16) if epi == lib.EPI_BIAS_GELU_DG else None
            lib.gemm_nt(x, w, out, bias=bias, epilogue=epi, aux=aux, out2=out2, drop=dr, tile=tile, splitk=splitk, splitk_ws=ws)
            return out, out2
        ref, ref2 = run(0)
        for sk in (2, 3, 4, -1, 2):
            got, got2 = run(sk)
            torch.cuda.synchronize()
            scale = float(ref.float().abs().max())
            err = float((got.float() - ref.float()).abs().max())
            # fp32 sums in another order, then (bf16 outputs) one rounding: at most one bf16 ulp of the largest value
            assert err <= (2e-5 if dt == torch.float32 else 2 ** -7) * scale, (epi, sk, err, scale)
            if got2 is not None:
                assert float((got2.float() - ref2.float()).abs().max()) <= 2 ** -7 * float(ref2.float().abs().max())
    assert int(ws[:16384].view(torch.int32).abs().max()) == 0          # tickets returned to zero
    # and against fp32 torch for one epilogue
    want = x.float() @ w.float().t() + bias
    got = torch.empty((M, N), device="cuda")
    lib.gemm_nt(x, w, got, bias=bias, tile=tile, splitk=4, splitk_ws=ws)
    torch.cuda.synchronize()
    assert float((got - want).abs().max()) <= 2e-3 * float(want.abs().max())


def test_gemm_nt_split_k_hand_off_under_uneven_load():
    """The split-K hand-off (write-through slab stores -> vmcnt(0) -> barrier -> ticket; last arriver: agent-scope acquire) with
    another stream keeping part of the chip busy, inputs ALTERNATING between launches on one workspace: a slab read that saw the
    previous launch's bytes (a stale L1 / L2 line) would be off by O(1), every word of every launch is checked."""
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(11)
    M, N, K = 3900, 768, 3072
    xs = [_rand((M, K), g), _rand((M, K), g, 3.0)]
    w = _rand((N, K), g, 0.05)
    aux = _rand((M, N), g)
    ws = torch.zeros(64 << 20, dtype=torch.uint8, device="cuda")
    refs = []
    for x in xs:
        r = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
        lib.gemm_nt(x, w, r, epilogue=lib.EPI_ADD, aux=aux, tile=1)
        refs.append(r.float())
    side = torch.cuda.Stream()
    xb, wb = _rand((2048, 1024), g), _rand((1024, 1024), g)
    ob = torch.empty((2048, 1024), device="cuda", dtype=torch.bfloat16)
    outs = [torch.empty((M, N), device="cuda", dtype=torch.bfloat16) for _ in range(8)]
    torch.cuda.synchronize()
    worst = 0.0
    for it in range(160):
        with torch.cuda.stream(side), lib.stream_scope(side):            # uneven load: small GEMMs beside the split launches
            for _ in range(3):
                lib.gemm_nt(xb, wb, ob, tile=7)
        o = outs[it % 8]
        lib.gemm_nt(xs[it % 2], w, o, epilogue=lib.EPI_ADD, aux=aux, tile=1, splitk=2 + (it % 3 == 0), splitk_ws=ws)
        if it % 8 == 7:
            torch.cuda.synchronize()
            for j in range(8):
                k = it - 7 + j
                ref = refs[k % 2]
                worst = max(worst, float((outs[j].float() - ref).abs().max() / ref.abs().max()))
    assert worst <= 2 ** -7, worst
    # ... and with TWO splits (what the engine uses) bit-identical from launch to launch: a + b does not depend on who arrives last
    a = torch.empty((M, N), device="cuda", dtype=torch.bfloat16)
    lib.gemm_nt(xs[0], w, a, epilogue=lib.EPI_ADD, aux=aux, tile=1, splitk=2, splitk_ws=ws)
    for _ in range(10):
        b2 = torch.empty_like(a)
        lib.gemm_nt(xs[0], w, b2, epilogue=lib.EPI_ADD, aux=aux, tile=1, splitk=2, splitk_ws=ws)
        assert torch.equal(a, b2)


def test_gemm_tn_overwrite_writes_what_the_atomic_path_adds_to_zeros():
    """unimm_gemm_tn_args.overwrite: a problem whose gradient is known to be zero and has no other contributor is WRITTEN (plain
    stores) -- bit-identical to the atomic path on a zeroed buffer where one workgroup reduces a tile, whatever the buffer held
    before (NaN-filled here); pad columns of a strided gradient stay untouched; the bias gradient still accumulates; and where
    the launch splits the reduction over several workgroups (short M against few tiles) the flag is ignored and += survives."""
    from unimm_amd import lib
    g = torch.Generator(device="cuda").manual_seed(77)
    M = 1500                                # < 2,048 reduction rows: the launch never splits the reduction (one workgroup per tile)
    shapes = [(768, 768), (2304, 768), (768, 3072), (1000, 520), (200, 72)]
    pa, pb = [], []
    for (N, K) in shapes:
        ldy, ldx, ldw = (N + 7) // 8 * 8, (K + 7) // 8 * 8, K + 4
        dy = torch.zeros((M, ldy), device="cuda", dtype=torch.bfloat16)
        x = torch.zeros((M, ldx), device="cuda", dtype=torch.bfloat16)
        dy[:, :N] = _rand((M, N), g)
        x[:, :K] = _rand((M, K), g)
        dwa = torch.zeros((N, ldw), device="cuda")
        dwb = torch.full((N, ldw), float("nan"), device="cuda")
        dwb[:, K:] = 7.0
        dba, dbb = torch.ones(N, device="cuda"), torch.ones(N, device="cuda")
        pa.append((dy, x, dwa[:, :K], M, N, K, dba, None, False))
        pb.append((dy, x, dwb[:, :K], M, N, K, dbb, None, True))
    lib.gemm_tn_grouped(pa, shared=True)
    lib.gemm_tn_grouped(pb, shared=True)
    torch.cuda.synchronize()
    for (_, _, dwa, _, N, K, dba, *_), (_, _, dwb, _, _, _, dbb, *_) in zip(pa, pb):
        assert torch.equal(dwa, dwb), (N, K, float((dwa - dwb).abs().max()))
        assert (dwb.as_strided((N, 4), (K + 4, 1), dwb.storage_offset() + K) == 7.0).all()
        assert (dba - dbb).abs().max().item() <= 1e-4 * max(1.0, dba.abs().max().item())     # (atomics from several tiles: order)
    # a launch that splits the reduction: 3 tiles of 256 x 256 against 40,000 rows -> several workgroups per tile, += kept
    M2, N2, K2 = 40000, 256, 768
    dy, x = _rand((M2, N2), g), _rand((M2, K2), g)
    base = torch.randn((N2, K2), generator=g, device="cuda")
    dw = base.clone()
    lib.gemm_tn_grouped([(dy, x, dw, M2, N2, K2, None, None, True)], shared=False)
    torch.cuda.synchronize()
    ref = base + dy.float().t() @ x.float()
    assert (dw - ref).abs().max().item() <= 3e-3 * ref.abs().max().item()
