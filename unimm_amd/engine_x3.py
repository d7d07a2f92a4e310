"""The fp32-accuracy engine ("fp32x3"): the same launch schedule as `unimm_amd.engine.Engine` with fp32 activations,
fp32 gradients and fp32-grade GEMMs, for callers that run the reference WITHOUT autocast
(dense_annotation_finetuning.py:253 calls `forward` in fp32 end to end; north_star gates fp32 results at 1e-3).

gfx950 has no fast fp32 matrix path, so every nn.Linear still runs on the bf16 MFMA GEMM kernels, over SPLIT operands
(include/unimm_hip.h "fp32x3"): x = hi + lo, an activation operand is [hi | lo | hi], a weight operand [hi | hi | lo], and
one GEMM over the three-plane reduction axis gives hi hi + lo hi + hi lo in an fp32 accumulator (~2^-16 relative).  Every
GEMM writes fp32; the kernels of csrc/x3ops.hip turn fp32 results into the next split operand (with the GELU / GELU' /
residual-join that the bf16 path fuses into GEMM epilogues), run LayerNorm / embedding / loss backward on fp32 gradients,
and compute the attention cores in fp32 on the vector ALUs.  Masks, the unpadded schedule, dropout counters, the row-sparse
decoder, the fp32 pooler / NSP heads, the loss kernels and the flat gradient arena are the base engine's.

The base engine's two-stream schedule (image side beside the text side), eager launches or the graph executor, no lazy
LayerNorm: this is the accuracy mode, ~0.3 of the bf16 engine's throughput.  There is no CPU / eager-PyTorch fallback here either."""
from __future__ import annotations

import math

import torch

from . import lib as L
from . import params as PM
from .engine import BF16, F32, Engine, _rup


class _Lin3:
    """One (possibly fused) linear of the fp32x3 mode: fp32 master weight view [N, K], its w-type split [N, 3 Kp], the
    transposed w-type split [K, 3 Np] (input-gradient operand), fp32 bias, gradient views."""
    __slots__ = ("w32", "w3", "wt3", "bias", "gw", "gb", "N", "K", "Np", "Kp")

    def __init__(self, w32, bias, gw, gb, device, make_wt=True):
        self.w32, self.bias, self.gw, self.gb = w32, bias, gw, gb
        self.N, self.K = w32.shape
        self.Np, self.Kp = _rup(self.N, 64), _rup(self.K, 64)
        self.w3 = torch.zeros((self.N, 3 * self.Kp), dtype=BF16, device=device)
        self.wt3 = torch.zeros((self.K, 3 * self.Np), dtype=BF16, device=device) if make_wt else None


class EngineX3(Engine):
    compute_dtype = "fp32x3"

    def __init__(self, model, cfg):
        super().__init__(model, cfg)
        self.dual_stream = True           # the base engine's two-stream schedule (image side beside the text side)
        self.lazy_ln = False
        self.splitk = True                # small batches: the long reductions (K = 3 x 768 ... 3 x 3072) as two workgroups per tile
        self.attn_planes = True      # attention kernels write the next GEMM's split operand themselves (matrix kernels; False for
                                     # A/B runs of the vector kernels: fp32 result + a split pass)

    # ------------------------------------------------------------------------------------------
    # weights: split copies instead of the bf16 copies
    # ------------------------------------------------------------------------------------------
    def _mk_lin(self, key, wnames, bnames, device, kpad=None, make_wt=True):
        _, w32, gw = self._fused(wnames)
        _, b32, gb = self._fused(bnames) if bnames else (None, None, None)
        self.lin[key] = _Lin3(w32, b32, gw, gb, device, make_wt)

    def _build_tables(self, device):
        super()._build_tables(device)
        cfg = self.cfg
        self.vemb_w32 = torch.zeros((cfg.v_hidden_size, self.vemb_k), dtype=F32, device=device)   # [W_feat | W_loc | 0]
        self.vemb_w3 = torch.zeros((cfg.v_hidden_size, 3 * self.vemb_k), dtype=BF16, device=device)

    def refresh_weights(self, force=False, cast=True):
        """fp32 arena -> w-type split copies (and their transposes for the input-gradient GEMMs)."""
        A = self.arena
        ver = self._weight_version()
        if not force and ver == self._w_version:
            return
        for lin in self.lin.values():
            L.x3_split(lin.w32, out3=lin.w3, rows=lin.N, cols=lin.K, cp=lin.Kp, wtype=True)
            if lin.wt3 is not None:
                L.x3_split_wt(lin.w32, lin.wt3, lin.N, lin.K, lin.Np)
        cfg = self.cfg
        F = cfg.v_feature_size
        v = "bert.v_embeddings."
        self.vemb_w32[:, :F].copy_(A.view(v + "image_embeddings.weight"))
        self.vemb_w32[:, F:F + 5].copy_(A.view(v + "image_location_embeddings.weight"))
        L.x3_split(self.vemb_w32, out3=self.vemb_w3, wtype=True)
        torch.add(A.view(v + "image_embeddings.bias"), A.view(v + "image_location_embeddings.bias"), out=self.vemb_b)
        self._w_version = ver

    # (enable_graphs is the base engine's: the fp32x3 launch sequence carries the same device-side row counts, loss
    #  denominators and dropout salt word through its kernels, so unimm_amd/graphs.py captures and replays it unchanged --
    #  round 5; configs[3] on 8 GPUs is 12-13 sequences per rank in this arithmetic class, where ~17 ms of host calls per
    #  eager step would otherwise bound the step)

    def _splitk(self, M, N, K):
        """(tile code, splitk, workspace) for a GEMM of the small-batch regime, or None.  Every reduction of this mode is three
        planes long (K = 2304 ... 9216) while a per-rank share of configs[3] has ~1.7k text rows: 162 tiles of 64x128, each a
        36-144 step chain.  Measured at 1,700 rows (profiles/r5*_x3_small_batch_gemm_microbench.txt): N = 768, K = 9216 78.7 us ->
        41.5 us as 128x128 tiles with four workgroups per tile; K = 2304 / 3072 22.3 / 30.2 -> 18.1 / 24.1 us on the 3-slot ring
        of the 64x128 tile (two K steps in flight), no split.  (Four splits: the sum's last bits depend on which split arrives
        last, like the atomically accumulated weight gradients.)"""
        if not self.splitk or self._on_side or self.gemm_tile != 0 or self._step_rows is None or self._step_rows >= self.small_rows:
            return None
        if M != self._step_rows or N > 1024 or K < 2048:
            return None
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        if K >= 6144 and 2 * t128 <= 512:
            dev = self.arena.flat.device
            key = int(torch.cuda.current_stream(dev).cuda_stream)
            ws = self._splitk_ws.get(key)
            if ws is None:
                ws = self._splitk_ws[key] = torch.zeros(self.splitk_ws_bytes, dtype=torch.uint8, device=dev)
            return 1, (4 if 4 * t128 <= 512 else 2), ws
        if ((M + 63) // 64) * ((N + 127) // 128) <= 512:
            return 9, 0, None
        return None

    # ------------------------------------------------------------------------------------------
    # helpers
    # ------------------------------------------------------------------------------------------
    def _op3(self, a, op=L.X3_COPY, b=None, want3=True, want32=False, rows=None, cols=None):
        """y = op(a, b) in fp32 -> (split operand [rows, 3 cp] | None, fp32 [rows, cols] | None)"""
        rows = a.shape[0] if rows is None else rows
        cols = a.shape[1] if cols is None else cols
        cp = _rup(cols, 64)
        o3 = torch.empty((rows, 3 * cp), dtype=BF16, device=a.device) if want3 else None
        o32 = torch.empty((rows, cols), dtype=F32, device=a.device) if want32 else None
        L.x3_split(a, out3=o3, out32=o32, op=op, b=b, rows=rows, cols=cols, cp=cp)
        return o3, o32

    def _split(self, a, rows=None, cols=None):
        return self._op3(a, rows=rows, cols=cols)[0]

    def _lin3(self, x3, lin, epi=L.EPI_BIAS, aux=None, drop=None, ldo=None, M=None, bias=True):
        M = x3.shape[0] if M is None else M
        out = torch.empty((M, ldo or lin.N), dtype=F32, device=x3.device)
        sk = self._splitk(M, lin.N, 3 * lin.Kp)
        skw = dict(tile=sk[0], splitk=sk[1], splitk_ws=sk[2]) if sk is not None else {}
        L.gemm_nt(x3, lin.w3, out, bias=lin.bias if bias else None, epilogue=epi, aux=aux, drop=drop, M=M, N=lin.N, K=3 * lin.Kp, **skw)
        return out

    def _wgrad3(self, dy3, x3, gw, M, N, K, Np, Kp, dbias=None, m_dev=None, xcol0=0):
        """dW += dY^T X on split operands: (hi, hi) + (lo, hi) + (hi, lo); the bias gradient = column sums of hi + lo."""
        xh, xl = x3[:, xcol0:xcol0 + K], x3[:, Kp + xcol0:Kp + xcol0 + K]
        self._wgrad(dy3[:, :N], xh, gw, M, N, K, dbias=dbias, m_dev=m_dev)
        self._wgrad(dy3[:, Np:Np + N], xh, gw, M, N, K, dbias=dbias, m_dev=m_dev)
        self._wgrad(dy3[:, :N], xl, gw, M, N, K, dbias=None, m_dev=m_dev)

    def _lin3_bwd(self, dy3, x3, lin, need_dx=True, bias_grad=True, M=None, m_dev=None, add=None):
        """dW += dy^T x ; db += colsum(dy) ; returns dx = dy @ W [+ add] (fp32) or None.  `add` (fp32 [M, K]): the other
        gradient that meets dx at a residual fork, added in the GEMM's residual epilogue instead of a pass of its own."""
        M = dy3.shape[0] if M is None else M
        self._wgrad3(dy3, x3, lin.gw, M, lin.N, lin.K, lin.Np, lin.Kp,
                     dbias=lin.gb if (bias_grad and lin.gb is not None) else None, m_dev=m_dev)
        if not need_dx:
            return None
        dx = torch.empty((M, lin.K), dtype=F32, device=dy3.device)
        sk = self._splitk(M, lin.K, 3 * lin.Np)
        skw = dict(tile=sk[0], splitk=sk[1], splitk_ws=sk[2]) if sk is not None else {}
        if add is None:
            L.gemm_nt(dy3, lin.wt3, dx, bias=None, M=M, N=lin.K, K=3 * lin.Np, **skw)
        else:
            L.gemm_nt(dy3, lin.wt3, dx, bias=None, epilogue=L.EPI_BIAS_DROP_RESID, aux=add, M=M, N=lin.K, K=3 * lin.Np, **skw)
        return dx

    def _ln3(self, x, key, save, drop=L.NO_DROP, want3=True):
        """x: fp32 pre-LayerNorm sum -> (y32, y3 split operand, mean, rstd)"""
        gmm, bta, _, _ = self.ln[key]
        M, H = x.shape
        y32 = torch.empty((M, H), dtype=F32, device=x.device)
        mean = torch.empty(M, dtype=F32, device=x.device) if save else None
        rstd = torch.empty(M, dtype=F32, device=x.device) if save else None
        if not want3:
            L.layernorm_fwd(x, gmm, bta, y32, None, mean, rstd, M, H, drop=drop)
            return y32, None, mean, rstd
        y3 = torch.empty((M, 3 * H), dtype=BF16, device=x.device)       # the next GEMM's operand, written by the same kernel
        L.x3_layernorm_fwd(x, gmm, bta, y32, y3, mean, rstd, M, H, drop=drop)
        return y32, y3, mean, rstd

    def _ln3_bwd(self, dy, x, mean, rstd, key, dbias=None, drop=L.NO_DROP, out_drop=L.NO_DROP, m_dev=None, want3=True, want32=True,
                 dbias2=None):
        """-> (dx32: gradient w.r.t. the pre-LayerNorm sum, dxd3: its dropout-masked copy as a split operand).  Column sums
        (dgamma, dbeta, dbias [, dbias2]) are reduced by the grouped launch at the end of the block."""
        gmm, _, gg, gb = self.ln[key]
        M, H = x.shape
        dx32 = torch.empty((M, H), dtype=F32, device=x.device) if want32 else None
        dxd3 = torch.empty((M, 3 * H), dtype=BF16, device=x.device) if want3 else None
        part = torch.empty(self.part[H].numel(), dtype=F32, device=x.device)
        blocks = L.x3_layernorm_bwd_partials(dy, x, mean, rstd, gmm, dx32, dxd3, part, M, H, drop=drop, out_drop=out_drop, m_dev=m_dev)
        fq = self._fq_img if self._on_side else self._fq
        fq.append((part, blocks, H, [gg, gb, dbias]))
        if dbias2 is not None:
            fq.append((part, blocks, H, [None, None, dbias2]))
        return dx32, dxd3

    def _attn(self, q, k, v, mask, B, H, Tq, Tk, D, drop, save, qvar=None, kvar=None, tag=None):
        """-> (context fp32, context as a split operand, log-sum-exp)"""
        out = torch.empty((q.shape[0], H * D), dtype=F32, device=q.device)
        lse = torch.empty((B, H, Tq), dtype=F32, device=q.device) if save else None
        words, mq, mb = mask
        out3 = torch.empty((q.shape[0], 3 * _rup(H * D, 64)), dtype=BF16, device=q.device) if self.attn_planes else None
        L.x3_attn_fwd(q, k, v, out, lse, words, B, H, Tq, Tk, D, 1.0 / math.sqrt(D), mq, mb, drop, qvar=qvar, kvar=kvar, out3=out3)
        if out3 is None:
            out3 = self._split(out)
        if self.attn_sink is not None:          # diagnostic output (output_all_attention_masks): bf16-operand probabilities
            if qvar is not None or kvar is not None:
                raise RuntimeError("attention probabilities are collected on the padded schedule only")
            probs = torch.empty((B, H, Tq, Tk), dtype=F32, device=q.device)
            L.attn_probs(q.to(BF16), k.to(BF16), probs, words, B, H, Tq, Tk, D, 1.0 / math.sqrt(D), mq, mb, drop)
            self.attn_sink[tag] = probs
        return out, out3, lse

    def _qkv_grad(self, qkv):
        """Gradient buffer of a fused projection output qkv (fp32 [rows, N], N % 64 == 0): with the matrix attention kernels a
        split operand [rows, 3 N] whose planes the attention backward fills directly, otherwise fp32 (split afterwards)."""
        if self.attn_planes:
            return torch.empty((qkv.shape[0], 3 * qkv.shape[1]), dtype=BF16, device=qkv.device)
        return torch.empty_like(qkv)

    def _qkv_grad3(self, g):
        return g if self.attn_planes else self._split(g)

    def _attn_bwd(self, q, k, v, ctx, dctx, lse, mask, dq, dk, dv, N, B, H, Tq, Tk, D, drop, qvar=None, kvar=None):
        """dq / dk / dv: column slices [:, c0:c1] of `_qkv_grad` buffers (N = their projection width = the plane stride)."""
        delta = torch.empty_like(lse)
        words, mq, mb = mask
        L.x3_attn_bwd(q, k, v, ctx, dctx, lse, delta, dq, dk, dv, words, B, H, Tq, Tk, D, 1.0 / math.sqrt(D), mq, mb, drop,
                      qvar=qvar, kvar=kvar, planes=(N, dq.stride(0)) if self.attn_planes else None)

    # ------------------------------------------------------------------------------------------
    # blocks
    # ------------------------------------------------------------------------------------------
    def _self_block(self, key, x32, x3, mask, B, T, heads, pname, p_attn, p_hid, st, var=None):
        """BertLayer / BertImageLayer (models/vilbert_dialog.py:385-483, :514-612).  (x32, x3): the fp32 residual stream and
        its split copy (the GEMM operand)."""
        train, tape = st["train"], st["tape"]
        save = tape is not None
        Hd = x32.shape[1]
        D = Hd // heads
        qkv_l, so, ff1, ff2 = (self.lin[key + s] for s in (".qkv", ".so", ".ff1", ".ff2"))
        qkv = self._lin3(x3, qkv_l)
        q, k, v = qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:]
        d_attn = self._drop(pname + "attn", p_attn, train)
        ctx, ctx3, lse = self._attn(q, k, v, mask, B, heads, T, T, D, d_attn, save, qvar=var, kvar=var, tag=key)
        d_so = self._drop(pname + "so", p_hid, train)
        pre1 = self._lin3(ctx3, so, L.EPI_BIAS_DROP_RESID, aux=x32, drop=d_so)
        x1_32, x1_3, m1, r1 = self._ln3(pre1, key + ".ln1", save)
        u = self._lin3(x1_3, ff1)                                   # pre-activation, fp32
        h3 = self._op3(u, op=L.X3_GELU)[0]
        d_out = self._drop(pname + "out", p_hid, train)
        pre2 = self._lin3(h3, ff2, L.EPI_BIAS_DROP_RESID, aux=x1_32, drop=d_out)
        x2_32, x2_3, m2, r2 = self._ln3(pre2, key + ".ln2", save)
        md = var[2] if var is not None else None
        if save:
            def bwd(dx2):
                dpre2, dpre2d3 = self._ln3_bwd(dx2, pre2, m2, r2, key + ".ln2", dbias=ff2.gb, drop=d_out, m_dev=md)
                du_t = self._lin3_bwd(dpre2d3, h3, ff2, bias_grad=False, m_dev=md)
                du3 = self._op3(du_t, op=L.X3_MUL_DGELU, b=u)[0]
                dx1 = self._lin3_bwd(du3, x1_3, ff1, m_dev=md, add=dpre2)
                dpre1, dpre1d3 = self._ln3_bwd(dx1, pre1, m1, r1, key + ".ln1", dbias=so.gb, drop=d_so, m_dev=md)
                dctx = self._lin3_bwd(dpre1d3, ctx3, so, bias_grad=False, m_dev=md)
                dqkv = self._qkv_grad(qkv)
                self._attn_bwd(q, k, v, ctx, dctx, lse, mask, dqkv[:, :Hd], dqkv[:, Hd:2 * Hd], dqkv[:, 2 * Hd:3 * Hd], 3 * Hd,
                               B, heads, T, T, D, d_attn, qvar=var, kvar=var)
                return self._lin3_bwd(self._qkv_grad3(dqkv), x3, qkv_l, m_dev=md, add=dpre1)
            tape.append((key, bwd))
        return x2_32, x2_3

    def _conn_block(self, key, i, xv32, xv3, xt32, xt3, B, R, T, vmask, comask, st, var=None):
        """BertConnectionLayer (models/vilbert_dialog.py:655-783)."""
        cfg = self.cfg
        train, tape = st["train"], st["tape"]
        save = tape is not None
        pn = f"bert.encoder.c_layer.{i}."
        Hb, nh = cfg.bi_hidden_size, cfg.bi_num_attention_heads
        D = Hb // nh
        lq1, lq2, d1, d2 = (self.lin[key + s] for s in (".qkv1", ".qkv2", ".d1", ".d2"))
        vff1, vff2, tff1, tff2 = (self.lin[key + s] for s in (".vff1", ".vff2", ".tff1", ".tff2"))
        # The two halves run on their own streams (`_img()` = image side, otherwise the text side); the only exchanges are the
        # other side's K / V for the two co-attention directions (the base engine's schedule).
        with self._img():
            qkv1 = self._lin3(xv3, lq1)       # image side  [B*R, 3Hb]
        qkv2 = self._lin3(xt3, lq2)           # text side   [rows, 3Hb]
        self._to_txt(qkv1)
        self._to_img(qkv2)
        q1, k1, v1 = qkv1[:, :Hb], qkv1[:, Hb:2 * Hb], qkv1[:, 2 * Hb:]
        q2, k2, v2 = qkv2[:, :Hb], qkv2[:, Hb:2 * Hb], qkv2[:, 2 * Hb:]
        da1 = self._drop(pn + "attn1", cfg.v_attention_probs_dropout_prob, train)
        da2 = self._drop(pn + "attn2", cfg.attention_probs_dropout_prob, train)
        db1 = self._drop(pn + "bo1", cfg.v_hidden_dropout_prob, train)
        db2 = self._drop(pn + "bo2", cfg.hidden_dropout_prob, train)
        dvo = self._drop(pn + "vout", cfg.v_hidden_dropout_prob, train)
        dto = self._drop(pn + "tout", cfg.hidden_dropout_prob, train)
        # image half: regions attend text (:701-721), BertBiOutput (:744-754, call order :775), image FFN
        with self._img():
            ctx_v, ctx_v3, lse_v = self._attn(q1, k2, v2, comask, B, nh, R, T, D, da2, save, kvar=var, tag=key + "/2")
            prev = self._lin3(ctx_v3, d1, L.EPI_BIAS_DROP_RESID, aux=xv32, drop=db1)
            av32, av3, mv1, rv1 = self._ln3(prev, key + ".lnb1", save)
            uv = self._lin3(av3, vff1)
            hv3 = self._op3(uv, op=L.X3_GELU)[0]
            prev2 = self._lin3(hv3, vff2, L.EPI_BIAS_DROP_RESID, aux=av32, drop=dvo)
            ov32, ov3, mv2, rv2 = self._ln3(prev2, key + ".lnv", save)
        # text half: text attends regions (:681-698)
        ctx_t, ctx_t3, lse_t = self._attn(q2, k1, v1, vmask, B, nh, T, R, D, da1, save, qvar=var, tag=key + "/1")
        pret = self._lin3(ctx_t3, d2, L.EPI_BIAS_DROP_RESID, aux=xt32, drop=db2)
        at32, at3, mt1, rt1 = self._ln3(pret, key + ".lnb2", save)
        ut = self._lin3(at3, tff1)
        ht3 = self._op3(ut, op=L.X3_GELU)[0]
        pret2 = self._lin3(ht3, tff2, L.EPI_BIAS_DROP_RESID, aux=at32, drop=dto)
        ot32, ot3, mt2, rt2 = self._ln3(pret2, key + ".lnt", save)
        md = var[2] if var is not None else None
        if save:
            def bwd(dov, dot):
                # gradient buffers of the two projections: each is written by BOTH attention backward kernels (every slice exactly
                # once), i.e. from both streams -> allocate first and let each stream see the other's
                with self._img():
                    dqkv1 = self._qkv_grad(qkv1)
                dqkv2 = self._qkv_grad(qkv2)
                self._to_txt(dqkv1)
                self._to_img(dqkv2)
                with self._img():                                   # image half
                    dp, dpd3 = self._ln3_bwd(dov, prev2, mv2, rv2, key + ".lnv", dbias=vff2.gb, drop=dvo)
                    duv3 = self._op3(self._lin3_bwd(dpd3, hv3, vff2, bias_grad=False), op=L.X3_MUL_DGELU, b=uv)[0]
                    dav = self._lin3_bwd(duv3, av3, vff1, add=dp)
                    dprev, dprevd3 = self._ln3_bwd(dav, prev, mv1, rv1, key + ".lnb1", dbias=d1.gb, drop=db1)
                    dctx_v = self._lin3_bwd(dprevd3, ctx_v3, d1, bias_grad=False)
                    self._attn_bwd(q1, k2, v2, ctx_v, dctx_v, lse_v, comask, dqkv1[:, :Hb], dqkv2[:, Hb:2 * Hb], dqkv2[:, 2 * Hb:3 * Hb],
                                   3 * Hb, B, nh, R, T, D, da2, kvar=var)
                # text half
                dp, dpd3 = self._ln3_bwd(dot, pret2, mt2, rt2, key + ".lnt", dbias=tff2.gb, drop=dto, m_dev=md)
                dut3 = self._op3(self._lin3_bwd(dpd3, ht3, tff2, bias_grad=False, m_dev=md), op=L.X3_MUL_DGELU, b=ut)[0]
                dat = self._lin3_bwd(dut3, at3, tff1, m_dev=md, add=dp)
                dpret, dpretd3 = self._ln3_bwd(dat, pret, mt1, rt1, key + ".lnb2", dbias=d2.gb, drop=db2, m_dev=md)
                dctx_t = self._lin3_bwd(dpretd3, ctx_t3, d2, bias_grad=False, m_dev=md)
                self._attn_bwd(q2, k1, v1, ctx_t, dctx_t, lse_t, vmask, dqkv2[:, :Hb], dqkv1[:, Hb:2 * Hb], dqkv1[:, 2 * Hb:3 * Hb],
                               3 * Hb, B, nh, T, R, D, da1, qvar=var)
                self._to_img()                                      # dK1 / dV1 written by the text side
                self._to_txt()                                      # dK2 / dV2 written by the image side
                with self._img():
                    dxv = self._lin3_bwd(self._qkv_grad3(dqkv1), xv3, lq1, add=dprev)
                dxt = self._lin3_bwd(self._qkv_grad3(dqkv2), xt3, lq2, m_dev=md, add=dpret)
                return dxv, dxt
            tape.append((key, bwd))
        return ov32, ov3, ot32, ot3

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _forward(self, inp: dict, train: bool, save: bool, lm_rows: str, want_pred_v: bool):
        cfg = self.cfg
        if cfg.fixed_t_layer or cfg.fixed_v_layer or not cfg.with_coattention:
            raise NotImplementedError("fixed_t_layer / fixed_v_layer / with_coattention=False run on the bf16 engine only")
        dev = self.arena.device
        self.refresh_weights()
        ids = inp["input_ids"]
        B, T = ids.shape
        feat = inp["image_feat"]
        R = feat.shape[1]
        img_idx = inp.get("image_index")
        if img_idx is not None:
            img_idx = img_idx.to(dev, dtype=torch.int64, non_blocking=True).reshape(-1)
            if img_idx.numel() != B:
                raise ValueError("image_index needs one entry per sequence")
        elif feat.shape[0] != B:
            raise ValueError(f"image_feat has {feat.shape[0]} rows for {B} sequences and no image_index was given")
        if T > 256 or R > 256:
            raise ValueError("sequence / region count above 256 is not supported by the attention kernels")
        H, Hv = cfg.hidden_size, cfg.v_hidden_size
        st = dict(train=train, tape=[] if save else None)
        tape = st["tape"]
        tmask, vmask, comask = self._prep_masks(inp, B, T, R, dev)
        pl = self._prep_plan(inp, B, T, R, tmask, comask, lm_rows, dev)
        ids32, typ32, pos32, labels = pl["ids32"], pl["typ32"], pl["pos32"], pl["labels"]
        il32, plan, sel, n_img, dyn, st_nspw, var, Mt = (pl[k] for k in ("il32", "plan", "sel", "n_img", "dyn", "st_nspw", "var", "Mt"))
        A = self.arena
        F = cfg.v_feature_size

        # ---- image embedding (models/vilbert_dialog.py:1487-1493): one GEMM over [feat | loc | 0], on the image stream -
        self._to_img()                        # masks / plan are enqueued (and the previous step is behind us)
        with self._img():
            featd = feat.to(dev, dtype=F32, non_blocking=True)
            locd = inp["image_loc"].to(dev, dtype=F32, non_blocking=True)
            if img_idx is not None:
                featd, locd = featd.index_select(0, img_idx), locd.index_select(0, img_idx)
            packed32 = torch.zeros((B * R, self.vemb_k), dtype=F32, device=dev)
            packed32[:, :F].copy_(featd.reshape(B * R, F))
            packed32[:, F:F + 5].copy_(locd.reshape(B * R, 5))
            packed3 = self._split(packed32)
            prev = torch.empty((B * R, Hv), dtype=F32, device=dev)
            L.gemm_nt(packed3, self.vemb_w3, prev, bias=self.vemb_b, M=B * R, N=Hv, K=3 * self.vemb_k)
            d_embv = self._drop("emb_v", cfg.hidden_dropout_prob, train)
            xv32, xv3, mv, rv = self._ln3(prev, "emb_v", save, drop=d_embv)
            if save:
                v = "bert.v_embeddings."

                def bwd_embv(dxv):
                    _, dpre3 = self._ln3_bwd(dxv, prev, mv, rv, "emb_v", dbias=A.grad(v + "image_embeddings.bias"), out_drop=d_embv,
                                             want32=False, dbias2=A.grad(v + "image_location_embeddings.bias"))
                    K3 = self.vemb_k
                    self._wgrad3(dpre3, packed3, A.grad(v + "image_embeddings.weight"), B * R, Hv, F, Hv, K3)
                    self._wgrad3(dpre3, packed3, A.grad(v + "image_location_embeddings.weight"), B * R, Hv, 5, Hv, K3, xcol0=F)

        # ---- text embeddings (:326-356) --------------------------------------------------------------------------------
        erows = plan["rows"] if plan is not None else None
        emd = plan["var"][2] if plan is not None else None
        gmm, bta, ggm, gbt = self.ln["emb_t"]
        d_embt = self._drop("emb_t", cfg.hidden_dropout_prob, train)
        scratch16 = torch.empty((Mt, H), dtype=BF16, device=dev)
        xt32 = torch.empty((Mt, H), dtype=F32, device=dev)
        tabs = (self.tab["word"], self.tab["pos"], self.tab["type"], self.tab["ext"])
        L.embed_fwd(ids32, pos32, typ32, *tabs, gmm, bta, xt32, scratch16, Mt, H, cfg.type_vocab_size, drop=d_embt, m_dev=emd, rows=erows)
        del scratch16
        xt3 = self._split(xt32)
        e = "bert.embeddings."
        if save:
            def bwd_embt(dxt):
                L.embed_bwd_f32(ids32, pos32, typ32, *tabs, gmm, bta, dxt, A.grad(e + "word_embeddings.weight"),
                                A.grad(e + "position_embeddings.weight"), A.grad(e + "token_type_embeddings.weight"),
                                A.grad(e + "token_type_embeddings_extension.weight"), ggm, gbt, self.part[H], Mt, H,
                                cfg.type_vocab_size, drop=d_embt, m_dev=emd, rows=erows)

        # ---- encoder (schedule of :842-929) ------------------------------------------------------------------------------
        for kind, i in PM.encoder_schedule(cfg):
            if kind == "v":
                with self._img():
                    xv32, xv3 = self._self_block(f"v{i}", xv32, xv3, vmask, B, R, cfg.v_num_attention_heads, f"bert.encoder.v_layer.{i}.",
                                                 cfg.v_attention_probs_dropout_prob, cfg.v_hidden_dropout_prob, st)
            elif kind == "t":
                xt32, xt3 = self._self_block(f"t{i}", xt32, xt3, tmask, B, T, cfg.num_attention_heads, f"bert.encoder.layer.{i}.",
                                             cfg.attention_probs_dropout_prob, cfg.hidden_dropout_prob, st, var=var)
            else:
                with self._conn_tag():
                    xv32, xv3, xt32, xt3 = self._conn_block(f"c{i}", i, xv32, xv3, xt32, xt3, B, R, T, vmask, comask, st, var=var)
            if save:
                tape[-1] = (kind, tape[-1][0], tape[-1][1])

        # ---- image head (:1001-1005, :1085-1088) -------------------------------------------------------------------------
        img = None
        pred_v_out = None
        if want_pred_v or inp.get("image_target") is not None:
            itr, idec = self.lin["imgtr"], self.lin["imgdec"]
            C = cfg.v_target_size
            with self._img():
                uvh = self._lin3(xv3, itr)
                tv = self._op3(uvh, op=L.X3_GELU, want3=False, want32=True)[1]
                _, hvn3, mh, rh = self._ln3(tv, "imgtr", save)
                pred_v = self._lin3(hvn3, idec, ldo=_rup(C, 4))
            pred_v_out = pred_v.view(B, R, -1)[:, :, :C]
            img = dict(tv=tv, u=uvh, hn=hvn3, mean=mh, rstd=rh, pred=pred_v)
        self._to_txt(xv32, xv3, img["pred"] if img is not None else None)       # the heads read both streams
        out = dict(seq_out_t=xt3, seq_out_v=xv3, seq32_t=xt32, seq32_v=xv32, B=B, T=T, R=R, plan=plan, Mt=Mt,
                   nsp_weight_host=st_nspw, n_img=n_img, img_label32=il32, dyn=dyn, img=img)
        if pred_v_out is not None:
            out["pred_v"] = pred_v_out
        # ---- poolers + NSP (:946-967, :1064-1070): the base engine's fp32 heads -----------------------------------------
        cls_idx_t = var[0] if var is not None else torch.arange(0, B * T, T, dtype=torch.int32, device=dev)
        cls_idx_v = torch.arange(0, B * R, R, dtype=torch.int32, device=dev)
        cls_t = torch.empty((B, H), dtype=F32, device=dev)
        cls_v = torch.empty((B, Hv), dtype=F32, device=dev)
        L.gather_rows(xt32.view(BF16), cls_idx_t, cls_t.view(BF16), B, 2 * H)
        L.gather_rows(xv32.view(BF16), cls_idx_v, cls_v.view(BF16), B, 2 * Hv)
        pooled_t = self._linear32(cls_t, "tpool", relu=True)
        pooled_v = self._linear32(cls_v, "vpool", relu=True)
        d_fuse = self._drop("fuse", 0.1, train)
        fused = torch.empty_like(pooled_t)
        L.mul_dropout(pooled_t, pooled_v, fused, fused.numel(), d_fuse, fusion_sum=cfg.fusion_method == "sum")
        nsp = torch.zeros((B, 4), dtype=F32, device=dev)
        self._linear32(fused, "nsp", out=nsp)
        out["nsp"] = nsp[:, :2]
        # ---- MLM head on the selected rows (:982-986, :1023-1026) --------------------------------------------------------
        V = cfg.vocab_size
        Vp = _rup(V, 64)
        lm = None
        if lm_rows == "labelled" and labels is not None:
            n = sel["n"]
            if n > 0:
                xs3 = torch.empty((n, 3 * H), dtype=BF16, device=dev)
                L.gather_rows(xt3, sel["idx"], xs3, n, 3 * H, n_dev=dyn["n_lm"])      # split rows move as 3 H 16-bit elements
                lm = self._lm_head(xs3, n, sel["label"], sel["weight"], save, n_dev=dyn["n_lm"])
                lm.update(idx=sel["idx"], pos_idx=sel["pos"], n=n, n_dev=dyn["n_lm"], inv_dev=dyn["inv_lm"])
            out["lm"] = lm
        elif lm_rows == "all":
            out["pred_t"] = self.decode_rows(self.padded(out, xt3), B * T).view(B, T, Vp)[:, :, :V]
        if save:
            out["bwd"] = dict(tape=tape, embt=bwd_embt, embv=bwd_embv, pooled_t=pooled_t, pooled_v=pooled_v, fused=fused,
                              d_fuse=d_fuse, nsp_pad=nsp, cls_t=cls_t, cls_v=cls_v, cls_idx_t=cls_idx_t, cls_idx_v=cls_idx_v)
        return out

    def _lm_head(self, xs3, n, lab_sel, w_sel, save, n_dev=None):
        cfg = self.cfg
        V = cfg.vocab_size
        Vp = _rup(V, 64)
        lmtr, dec = self.lin["lmtr"], self.lin["dec"]
        u = self._lin3(xs3, lmtr)
        t1 = self._op3(u, op=L.X3_GELU, want3=False, want32=True)[1]
        _, hn3, mean, rstd = self._ln3(t1, "lmtr", save)
        logits = self._lin3(hn3, dec, ldo=Vp)
        rowloss, rownll, lse = (torch.empty(n, dtype=F32, device=xs3.device) for _ in range(3))
        L.lm_loss_fwd(logits, lab_sel, w_sel, rowloss, rownll, lse, n, V, n_dev=n_dev)
        return dict(xs=xs3, t1=t1, u=u, hn=hn3, mean=mean, rstd=rstd, logits=logits, rowloss=rowloss, rownll=rownll,
                    lse=lse, labels=lab_sel, weights=w_sel)

    def decode_rows(self, x3, n):
        """MLM transform + decoder for n rows given as a split operand [n, 3 H]: fp32 logits [n, Vpad]."""
        u = self._lin3(x3, self.lin["lmtr"], M=n)
        t1 = self._op3(u, op=L.X3_GELU, want3=False, want32=True)[1]
        _, hn3, _, _ = self._ln3(t1, "lmtr", False)
        return self._lin3(hn3, self.lin["dec"], ldo=_rup(self.cfg.vocab_size, 64))

    # ------------------------------------------------------------------------------------------
    # backward
    # ------------------------------------------------------------------------------------------
    def _backward(self, out, g_lm, g_img, g_nsp, g_nsp_scores=None):
        cfg = self.cfg
        dev = self.arena.device
        bw = out["bwd"]
        B, T, R = out["B"], out["T"], out["R"]
        H, Hv = cfg.hidden_size, cfg.v_hidden_size
        self.arena.attach_grads()

        def gvec(g):
            return torch.zeros(1, dtype=F32, device=dev) if g is None else g.detach().to(F32).reshape(1).contiguous()

        dseq_t = torch.zeros((out["Mt"], H), dtype=F32, device=dev)
        # ---- image head ---------------------------------------------------------------------------------------------
        img = out["img"]
        C = cfg.v_target_size
        itr, idec = self.lin["imgtr"], self.lin["imgdec"]
        gimg = gvec(g_img)
        self._to_img(gimg, img["target"], img["lse"], img["label"])
        with self._img():                                    # image head: on the image stream, beside the MLM head's backward
            dpred3 = torch.empty((B * R, 3 * idec.Np), dtype=BF16, device=dev)
            if cfg.predict_feature:
                L.mse_loss_bwd(img["pred"], img["target"], img["label"], gimg, img["inv"], dpred3, B * R, C, split=True)
            else:
                L.x3_kl_loss_bwd(img["pred"], img["target"], img["label"], img["lse"], gimg, img["inv"], dpred3, B * R, C,
                                 inv_dev=img.get("inv_dev"))
            dhn_v = self._lin3_bwd(dpred3, img["hn"], idec)
            dtv, _ = self._ln3_bwd(dhn_v, img["tv"], img["mean"], img["rstd"], "imgtr", want3=False)
            duv3 = self._op3(dtv, op=L.X3_MUL_DGELU, b=img["u"])[0]
            dseq_v = self._lin3_bwd(duv3, out["seq_out_v"], itr)
        # ---- MLM head -----------------------------------------------------------------------------------------------
        lm = out.get("lm")
        if lm is not None:
            n, V = lm["n"], cfg.vocab_size
            lmtr, dec = self.lin["lmtr"], self.lin["dec"]
            nd = lm.get("n_dev")
            dlog3 = torch.empty((n, 3 * dec.Np), dtype=BF16, device=dev)
            L.x3_lm_loss_bwd(lm["logits"], lm["labels"], lm["weights"], lm["lse"], gvec(g_lm), 1.0 / n, dlog3, n, V, n_dev=nd,
                             inv_dev=lm.get("inv_dev"))
            dhn = self._lin3_bwd(dlog3, lm["hn"], dec, m_dev=nd)       # dE += dlog^T hn ; dbias ; dhn = dlog @ E
            dt1, _ = self._ln3_bwd(dhn, lm["t1"], lm["mean"], lm["rstd"], "lmtr", m_dev=nd, want3=False)
            du3 = self._op3(dt1, op=L.X3_MUL_DGELU, b=lm["u"])[0]
            dxs = self._lin3_bwd(du3, lm["xs"], lmtr, m_dev=nd)
            L.gather_rows(dxs.view(BF16), lm["idx"], dseq_t.view(BF16), n, 2 * H, scatter=True, n_dev=nd)
        # ---- NSP + poolers (fp32 heads of the base engine) ----------------------------------------------------------
        nlab, w0, w1 = out["nsp_state"]
        dnsp = torch.empty((B, 2), dtype=F32, device=dev)
        extra = None
        if g_nsp_scores is not None:
            extra = g_nsp_scores.detach().to(device=dev, dtype=F32).reshape(B, 2).contiguous()
        L.nsp_loss_bwd(bw["nsp_pad"], nlab, w0, w1, gvec(g_nsp), dnsp, B, extra=extra)
        dfused = self._linear32_bwd(dnsp, bw["fused"], "nsp")
        dpt, dpv = torch.empty_like(dfused), torch.empty_like(dfused)
        L.mul_dropout_bwd(bw["pooled_t"], bw["pooled_v"], dfused, dpt, dpv, dfused.numel(), bw["d_fuse"],
                          fusion_sum=cfg.fusion_method == "sum")
        dcls_t = self._linear32_bwd(dpt, bw["cls_t"], "tpool")
        L.x3_rows_add(dseq_t, bw["cls_idx_t"], dcls_t, B, H)
        dcls_v = self._linear32_bwd(dpv, bw["cls_v"], "vpool")
        self._to_img(dcls_v)                     # the image pooler's input gradient joins the image head's on the image stream
        with self._img():
            L.x3_rows_add(dseq_v, bw["cls_idx_v"], dcls_v, B, Hv)
        self._bucket_done("heads")
        # ---- encoder blocks in reverse --------------------------------------------------------------------------------
        gt, gv = dseq_t, dseq_v
        entries = list(reversed(bw["tape"]))
        self._backward_encoder(bw, entries, gt, gv)      # the base engine's block loop and bucket order
        self._to_txt()                                       # everything joined before the caller continues
        self._bucket_done("text_embeddings")
