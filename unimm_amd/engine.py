"""Hot-path engine: the explicit forward / backward schedule of the two-stream co-attention
transformer over the hand-written HIP kernels (unimm_amd.lib), with weights in a flat arena.

What the reference runs as ~600 eager torch ops per step through autograd
(models/vilbert_dialog.py:1359-1626) is here a fixed launch schedule:

  forward  : mask bit-pack -> text/image embeddings -> [T|V|C blocks in BertEncoder order] ->
             poolers/NSP, row-sparse MLM decoder + fused log-softmax/UL loss, region head + KL
  backward : the same blocks in reverse, each a handful of fused kernels; weight gradients are
             accumulated straight into the flat fp32 gradient arena.

Activations are bf16 [rows, hidden] row-major; GEMMs accumulate in fp32; LayerNorm statistics,
softmax, log-sum-exp and all loss arithmetic are fp32.  There is no CPU / eager fallback here: every
compute step is a C-ABI call and raises if the library is missing."""
from __future__ import annotations

import math
import zlib
from contextlib import contextmanager
from typing import Dict, List, Optional

import torch

from . import bucket_plan as BP
from . import dropout as DR
from . import lib as L
from . import params as PM
from .arena import FlatArena
from .inputs import DialogMaskSpec, HostStager, PackedMask

BF16 = torch.bfloat16
F32 = torch.float32


def _site(name: str) -> int:
    return zlib.crc32(name.encode()) & 0xFFFFFFFF


def _rup(x, m):
    return (x + m - 1) // m * m


# inputs that `stage_host_inputs` moves to the device (nsp_weight stays on the host: it is read there)
_STAGED_KEYS = ("input_ids", "token_type_ids", "position_ids", "masked_lm_labels", "lm_weight", "next_sentence_label", "image_label",
                "image_attention_mask", "attention_mask", "co_attention_mask", "image_loc", "image_feat", "image_target", "image_index")


class _LazyLN:
    """fp32 residual-stream value that is LayerNorm(x) but was never written: (x, mean, rstd, gamma, beta)."""
    __slots__ = ("x", "mean", "rstd", "gamma", "beta")

    def __init__(self, x, mean, rstd, gamma, beta):
        self.x, self.mean, self.rstd, self.gamma, self.beta = x, mean, rstd, gamma, beta


class _Lin:
    """One (possibly fused) linear: bf16 weight [N,K], transposed bf16 copy [K,Npad], fp32 bias + grads."""
    __slots__ = ("w", "wt", "bias", "gw", "gb", "N", "K", "gbs")

    def __init__(self, w, wt, bias, gw, gb, gbs=None):
        self.w, self.wt, self.bias, self.gw, self.gb = w, wt, bias, gw, gb
        self.N, self.K = w.shape
        self.gbs = gbs  # extra bias-gradient destinations (image embedding: b_feat and b_loc)


class Engine:
    def __init__(self, model, cfg):
        self.model = model
        self.cfg = cfg
        self.arena: Optional[FlatArena] = None
        self.seed = 0x5EED
        self.step = 0
        self._w_version = None
        self.lin: Dict[str, _Lin] = {}
        self.ln: Dict[str, tuple] = {}
        self.grad_bucket_hook = None     # set by the data-parallel wrapper: f(group_name, more_follow_at_once)
        self._anchor = None
        self.last_seq_t = None
        self.unpad = True                # run the text stream on valid rows only (see the plan step of _forward)
        self._wq = []                    # queued weight-gradient problems (text side) waiting for their grouped launch
        # Schedule options are plain attributes (set them on `model.engine` before the first step; bench.py has flags for
        # the ones that are measured: --single-stream).  No environment switches.
        self.lazy_ln = True              # residual epilogues evaluate the previous LayerNorm instead of reading its fp32 output
        # The attention launches take their (sequence, head) items longest sequence first (unimm_attn_args.order, written by the
        # step's plan): the tail of a launch is then its shortest items.  5,691 / 5,732 -> 5,793 / 5,818 sequences/s at 240
        # sequences with the BATCH sorted that way (bench.py --order-by-length, alternating runs); same results either way.
        self.attn_longest_first = True
        # Option: grouped weight-gradient launches on a side stream (`wgrad_stream = True`): +1.8 % throughput in
        # interleaved runs (61.2 -> 60.1 ms) because the next block's GEMMs fill the partial last round and the
        # atomic drain.  Off by default: overlapped kernels stretch each other's durations, so per-kernel event /
        # rocprof timings (bench.py's roofline block) would no longer be exclusive.
        self.wgrad_stream = False
        self._side = None
        # Image-stream blocks on their own HIP stream (`dual_stream`, default on): between two connection layers the
        # image layer and the text layer are independent (models/vilbert_dialog.py:842-929), and the image side's
        # kernels are too small to fill 256 CUs (M = B*37 rows), so they run beside the text layer's and fill its
        # partial rounds; the same holds for the image half of a connection layer once the two co-attention
        # directions have exchanged their K/V.  Same kernels, same order within each stream.
        self.dual_stream = True
        self._text_stream = None      # the stream of the running engine entry (see _on_text_stream / _img)
        self._text_scope = None       # ... and the raw-stream object lib.stream_scope installed for it
        self._vside = None
        self.text_priority = False       # text side on an internal high-priority stream (measured -0.6 %: off)
        self._tstream = None
        self._on_side = False            # inside `_img()`: launches (and queued weight gradients) belong to the image side
        self._wq_img = []
        self._fq, self._fq_img = [], []  # pending column-partials reductions of LayerNorm backward calls, per stream
        self.last_plan = None
        self._arena_users = []           # weakrefs of objects that cache `self.arena` (FusedAdamW, DataParallelRCCL)
        # Partial-tile workspaces of the weight-gradient launches, one per launch stream (text / image / optional side).
        # OFF by default (0 MB = every split adds its partial tile with fp32 atomics): the slab + last-arriver reducer
        # of unimm_gemm_tn_grouped_ws measured SLOWER at 240 sequences (4,797 against 4,949 sequences/s, interleaved
        # runs on one box; single problems 220 against 128 us): the last round's ~108 reducers each read 1.5 MB of
        # slabs serially and every split's agent-scope release writes back its XCD's L2.  `wgrad_ws_bytes = 512 << 20` enables it.
        self.wgrad_ws_bytes = 0
        self._wgrad_ws = {}
        self._plist = []
        self.gemm_tile = 0               # tuning code handed to every encoder GEMM (unimm_gemm_nt_args.tile; 0 = automatic)
        # Weight-gradient launches cover SEVERAL encoder blocks (see `_flush_due`): a launch is due once its 256x256 tiles
        # fill about `wgrad_group_rounds` rounds of the chip; a data-parallel wrapper may lower it (smaller, earlier buckets).
        self.wgrad_group_rounds = 4
        # Row-count capacities.  Launch arguments hold CAPACITIES (the step's valid text rows / decoded rows rounded up to
        # these buckets); kernels whose result depends on the surplus rows read the real counts from the step's own device
        # words (`out["dyn"]`, written by unimm_plan_build; allocated per forward).  1 = exact sizes (default); the graph executor (unimm_amd/graphs.py) raises them
        # so that steps with nearby row counts replay one captured launch sequence.
        self.row_bucket, self.lm_bucket = 1, 1
        self.salt_word = None            # int32 [1] device tensor holding dropout.step_salt(seed, step), or None (see _drop)
        self._inject_header = None       # header values of the step, read ahead of a graph capture (see _forward, graphs.py)
        self.graphs = None               # unimm_amd.graphs.StepGraphs once enable_graphs() was called
        self.image_tile = 0              # tile code of the image side's GEMMs in the large-batch regime (0 = the library's choice; see _tile)
        self.wgrad_overwrite = True      # weight gradients with a single contributor are WRITTEN into a freshly zeroed arena (no atomics)
        self._bwd_fresh = False
        self.debug_fresh = False         # check (one device scan + sync per backward) that a `fresh` arena really is zero
        self.tile_table = {}             # {("t" | "i", N, K): tile code}: per-shape choices of the small-batch regime (see _tile)
        self.small_rows = 12000          # text rows per step below which the encoder GEMMs take the small-batch tile rule (_tile)
        self._step_rows = None           # text rows of the running step (set by _forward)
        self.image_head_side = True      # image prediction head (forward and backward) on the image stream, beside the MLM head
        self.attn_sink = None            # dict while a forward collects attention probabilities (forward_with_attention)
        self.skinny_dx_rows = 3072       # decoded-row count up to which the decoder's input gradient runs as a split reduction (_decoder_dx)
        self.prof_conn = None            # bench.py: dict while the launch profiler tags the connection layers' GEMMs (see _conn_tag)
        # Split-K for the long reductions of the small-batch regime (see _splitk): workspace per launch stream, zero-filled once
        self.splitk = True
        self.splitk_ws_bytes = 48 << 20
        self._splitk_ws = {}
        self._pending = []               # gradient buckets whose weight gradients are still queued: (group, #text, #image queued)
        self._nq = [0, 0]                # weight-gradient problems ever queued (text side, image side) ...
        self._nf = [0, 0]                # ... and launched
        self._reported = set()
        self.host_staging = True         # CPU tensors handed to forward() go through a pinned staging ring + copy stream (stage_host_inputs)
        self._stager = None

    def schedule_key(self):
        """Every schedule attribute that is frozen into a captured launch sequence (unimm_amd/graphs.py keys its entries on
        it: changing one after a capture re-captures instead of replaying the old schedule beside an eager new one)."""
        return (self.dual_stream, self.unpad, self.lazy_ln, self.gemm_tile, self.wgrad_group_rounds, self.wgrad_stream, self.splitk,
                self.attn_longest_first, self.wgrad_overwrite, tuple(sorted(self.tile_table.items())), self.image_tile,
                self.small_rows, self.image_head_side, self.skinny_dx_rows, self.wgrad_ws_bytes, self.text_priority)

    def register_arena_user(self, obj):
        import weakref
        self._arena_users.append(weakref.ref(obj))

    # ------------------------------------------------------------------------------------------
    # arenas + bf16 weight copies
    # ------------------------------------------------------------------------------------------
    def ensure(self, device):
        device = torch.device(device)
        if device.type != "cuda":
            raise L.UnimmHipError("unimm_amd runs on the GPU only (gfx950); move the model with .cuda() first")
        L.lib()
        if self.arena is not None and self.arena.is_current() and self.arena.device == device:
            return
        if self.arena is not None:
            users = [u() for u in self._arena_users]
            users = [type(u).__name__ for u in users if u is not None]
            if users:
                raise L.UnimmHipError(
                    "the model's parameters were re-pointed (module.to(), load_state_dict(assign=True), a replaced "
                    f"Parameter) while {', '.join(users)} still hold the engine's flat arena: build the optimizer / "
                    "data-parallel wrapper after the model has reached its final device")
        self.cfg.validate_for_hip()
        named = {n: p for n, p in self.model.named_parameters()}   # tied decoder alias is deduplicated
        self.arena = FlatArena(named, PM.arena_groups(self.cfg), device=device, no_grad=PM.no_grad_names(self.cfg))
        A = self.arena
        self.w16 = torch.zeros(A.numel, dtype=BF16, device=device)
        self._anchor = torch.zeros(1, device=device, requires_grad=True)
        self._plist = list(named.values())
        self._build_tables(device)
        self._w_version = None
        self._wt_table = None

    def invalidate_weights(self):
        """The fp32 master weights changed behind the engine's back (load_state_dict, a broadcast, a raw-pointer
        write): the next forward re-casts the bf16 copies and rebuilds the transposed ones."""
        self._w_version = None

    def _weight_version(self):
        """Changes whenever a parameter (or the flat arena) was written through torch.  The Parameters are views
        created with `p.data = view`, so each has its OWN version counter: `p.copy_()`, `load_state_dict` and eager
        optimizers bump those, not `flat._version`.  Writes torch cannot see (`p.data.mul_()`, kernels taking raw
        pointers) need `invalidate_weights()`; the fused AdamW step refreshes the copies itself."""
        v = self.arena.flat._version
        for p in self._plist:
            v += p._version
        return v

    def _fused(self, names):
        """bf16 / fp32 / grad views spanning several consecutive arena entries (fused QKV)."""
        A = self.arena
        o0, shape0 = A.offsets[names[0]]
        rows = sum(A.offsets[n][1][0] for n in names)
        tail = shape0[1:]
        n_el = rows * (tail[0] if tail else 1)
        shp = (rows,) + tail
        # consecutive entries must be gap-free for the fused view to be valid
        o = o0
        for n in names:
            oo, s = A.offsets[n]
            assert oo == o, f"arena layout: {n} not contiguous"
            o += s[0] * (s[1] if len(s) > 1 else 1)
        return (self.w16[o0:o0 + n_el].view(shp), A.flat[o0:o0 + n_el].view(shp), A.grad_flat[o0:o0 + n_el].view(shp))

    def _mk_lin(self, key, wnames, bnames, device, kpad=None, make_wt=True):
        w16, _, gw = self._fused(wnames)
        _, b32, gb = self._fused(bnames) if bnames else (None, None, None)
        N, K = w16.shape
        wt = None
        if make_wt:
            wt = torch.zeros((K, kpad or _rup(N, 64)), dtype=BF16, device=device)
        self.lin[key] = _Lin(w16, wt, b32, gw, gb)
        self._wt_src.append((key, wnames))

    def _build_tables(self, device):
        cfg, A = self.cfg, self.arena
        self.lin.clear(); self.ln.clear()
        self._wt_src = []

        def ln(key, name):
            self.ln[key] = (A.view(name + ".weight"), A.view(name + ".bias"), A.grad(name + ".weight"), A.grad(name + ".bias"))

        def self_block(k, p):
            a = p + "attention.self."
            self._mk_lin(k + ".qkv", [a + n + ".weight" for n in ("query", "key", "value")],
                         [a + n + ".bias" for n in ("query", "key", "value")], device)
            self._mk_lin(k + ".so", [p + "attention.output.dense.weight"], [p + "attention.output.dense.bias"], device)
            ln(k + ".ln1", p + "attention.output.LayerNorm")
            self._mk_lin(k + ".ff1", [p + "intermediate.dense.weight"], [p + "intermediate.dense.bias"], device)
            self._mk_lin(k + ".ff2", [p + "output.dense.weight"], [p + "output.dense.bias"], device)
            ln(k + ".ln2", p + "output.LayerNorm")

        for kind, i in PM.encoder_schedule(cfg):
            if kind == "t":
                self_block(f"t{i}", f"bert.encoder.layer.{i}.")
            elif kind == "v":
                self_block(f"v{i}", f"bert.encoder.v_layer.{i}.")
            else:
                p = f"bert.encoder.c_layer.{i}."
                b = p + "biattention."
                k = f"c{i}"
                self._mk_lin(k + ".qkv1", [b + n + ".weight" for n in ("query1", "key1", "value1")],
                             [b + n + ".bias" for n in ("query1", "key1", "value1")], device)
                self._mk_lin(k + ".qkv2", [b + n + ".weight" for n in ("query2", "key2", "value2")],
                             [b + n + ".bias" for n in ("query2", "key2", "value2")], device)
                o = p + "biOutput."
                self._mk_lin(k + ".d1", [o + "dense1.weight"], [o + "dense1.bias"], device)
                ln(k + ".lnb1", o + "LayerNorm1")
                self._mk_lin(k + ".d2", [o + "dense2.weight"], [o + "dense2.bias"], device)
                ln(k + ".lnb2", o + "LayerNorm2")
                self._mk_lin(k + ".vff1", [p + "v_intermediate.dense.weight"], [p + "v_intermediate.dense.bias"], device)
                self._mk_lin(k + ".vff2", [p + "v_output.dense.weight"], [p + "v_output.dense.bias"], device)
                ln(k + ".lnv", p + "v_output.LayerNorm")
                self._mk_lin(k + ".tff1", [p + "t_intermediate.dense.weight"], [p + "t_intermediate.dense.bias"], device)
                self._mk_lin(k + ".tff2", [p + "t_output.dense.weight"], [p + "t_output.dense.bias"], device)
                ln(k + ".lnt", p + "t_output.LayerNorm")

        ln("emb_t", "bert.embeddings.LayerNorm")
        ln("emb_v", "bert.v_embeddings.LayerNorm")
        # poolers and the NSP head run in fp32 (unimm_linear_f32): fp32 master weight / bias and their gradient views
        self.lin32 = {k: (A.view(n + ".weight"), A.view(n + ".bias"), A.grad(n + ".weight"), A.grad(n + ".bias"))
                      for k, n in (("tpool", "bert.t_pooler.dense"), ("vpool", "bert.v_pooler.dense"),
                                   ("nsp", "cls.bi_seq_relationship"))}
        self._mk_lin("lmtr", ["cls.predictions.transform.dense.weight"], ["cls.predictions.transform.dense.bias"], device)
        ln("lmtr", "cls.predictions.transform.LayerNorm")
        self._mk_lin("dec", [PM.WORD_EMB], None, device)
        d = self.lin["dec"]
        d.bias, d.gb = A.view("cls.predictions.bias"), A.grad("cls.predictions.bias")
        self._mk_lin("imgtr", ["cls.imagePredictions.transform.dense.weight"],
                     ["cls.imagePredictions.transform.dense.bias"], device)
        ln("imgtr", "cls.imagePredictions.transform.LayerNorm")
        self._mk_lin("imgdec", ["cls.imagePredictions.decoder.weight"], ["cls.imagePredictions.decoder.bias"], device)
        # image embedding: one GEMM over [feat | loc | 0] with W_cat = [W_feat | W_loc | 0], bias = b_feat + b_loc
        F = cfg.v_feature_size
        self.vemb_k = F + 64
        self.vemb_w = torch.zeros((cfg.v_hidden_size, self.vemb_k), dtype=BF16, device=device)
        self.vemb_b = torch.zeros(cfg.v_hidden_size, dtype=F32, device=device)
        # embedding tables: gathered from the fp32 master weights
        e = "bert.embeddings."
        self.tab = {k: A.view(e + n + ".weight") for k, n in
                    (("word", "word_embeddings"), ("pos", "position_embeddings"), ("type", "token_type_embeddings"),
                     ("ext", "token_type_embeddings_extension"))}
        # column-partials scratch of the non-deferred LayerNorm / embedding backward calls: one set per stream (with
        # hidden_size == v_hidden_size the image stream's `bwd_embv` and the text stream's `bwd_embt` would otherwise
        # share one buffer without any ordering between them)
        self.part = {h: torch.empty(L.colpartials_bytes(h) // 4, dtype=F32, device=device)
                     for h in {cfg.hidden_size, cfg.v_hidden_size}}
        self.part_side = {h: torch.empty(L.colpartials_bytes(h) // 4, dtype=F32, device=device)
                          for h in {cfg.hidden_size, cfg.v_hidden_size}}

    def refresh_weights(self, force=False, cast=True):
        """fp32 arena -> bf16 copies (one cast kernel) + transposed copies for the dgrad GEMMs.
        cast=False: the caller (the fused AdamW step) has already written the bf16 copy."""
        A = self.arena
        ver = self._weight_version()
        if not force and ver == self._w_version:
            return
        if cast:
            L.cast_f32_bf16(A.flat, self.w16, A.numel)
        if self._wt_table is None:          # one descriptor table for every transposed copy, built once
            ents = []
            for key, wnames in self._wt_src:
                lin = self.lin[key]
                if lin.wt is None:
                    continue
                o0, _ = A.offsets[wnames[0]]
                ents.append((A.flat[o0:o0 + lin.N * lin.K].view(lin.N, lin.K), lin.wt))
            self._wt_table = L.transpose_table(ents, A.flat.device)
        L.transpose_cast_grouped(*self._wt_table)
        cfg = self.cfg
        F = cfg.v_feature_size
        v = "bert.v_embeddings."
        self.vemb_w[:, :F].copy_(A.view(v + "image_embeddings.weight"))
        self.vemb_w[:, F:F + 5].copy_(A.view(v + "image_location_embeddings.weight"))
        torch.add(A.view(v + "image_embeddings.bias"), A.view(v + "image_location_embeddings.bias"), out=self.vemb_b)
        self._w_version = ver

    # ------------------------------------------------------------------------------------------
    # small op helpers (each returns its output and pushes its backward onto the tape)
    # ------------------------------------------------------------------------------------------
    def _drop(self, name, p, train):
        """Dropout triple of one site.  Eagerly the key carries seed, step and site; with `salt_word` set (the graph
        executor) the argument is the per-site key and the per-step salt is read from that device word at kernel entry:
        the same masks either way (dropout.make_key = site_key ^ step_salt)."""
        if not train or p <= 0.0:
            return L.NO_DROP
        if self.salt_word is not None:
            return DR.drop_arg(p, DR.site_key(self.seed, _site(name))) + (self.salt_word,)
        return DR.drop_arg(p, DR.make_key(self.seed, self.step, _site(name)))

    def _linear(self, x, lin, epi=L.EPI_BIAS, aux=None, want_u=False, drop=None, out_f32=False, ldo=None, M=None):
        M = x.shape[0] if M is None else M
        ldo = ldo or lin.N
        out = torch.empty((M, ldo), dtype=F32 if out_f32 else BF16, device=x.device)
        u = torch.empty((M, ldo), dtype=BF16, device=x.device) if want_u else None
        aux_ln = None
        if isinstance(aux, _LazyLN):
            aux, aux_ln = aux.x, (aux.mean, aux.rstd, aux.gamma, aux.beta)
        sk = self._splitk(M, lin.N, lin.K)
        if sk is not None:
            L.gemm_nt(x, lin.w, out, bias=lin.bias, epilogue=epi, aux=aux, out2=u, drop=drop, M=M, N=lin.N, K=lin.K, aux_ln=aux_ln,
                      tile=sk[0], splitk=sk[1], splitk_ws=sk[2])
        else:
            L.gemm_nt(x, lin.w, out, bias=lin.bias, epilogue=epi, aux=aux, out2=u, drop=drop, M=M, N=lin.N, K=lin.K, aux_ln=aux_ln,
                      tile=self._tile(M, lin.N, lin.K))
        return (out, u) if want_u else out

    def _tile(self, M, N, K=None):
        """Tile code of one encoder GEMM (unimm_gemm_nt_args.tile).  0 = the kernel library's own choice, which is tuned for
        launches that have the chip to themselves.  In the small-batch regime (fewer than `small_rows` text rows in the step: the
        per-GPU share of a batch split over 4-8 ranks) neither stream's launches fill the chip and the two streams' kernels run
        side by side all the time; there the 128x128 tile (64 KiB of LDS, two workgroups per CU, 15 KiB staged per MFLOP) beats
        both the one-per-CU tiles (112-128 KiB of LDS: nothing of the other stream fits beside them) and, on the image side,
        the 64x128 tile (twice the workgroups, 23 KiB per MFLOP): +3 % at 60 sequences; the text side's N = 768 GEMMs at ~4k rows
        (fewer than 256 tiles of 128x128) stay on 64x128 (-1.7 % otherwise at 30 sequences).
        In the large-batch regime the text side's launches fill the chip with one-per-CU workgroups, and what the image side's
        launches (M = 37 rows per sequence: a fraction of a round) cost the step is the CUs they keep from the text side: in
        rounds 3-4 the 256x256 ping-pong tile (140 CUs per image GEMM instead of 188) was +0.8 % at 240 sequences; with the
        three-slot X ring of round 5 the library's own 192x256 choice is back in front (+0.3 %, two alternating pairs), so
        `image_tile` is 0 again."""
        if self.gemm_tile != 0:                   # an explicit tuning code reaches both sides (A/B runs: bench.py --gemm-tile)
            return self.gemm_tile
        if self._step_rows is None:               # outside a step: the library's own choice
            return 0
        if self.tile_table and K is not None:     # per-shape codes (side, N, K) of the small-batch regime (bench.py --tile-table)
            code = self.tile_table.get(("i" if self._on_side else "t", N, K))
            if code is not None and self._step_rows < self.small_rows:
                return code
        if self._step_rows >= self.small_rows:
            return self.image_tile if (self._on_side and self.image_tile) else 0
        if self._on_side:
            return 14                             # 128x128 with the X operand on a three-slot ring (round 5: +0.5 % at 30, +1 % at 60 over tile 1)
        t128 = ((M + 127) // 128) * ((N + 127) // 128)
        return 7 if t128 < 256 else 1

    def _splitk(self, M, N, K):
        """(tile code, splitk, workspace) of a long-reduction GEMM in the small-batch regime, or None.  At ~4k text rows a
        K = 2304 / 3072, N = 768 GEMM is 366 tiles of 64x128, each a 36-48 step chain, and stages 430 MB through the L2s (the W
        panel is re-read by 61 row tiles): 128x128 tiles halve the staged bytes but leave 186 workgroups for 256 CUs -- two
        workgroups per tile, each reducing half of K and meeting in a workspace (unimm_gemm_nt_args.splitk), get both: 30.3
        against 33.6 us alone (profiles/r4c_small_batch_gemm_microbench.txt)."""
        if not self.splitk or self._on_side or self.gemm_tile != 0 or self._step_rows is None or self._step_rows >= self.small_rows:
            return None
        if M != self._step_rows:      # the text stream's GEMMs only, whatever stream they run on (one-stream runs must round alike)
            return None
        if K < 2048 or N > 1024 or ((M + 127) // 128) * ((N + 127) // 128) > 256:
            return None
        dev = self.arena.flat.device
        # The workspace (slabs + zero-between-launches tickets) is private to launches of ONE stream: key it by the stream the
        # launch goes to (the caller's stream in eager steps, the capturing stream under graph capture), so a text-side GEMM
        # issued from another stream can never share tickets with one in flight
        key = int(torch.cuda.current_stream(dev).cuda_stream)
        ws = self._splitk_ws.get(key)
        if ws is None:
            ws = self._splitk_ws[key] = torch.zeros(self.splitk_ws_bytes, dtype=torch.uint8, device=dev)
        return 1, 2, ws

    def _linear32(self, x, key, relu=False, out=None):
        """y = act(x W^T + b) in fp32 from the fp32 master weights (poolers, NSP head; models/vilbert_dialog.py:946-967, :1070)."""
        w32, b32 = self.lin32[key][:2]
        N, K = w32.shape
        M = x.shape[0]
        if out is None:
            out = torch.empty((M, N), dtype=F32, device=x.device)
        return L.linear_f32(x, w32, out, M, N, K, (x.stride(0), 1), (1, w32.stride(0)), bias=b32, relu=relu)

    def _linear32_bwd(self, dy, x, key):
        """dW += dy^T x, db += colsum(dy) (fp32 atomics into the gradient arena), returns dx = dy W; all fp32."""
        w32, _, gw, gb = self.lin32[key]
        N, K = w32.shape
        M = dy.shape[0]
        L.linear_f32(dy, x, gw, N, K, M, (1, dy.stride(0)), (x.stride(0), 1), accumulate=True, rowsum=gb)
        dx = torch.empty((M, K), dtype=F32, device=dy.device)
        return L.linear_f32(dy, w32, dx, M, K, N, (dy.stride(0), 1), (w32.stride(0), 1))

    def _wgrad(self, dy, x, gw, M, N, K, dbias=None, m_dev=None, sole=False):
        """dW += dy^T x (+ bias gradient).  Nothing downstream in the backward chain reads a weight gradient,
        so the call is only queued; `_flush_wgrad` hands the list of SEVERAL encoder blocks to one grouped launch
        (`_flush_due` says when).  dy / x stay referenced by the queue until then.
        sole: this call is the ONLY contribution to gw in a backward pass (every encoder nn.Linear; not the tied decoder /
        word-embedding matrix, not the three split-operand products of the fp32x3 mode).  When the gradient arena is also known
        to be zero (`arena.fresh`: zeroed since the last backward) the kernel then writes the tile with plain stores instead of
        256 KiB of memory-side atomics per workgroup (unimm_gemm_tn_args.overwrite)."""
        ow = bool(sole and self._bwd_fresh and self.wgrad_overwrite)
        (self._wq_img if self._on_side else self._wq).append((dy, x, gw, M, N, K, dbias, m_dev, ow))
        self._nq[1 if self._on_side else 0] += 1
        if self.prof_conn is not None:            # FLOPs of the weight gradients queued from inside / outside a connection layer
            MM = dy.shape[0] if M is None else M
            NN = dy.shape[1] if N is None else N
            KK = x.shape[1] if K is None else K
            self.prof_conn["tn_conn" if self.prof_conn["in"] else "tn_other"] += 2.0 * MM * NN * KK

    @contextmanager
    def _conn_tag(self):
        """While bench.py profiles (`prof_conn` is a dict): the GEMM launches issued inside the block carry tag 1 in the launch
        profiler and the weight gradients queued inside it are counted as connection-layer FLOPs -- `roofline.coattention_gemms`,
        the projections and FFNs of models/vilbert_dialog.py:655-783, forward and backward."""
        if self.prof_conn is None:
            yield
            return
        L.prof_tag(1)
        self.prof_conn["in"] = True
        try:
            yield
        finally:
            self.prof_conn["in"] = False
            L.prof_tag(0)

    @staticmethod
    def _big_tiles(queue):
        """256x256 output tiles of the queue's problems that take the big tile (csrc/gemm.hip: tn_is_big)."""
        t = 0
        for dy, x, gw, M, N, K, *_ in queue:
            M = dy.shape[0] if M is None else M
            N = dy.shape[1] if N is None else N
            K = x.shape[1] if K is None else K
            t += BP.big_tiles_of(M, N, K)
        return t

    def _flush_due(self, queue):
        """Weight-gradient launches are grouped over blocks.  One block's 4-5 gradients are ~110 tiles: less than half a
        round of the 256 CUs, so the launcher had to split the reduction 7 ways and every one of the 756 workgroups ended
        by adding its 256 KiB partial tile with fp32 atomics (51 us per workgroup at the per-CU atomic rate against
        118 us of main loop; 8x the algorithmic write traffic).  Nothing waits for a weight gradient, so the queue
        simply keeps growing until its tiles fill whole rounds WITHOUT a split: each tile is then reduced by one
        workgroup over all rows and written once (round 3: 484 -> 394 us per text block at 7 blocks per launch).
        The rule itself is host arithmetic shared with the exchange planner: `bucket_plan.flush_due`."""
        return BP.flush_due(len(queue), self._big_tiles(queue), self.wgrad_group_rounds)

    def _ws(self, which):
        """Zero-initialised workspace of the weight-gradient launches of one stream (0 bytes = fp32-atomic path)."""
        if self.wgrad_ws_bytes <= 0:
            return None
        t = self._wgrad_ws.get(which)
        if t is None:
            t = self._wgrad_ws[which] = torch.zeros(self.wgrad_ws_bytes, dtype=torch.uint8, device=self.arena.flat.device)
        return t

    def _flush_wgrad(self, force=False, force_img=False):
        """Launch the queues that are due (all of them with force=True; force_img: the image side's whatever its size)."""
        shared = self._dual()                     # the launches of this backward share the chip with the other stream's
        if (self._wq_img or self._fq_img) and (force or force_img or self._flush_due(self._wq_img)):
            with self._img():                     # image-side problems: operands were produced on that stream
                L.gemm_tn_grouped(self._wq_img, shared=shared, ws=self._ws("img"))
                L.colpartials_finish_grouped(self._fq_img)
            self._wq_img, self._fq_img = [], []
            self._nf[1] = self._nq[1]
        if self._on_side:
            return                                # the text side's queues are flushed from the text side
        if not (force or self._flush_due(self._wq)):
            return
        L.colpartials_finish_grouped(self._fq)
        self._fq = []
        self._nf[0] = self._nq[0]
        if not self._wq:
            return
        if not self.wgrad_stream:
            L.gemm_tn_grouped(self._wq, shared=shared, ws=self._ws("txt"))
            self._wq = []
            return
        # side stream: the grouped launch ends with a partial last round and a memory-side atomic drain during
        # which most CUs idle; the next block's input-gradient GEMMs can fill them
        main = torch.cuda.current_stream()
        if self._side is None:
            self._side = torch.cuda.Stream(device=self.arena.flat.device)
        self._side.wait_stream(main)
        with torch.cuda.stream(self._side), L.stream_scope(self._side):
            L.gemm_tn_grouped(self._wq, shared=shared, ws=self._ws("side"))
        for dy, x, *_ in self._wq:              # keep the caching allocator from recycling them early
            dy.record_stream(self._side)
            x.record_stream(self._side)
        self._wq = []

    def _join_wgrad(self):
        if self._side is not None:
            torch.cuda.current_stream().wait_stream(self._side)

    def _linear_bwd(self, dy, x, lin, epi=L.EPI_BIAS, aux=None, need_dx=True, bias_grad=True, M=None, N=None, xk=None, m_dev=None):
        """dW += dy^T x ; db += colsum(dy) ; returns dx = epi(dy @ W).  m_dev: device word with the real row count when
        dy / x hold a capacity of rows (the surplus rows must not enter the reduction over rows)."""
        M = dy.shape[0] if M is None else M
        N = lin.N if N is None else N
        self._wgrad(dy, x, lin.gw, M, N, lin.K if xk is None else xk,
                    dbias=lin.gb if (bias_grad and lin.gb is not None) else None, m_dev=m_dev, sole=lin is not self.lin.get("dec"))
        if not need_dx:
            return None
        dx = torch.empty((M, lin.K), dtype=BF16, device=dy.device)
        kdim = lin.wt.shape[1]
        sk = self._splitk(M, lin.K, kdim)
        if sk is not None:
            L.gemm_nt(dy, lin.wt, dx, bias=None, epilogue=epi, aux=aux, M=M, N=lin.K, K=kdim, tile=sk[0], splitk=sk[1], splitk_ws=sk[2])
        else:
            L.gemm_nt(dy, lin.wt, dx, bias=None, epilogue=epi, aux=aux, M=M, N=lin.K, K=kdim, tile=self._tile(M, lin.K, kdim))
        return dx

    def _decoder_dx(self, dlog, dec, n, V):
        """dhn[n, H] = dlog[n, V] @ E[V, H] for FEW decoded rows (the per-GPU share of a split batch: ~630 rows at 30
        sequences).  As an NT GEMM this is 10 x 6 tiles of 64 x 128, each with a 477-step reduction over the vocabulary:
        60 workgroups on 256 CUs, 348 us -- 3.5 % of the 30-sequence step for 29 GFLOP.  The reduction axis is the long
        one, so it runs on the weight-gradient kernel instead (dW[N, K] += dY[M, N]^T X[M, K] with M = vocabulary,
        dY = dlog^T, X = the bf16 embedding table itself): the vocabulary is cut into S chunks, chunk s is ONE problem of a
        grouped launch with its own zeroed fp32 slab [n, H] (9 tiles of 256 x 256 per chunk, S ~ 27: one round of the chip,
        every tile reduced by one workgroup and written once), and the slabs are added in chunk order and rounded to bf16 by
        one small kernel -- the result does not depend on which workgroup finishes first (atomics into one buffer did:
        rounded to bf16 that showed up as 4e-3 of the gradient scale between two schedules of the same step).
        Costs a bf16 transpose of dlog (38 MB), the slabs (S x 2 MB) and the reduction."""
        H = dec.K
        ldT = _rup(n, 64)
        dlogT = torch.empty((V, ldT), dtype=BF16, device=dlog.device)
        L.transpose_bf16(dlog, dlogT, n, V)
        tiles = ((ldT + 255) // 256) * ((H + 255) // 256)
        S = max(1, min(48, 256 // tiles, V // 1024))
        chunk = _rup((V + S - 1) // S, 64)
        S = max(1, V // chunk)                                 # the last chunk takes the remainder (< 2 chunks long)
        slabs = torch.zeros((S, ldT, H), dtype=F32, device=dlog.device)
        ends = [(s + 1) * chunk for s in range(S - 1)] + [V]
        L.gemm_tn_grouped([(dlogT[e0:e1], dec.w[e0:e1], slabs[s], e1 - e0, ldT, H, None)
                           for s, (e0, e1) in enumerate(zip([0] + ends[:-1], ends))], shared=0)
        dhn = torch.empty((ldT, H), dtype=BF16, device=dlog.device)
        L.sum_slabs_bf16(slabs, dhn, ldT * H)
        return dhn[:n]

    def _layernorm(self, x, key, save, drop=L.NO_DROP, want32=True, lazy=False):
        """x: fp32 pre-LayerNorm sum -> (y32 residual stream | None, y16 GEMM operand, mean, rstd).
        lazy=True: the fp32 output is NOT written; a `_LazyLN` stands in for it and the next residual epilogue
        evaluates LayerNorm(x) from (x, mean, rstd, gamma, beta) itself (4 of the 10 bytes per element this
        HBM-bound kernel moved)."""
        gmm, bta, _, _ = self.ln[key]
        M, H = x.shape
        lazy = lazy and self.lazy_ln and drop[1] == 0
        y32 = torch.empty((M, H), dtype=F32, device=x.device) if (want32 and not lazy) else None
        y16 = torch.empty((M, H), dtype=BF16, device=x.device)
        keep = save or lazy
        mean = torch.empty(M, dtype=F32, device=x.device) if keep else None
        rstd = torch.empty(M, dtype=F32, device=x.device) if keep else None
        L.layernorm_fwd(x, gmm, bta, y32, y16, mean, rstd, M, H, drop=drop)
        if lazy:
            y32 = _LazyLN(x, mean, rstd, gmm, bta)
        return y32, y16, mean, rstd

    def _dense32(self, r):
        """fp32 tensor of a residual-stream value (materialises a lazy LayerNorm output)."""
        if not isinstance(r, _LazyLN):
            return r
        M, H = r.x.shape
        y32 = torch.empty((M, H), dtype=F32, device=r.x.device)
        scratch = torch.empty((M, H), dtype=BF16, device=r.x.device)
        L.layernorm_fwd(r.x, r.gamma, r.beta, y32, scratch, None, None, M, H)
        return y32

    # ------------------------------------------------------------------------------------------
    # image-side stream
    # ------------------------------------------------------------------------------------------
    def _dual(self):
        return self.dual_stream and self.arena.flat.is_cuda

    @staticmethod
    def _touch(obj, stream):
        """Tell the caching allocator that `obj` (allocated on another stream) is read on `stream`."""
        if obj is None:
            return
        if isinstance(obj, _LazyLN):
            for t in (obj.x, obj.mean, obj.rstd):
                t.record_stream(stream)
        elif torch.is_tensor(obj) and obj.is_cuda:
            obj.record_stream(stream)

    def _side_stream(self):
        if self._vside is None:
            self._vside = torch.cuda.Stream(device=self.arena.flat.device)
        return self._vside

    @contextmanager
    def _img(self):
        """Launches inside the block go to the image-side stream, in that stream's order (no waits: use
        `_to_img` / `_to_txt` where the two streams exchange data).  No-op when the dual-stream schedule is off."""
        if not self._dual():
            # one stream: the same schedule serialised -- the block's launches still COUNT as image-side ones (their own
            # weight-gradient queue, i.e. the same grouped launches as the two-stream step, and the image side's tile rule),
            # so that `dual_stream = False` measures the production kernels with the chip to themselves
            was, self._on_side = self._on_side, True
            try:
                yield
            finally:
                self._on_side = was
            return
        side = self._side_stream()
        was, self._on_side = self._on_side, True
        main = self._text_stream
        if was or main is None or L.scoped_stream() is not self._text_scope:   # nested, outside an engine entry, or on a third stream
            try:
                with torch.cuda.stream(side), L.stream_scope(side):
                    yield
            finally:
                self._on_side = was
            return
        # hot path (~110 blocks per step): torch.cuda.stream() costs ~15 us per entry (device-index lookups and a
        # current_stream() query); the text stream of this engine entry is known, so switch and switch back directly
        torch.cuda.set_stream(side)
        try:
            with L.stream_scope(side):
                yield
        finally:
            torch.cuda.set_stream(main)
            self._on_side = was

    @contextmanager
    def _img_if(self, flag):
        if flag:
            with self._img():
                yield
        else:
            yield

    def _to_img(self, *reads):
        """Everything enqueued on the main (text) stream so far happens before whatever the image-side stream is
        given next; `reads` are main-stream tensors the image side is about to use."""
        if not self._dual():
            return
        side = self._side_stream()
        ev = torch.cuda.Event()
        ev.record(torch.cuda.current_stream())
        side.wait_event(ev)
        for t in reads:
            self._touch(t, side)

    def _to_txt(self, *reads):
        """The mirror image: the main stream waits for what the image-side stream has been given so far."""
        if not self._dual() or self._vside is None:
            return
        main = torch.cuda.current_stream()
        ev = torch.cuda.Event()
        ev.record(self._vside)
        main.wait_event(ev)
        for t in reads:
            self._touch(t, main)

    def _layernorm_bwd(self, dy, x, mean, rstd, key, dbias=None, drop=L.NO_DROP, out_drop=L.NO_DROP, defer=True, m_dev=None):
        """Row kernel now; the column sums (dgamma, dbeta, dbias) of all the calls of a block are reduced by one grouped
        launch at the end of the block (`_flush_wgrad`), where nothing waits for them.  defer=False: reduce right away."""
        gmm, _, gg, gb = self.ln[key]
        M, H = x.shape
        dx = torch.empty((M, H), dtype=BF16, device=x.device)
        dxd = torch.empty((M, H), dtype=BF16, device=x.device) if drop[1] != 0 else None
        if not defer:
            scratch = (self.part_side if self._on_side else self.part)[H]
            L.layernorm_bwd(dy, x, mean, rstd, gmm, dx, dxd, gg, gb, dbias, scratch, M, H, drop=drop, out_drop=out_drop)
            return dx, (dxd if dxd is not None else dx)
        part = torch.empty(self.part[H].numel(), dtype=F32, device=x.device)      # private until the grouped reduction
        blocks = L.layernorm_bwd_partials(dy, x, mean, rstd, gmm, dx, dxd, part, M, H, drop=drop, out_drop=out_drop, m_dev=m_dev)
        (self._fq_img if self._on_side else self._fq).append((part, blocks, H, [gg, gb, dbias]))
        return dx, (dxd if dxd is not None else dx)

    # ------------------------------------------------------------------------------------------
    # blocks
    # ------------------------------------------------------------------------------------------
    def _attn(self, q, k, v, mask, B, H, Tq, Tk, D, drop, save, qvar=None, kvar=None, tag=None):
        out = torch.empty((q.shape[0], H * D), dtype=BF16, device=q.device)
        lse = torch.empty((B, H, Tq), dtype=F32, device=q.device) if save else None
        words, mq, mb = mask
        L.attn_fwd(q, k, v, out, lse, words, B, H, Tq, Tk, D, 1.0 / math.sqrt(D), mq, mb, drop, qvar=qvar, kvar=kvar)
        if self.attn_sink is not None:          # output_all_attention_masks: the probabilities as tensors (diagnostic, padded layout)
            if qvar is not None or kvar is not None:
                raise RuntimeError("attention probabilities are collected on the padded schedule only")
            probs = torch.empty((B, H, Tq, Tk), dtype=F32, device=q.device)
            L.attn_probs(q, k, probs, words, B, H, Tq, Tk, D, 1.0 / math.sqrt(D), mq, mb, drop)
            self.attn_sink[tag] = probs
        return out, lse

    def _self_block(self, key, x32, x, mask, B, T, heads, pname, p_attn, p_hid, st, var=None):
        """BertLayer / BertImageLayer (models/vilbert_dialog.py:385-483, :514-612).
        (x32, x): fp32 residual stream and its bf16 copy (the GEMM operand)."""
        train, tape = st["train"], st["tape"]
        save = tape is not None
        Hd = x.shape[1]
        D = Hd // heads
        qkv_l, so, ff1, ff2 = (self.lin[key + s] for s in (".qkv", ".so", ".ff1", ".ff2"))
        qkv = self._linear(x, qkv_l)
        q, k, v = qkv[:, :Hd], qkv[:, Hd:2 * Hd], qkv[:, 2 * Hd:]
        d_attn = self._drop(pname + "attn", p_attn, train)
        ctx, lse = self._attn(q, k, v, mask, B, heads, T, T, D, d_attn, save, qvar=var, kvar=var, tag=key)
        d_so = self._drop(pname + "so", p_hid, train)
        pre1 = self._linear(ctx, so, L.EPI_BIAS_DROP_RESID, aux=x32, drop=d_so, out_f32=True)
        x1_32, x1, m1, r1 = self._layernorm(pre1, key + ".ln1", save, lazy=True)
        # training keeps GELU'(u) (not u): the backward epilogue is then a plain multiply
        h, u = self._linear(x1, ff1, L.EPI_BIAS_GELU_DG, want_u=True) if save else (self._linear(x1, ff1, L.EPI_BIAS_GELU), None)
        d_out = self._drop(pname + "out", p_hid, train)
        pre2 = self._linear(h, ff2, L.EPI_BIAS_DROP_RESID, aux=x1_32, drop=d_out, out_f32=True)
        x2_32, x2, m2, r2 = self._layernorm(pre2, key + ".ln2", save, lazy=True)
        md = var[2] if var is not None else None      # device word: valid rows, when the row dimension is a capacity
        if save:
            def bwd(dx2):
                dpre2, dpre2d = self._layernorm_bwd(dx2, pre2, m2, r2, key + ".ln2", dbias=ff2.gb, drop=d_out, m_dev=md)
                du = self._linear_bwd(dpre2d, h, ff2, L.EPI_MUL, aux=u, bias_grad=False, m_dev=md)
                dx1 = self._linear_bwd(du, x1, ff1, L.EPI_ADD, aux=dpre2, m_dev=md)
                dpre1, dpre1d = self._layernorm_bwd(dx1, pre1, m1, r1, key + ".ln1", dbias=so.gb, drop=d_so, m_dev=md)
                dctx = self._linear_bwd(dpre1d, ctx, so, bias_grad=False, m_dev=md)
                dqkv = torch.empty_like(qkv)
                delta = torch.empty_like(lse)
                words, mq, mb = mask
                L.attn_bwd(q, k, v, ctx, dctx, lse, delta, dqkv[:, :Hd], dqkv[:, Hd:2 * Hd], dqkv[:, 2 * Hd:], words,
                           B, heads, T, T, D, 1.0 / math.sqrt(D), mq, mb, d_attn, qvar=var, kvar=var)
                return self._linear_bwd(dqkv, x, qkv_l, L.EPI_ADD, aux=dpre1, m_dev=md)
            tape.append((key, bwd))
        return x2_32, x2

    def _conn_block(self, key, i, xv32, xv, xt32, xt, B, R, T, vmask, comask, st, var=None):
        """BertConnectionLayer (models/vilbert_dialog.py:655-783)."""
        cfg = self.cfg
        train, tape = st["train"], st["tape"]
        save = tape is not None
        pn = f"bert.encoder.c_layer.{i}."
        Hb, nh = cfg.bi_hidden_size, cfg.bi_num_attention_heads
        D = Hb // nh
        lq1, lq2, d1, d2 = (self.lin[key + s] for s in (".qkv1", ".qkv2", ".d1", ".d2"))
        vff1, vff2, tff1, tff2 = (self.lin[key + s] for s in (".vff1", ".vff2", ".tff1", ".tff2"))
        # The two halves run on their own streams (`_img()` = image side, otherwise the text side); the only
        # exchanges are the other side's K/V for the two co-attention directions.
        with self._img():
            qkv1 = self._linear(xv, lq1)      # image side  [B*R, 3Hb]
        qkv2 = self._linear(xt, lq2)          # text side   [B*T, 3Hb]
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
        with self._img():
            ctx_v, lse_v = self._attn(q1, k2, v2, comask, B, nh, R, T, D, da2, save, kvar=var, tag=key + "/2")    # regions attend text (:701-721)
            prev = self._linear(ctx_v, d1, L.EPI_BIAS_DROP_RESID, aux=xv32, drop=db1, out_f32=True)   # BertBiOutput (:744-754, call order :775)
            av32, av, mv1, rv1 = self._layernorm(prev, key + ".lnb1", save, lazy=True)
            if save:
                hv, uv = self._linear(av, vff1, L.EPI_BIAS_GELU_DG, want_u=True)
            else:
                hv, uv = self._linear(av, vff1, L.EPI_BIAS_GELU), None
            prev2 = self._linear(hv, vff2, L.EPI_BIAS_DROP_RESID, aux=av32, drop=dvo, out_f32=True)
            ov32, ov, mv2, rv2 = self._layernorm(prev2, key + ".lnv", save, lazy=True)
        ctx_t, lse_t = self._attn(q2, k1, v1, vmask, B, nh, T, R, D, da1, save, qvar=var, tag=key + "/1")     # text attends regions (:681-698)
        pret = self._linear(ctx_t, d2, L.EPI_BIAS_DROP_RESID, aux=xt32, drop=db2, out_f32=True)
        at32, at, mt1, rt1 = self._layernorm(pret, key + ".lnb2", save, lazy=True)
        if save:
            ht, ut = self._linear(at, tff1, L.EPI_BIAS_GELU_DG, want_u=True)
        else:
            ht, ut = self._linear(at, tff1, L.EPI_BIAS_GELU), None
        pret2 = self._linear(ht, tff2, L.EPI_BIAS_DROP_RESID, aux=at32, drop=dto, out_f32=True)
        ot32, ot, mt2, rt2 = self._layernorm(pret2, key + ".lnt", save, lazy=True)
        md = var[2] if var is not None else None      # device word: valid text rows (capacity-sized text tensors)
        if save:
            def bwd(dov, dot):
                sc = 1.0 / math.sqrt(D)
                # gradient buffers of the two projections: each is written by BOTH attention backward kernels (every
                # slice exactly once), i.e. from both streams -> allocate first and let each stream see the other's
                with self._img():
                    dqkv1 = torch.empty_like(qkv1)
                dqkv2 = torch.empty_like(qkv2)
                self._to_txt(dqkv1)
                self._to_img(dqkv2)
                with self._img():                                   # image half: FFN, bi-output
                    dp, dpd = self._layernorm_bwd(dov, prev2, mv2, rv2, key + ".lnv", dbias=vff2.gb, drop=dvo)
                    duv = self._linear_bwd(dpd, hv, vff2, L.EPI_MUL, aux=uv, bias_grad=False)
                    dav = self._linear_bwd(duv, av, vff1, L.EPI_ADD, aux=dp)
                    dprev, dprevd = self._layernorm_bwd(dav, prev, mv1, rv1, key + ".lnb1", dbias=d1.gb, drop=db1)
                    dctx_v = self._linear_bwd(dprevd, ctx_v, d1, bias_grad=False)
                    delta_v = torch.empty_like(lse_v)
                    w, mq, mb = comask
                    L.attn_bwd(q1, k2, v2, ctx_v, dctx_v, lse_v, delta_v, dqkv1[:, :Hb], dqkv2[:, Hb:2 * Hb], dqkv2[:, 2 * Hb:],
                               w, B, nh, R, T, D, sc, mq, mb, da2, kvar=var)
                dp, dpd = self._layernorm_bwd(dot, pret2, mt2, rt2, key + ".lnt", dbias=tff2.gb, drop=dto, m_dev=md)   # text half
                dut = self._linear_bwd(dpd, ht, tff2, L.EPI_MUL, aux=ut, bias_grad=False, m_dev=md)
                dat = self._linear_bwd(dut, at, tff1, L.EPI_ADD, aux=dp, m_dev=md)
                dpret, dpretd = self._layernorm_bwd(dat, pret, mt1, rt1, key + ".lnb2", dbias=d2.gb, drop=db2, m_dev=md)
                dctx_t = self._linear_bwd(dpretd, ctx_t, d2, bias_grad=False, m_dev=md)
                delta_t = torch.empty_like(lse_t)
                w, mq, mb = vmask
                L.attn_bwd(q2, k1, v1, ctx_t, dctx_t, lse_t, delta_t, dqkv2[:, :Hb], dqkv1[:, Hb:2 * Hb], dqkv1[:, 2 * Hb:],
                           w, B, nh, T, R, D, sc, mq, mb, da1, qvar=var)
                self._to_img()                                      # dK1/dV1 written by the text side
                self._to_txt()                                      # dK2/dV2 written by the image side
                with self._img():
                    dxv = self._linear_bwd(dqkv1, xv, lq1, L.EPI_ADD, aux=dprev)
                dxt = self._linear_bwd(dqkv2, xt, lq2, L.EPI_ADD, aux=dpret, m_dev=md)
                return dxv, dxt
            tape.append((key, bwd))
        return ov32, ov, ot32, ot

    # ------------------------------------------------------------------------------------------
    # inputs
    # ------------------------------------------------------------------------------------------
    @staticmethod
    def _i32(x, device):
        return x.to(device=device, dtype=torch.int32, non_blocking=True).contiguous()

    def stage_host_inputs(self, inp: dict, pack=None):
        """CPU tensors among the step's inputs -> device tensors through the engine's pinned staging ring and copy stream
        (inputs.HostStager; dense masks arrive bit-packed), in place.  Called by the model's forward entry points, so that the
        reference's calling convention -- host tensors handed to forward(), train.py:113-161 -- runs at the speed of resident
        inputs.  `host_staging = False` restores the plain `.to(device)` path."""
        if not self.host_staging or not self.arena.flat.is_cuda:
            return False
        if not any(torch.is_tensor(inp.get(k)) and not inp[k].is_cuda for k in _STAGED_KEYS):
            return False
        if self._stager is None:
            self._stager = HostStager(self.arena.flat.device)
        staged = self._stager.stage(inp, _STAGED_KEYS, pack=pack)
        if staged and self._dual():
            side = self._side_stream()
            for k in ("image_feat", "image_loc", "image_attention_mask", "image_target", "image_label"):
                v = inp.get(k)
                self._touch(v.words if isinstance(v, PackedMask) else v, side)
        return staged

    def _pack_mask(self, m, device, rows_expected):
        """-> (words, q_stride, b_stride).  m: [B, Tk] (key padding) or [B, Tq, Tk]."""
        if isinstance(m, PackedMask):              # packed on the host side of the copy (inputs.HostStager)
            words = m.words
            nw = words.shape[-1]
            return (words, 0, nw) if m.dim() == 2 else (words, nw, m.shape[1] * nw)
        if m.dtype not in (torch.bool, torch.uint8, torch.int32, torch.int64, torch.float32):
            m = m.float()
        if not m.is_cuda and m.dtype == torch.int64:
            m = m.to(torch.uint8)          # 8x less PCIe traffic for the reference's int64 masks
        m = m.to(device, non_blocking=True)
        words = L.mask_pack(m)
        nw = words.shape[-1]
        self._dev_masks.append(m)
        if m.dim() == 2:
            return (words, 0, nw)
        return (words, nw, m.shape[1] * nw)

    # Unpadded schedule: a text row is INERT when no valid row ever attends it (its column of the text mask and of
    # the co-attention mask is empty), it attends nothing, carries no label / weight and is not the pooled first
    # token: whatever it computes never reaches a loss or a valid row, and its gradient is exactly zero (the
    # reference still spends ~45 % of its token FLOPs there: rows past each dialog's length,
    # models/vilbert_dialog.py:1418 / utils/data_utils.py:207).  The engine runs the text stream on rows
    # [0, len_b) of every sequence only; len_b comes from `unimm_plan_lengths` (see `_forward`).

    # ------------------------------------------------------------------------------------------
    # forward
    # ------------------------------------------------------------------------------------------
    def _on_text_stream(self, fn, *args):
        """Run one engine entry on the text stream.  Normally that is the caller's current stream.  With
        `text_priority` (and two streams) it is an internal HIGH-priority stream bracketed by waits in both
        directions, so that the text kernels - the critical path - are placed before the image side's."""
        if not self.arena.flat.is_cuda:
            return fn(*args)
        caller = torch.cuda.current_stream()
        if not (self.text_priority and self._dual()):
            self._text_stream = caller
            try:
                with L.stream_scope(caller) as sc:
                    self._text_scope = sc.ptr
                    return fn(*args)
            finally:
                self._text_stream = self._text_scope = None
        if self._tstream is None:
            self._tstream = torch.cuda.Stream(device=self.arena.flat.device, priority=-1)
        hp = self._tstream
        hp.wait_stream(caller)
        self._text_stream = hp
        try:
            with torch.cuda.stream(hp), L.stream_scope(hp) as sc:
                self._text_scope = sc.ptr
                return fn(*args)
        finally:
            self._text_stream = self._text_scope = None
            caller.wait_stream(hp)

    def enable_graphs(self, on=True, **kw):
        """Run the training step as two replayed hipGraphs per row-count bucket (unimm_amd/graphs.py) instead of ~650
        Python -> C launches: what the small per-GPU batches of the 8-GPU split need (host-bound otherwise)."""
        from .graphs import StepGraphs
        self.graphs = StepGraphs(self, **kw) if on else None
        return self.graphs

    def count_rows(self, inp: dict, lm_rows="labelled"):
        """The header of the step plan as host values, computed ahead of the step: what `_forward` reads at its one host
        sync.  Same kernels on the same inputs (masks bit-packed, unimm_plan_lengths)."""
        dev = self.arena.device
        ids, feat = inp["input_ids"], inp["image_feat"]
        B, T = ids.shape
        R = feat.shape[1]
        am, im, cm = inp.get("attention_mask"), inp.get("image_attention_mask"), inp.get("co_attention_mask")
        if am is None or isinstance(am, DialogMaskSpec) or cm is None:
            raise ValueError("count_rows needs dense attention_mask / co_attention_mask tensors")
        self._dev_masks = []
        tmask = self._pack_mask(am, dev, T)
        comask = self._pack_mask(cm, dev, R)
        self._dev_masks = []
        labels, weights = inp.get("masked_lm_labels"), inp.get("lm_weight")
        lab32 = self._i32(labels.reshape(B, T), dev) if labels is not None else None
        w32 = self._i32(weights.reshape(B, T), dev) if (weights is not None and labels is not None) else None
        il = inp.get("image_label")
        il32 = self._i32(il.reshape(B, R), dev) if (il is not None and inp.get("image_target") is not None) else None
        nw_in = inp.get("nsp_weight")
        nw_dev = nw_in.reshape(-1)[:2].to(F32).contiguous() if (torch.is_tensor(nw_in) and nw_in.is_cuda) else None
        header = L.plan_lengths(tmask, comask, R, lab32, w32, nw_dev, B, T, image_label=il32)
        return header.tolist()

    def forward(self, inp: dict, train: bool, save: bool, lm_rows: str, want_pred_v: bool):
        return self._on_text_stream(self._forward, inp, train, save, lm_rows, want_pred_v)

    def forward_with_attention(self, inp, train, lm_rows, want_pred_v):
        """Inference forward that also returns the attention probabilities of every layer as the reference's encoder
        collects them under output_all_attention_masks (models/vilbert_dialog.py:834-937): (text layers, image layers,
        connection layers as (text-attends-regions, regions-attend-text)), each fp32 [B, heads, Tq, Tk].  Runs the padded,
        one-stream schedule (the probabilities of padding rows are part of what the reference returns)."""
        was = (self.unpad, self.dual_stream, self.attn_sink)
        self.unpad, self.dual_stream, self.attn_sink = False, False, {}
        try:
            out = self.forward(inp, train=train, save=False, lm_rows=lm_rows, want_pred_v=want_pred_v)
            sink = self.attn_sink
        finally:
            self.unpad, self.dual_stream, self.attn_sink = was

        def order(prefix):
            ks = [k for k in sink if k.startswith(prefix) and "/" not in k]
            return sorted(ks, key=lambda k: int("".join(ch for ch in k if ch.isdigit())))
        att_t = [sink[k] for k in order("t")]
        att_v = [sink[k] for k in order("v")]
        cs = sorted({k.split("/")[0] for k in sink if "/" in k}, key=lambda k: int("".join(ch for ch in k if ch.isdigit())))
        att_c = [(sink[k + "/1"], sink[k + "/2"]) for k in cs]
        return out, (att_t, att_v, att_c)

    def backward(self, out, g_lm, g_img, g_nsp, g_nsp_scores=None):
        return self._on_text_stream(self._backward, out, g_lm, g_img, g_nsp, g_nsp_scores)

    def losses(self, out, inp):
        return self._on_text_stream(self._losses, out, inp)

    def _prep_masks(self, inp, B, T, R, dev):
        """Packed attention masks of the step: (text mask, image key mask, co-attention mask), each (words, q stride, b stride)."""
        ids, feat = inp["input_ids"], inp["image_feat"]
        # ---- masks (models/vilbert_dialog.py:1374-1431) ------------------------------------------
        am = inp.get("attention_mask")
        spec = am if isinstance(am, DialogMaskSpec) else None
        if spec is not None:
            if len(spec) != B or int(spec.length.max()) > T:
                raise ValueError(f"DialogMaskSpec for {len(spec)} sequences / max length {int(spec.length.max())} "
                                 f"does not fit input_ids of shape {tuple(ids.shape)}")
            if inp.get("co_attention_mask") is not None:
                raise ValueError("co_attention_mask must be None when attention_mask is a DialogMaskSpec (it is derived)")
            am = torch.ones((B, T), dtype=torch.uint8, device=dev)      # placeholder, never packed
        if am is None:
            am = torch.ones((B, T), dtype=torch.uint8, device=dev)
        if am.dim() not in (2, 3):
            raise ValueError(f"Wrong shape for txt input_ids (shape {tuple(ids.shape)}) or attention_mask (shape {tuple(am.shape)})")
        im = inp.get("image_attention_mask")
        if im is None:
            im = torch.ones((B, R), dtype=torch.uint8, device=dev)
        if im.dim() not in (2, 3):
            raise ValueError(f"Wrong shape for img input_ids (shape {tuple(feat.shape)}) or attention_mask (shape {tuple(im.shape)})")
        self._dev_masks = []
        if spec is not None:
            # masks synthesised on the device from (mode, L, n)
            tw, cw = L.mask_synth(*spec.to_device(dev), T)
            nw = tw.shape[-1]
            tmask, comask = (tw, nw, T * nw), (cw, 0, nw)
            vmask = self._pack_mask(im, dev, R)
        else:
            cm = inp.get("co_attention_mask")
            if cm is None:
                cm = torch.ones((B, R, T), dtype=torch.uint8, device=dev)
            assert cm.dim() == 3
            tmask = self._pack_mask(am, dev, T)
            vmask = self._pack_mask(im, dev, R)
            comask = self._pack_mask(cm, dev, R)
        self._dev_masks = []
        return tmask, vmask, comask

    def _prep_plan(self, inp, B, T, R, tmask, comask, lm_rows, dev):
        """Input conversions + the plan of the step (one device->host copy): valid prefix lengths of the unpadded schedule,
        the rows the MLM head decodes, loss denominators as device words.  Returns a dict."""
        ids = inp["input_ids"]
        # ---- plan of the step: everything the host has to know, in ONE device->host copy ------------------
        # (valid prefix lengths for the unpadded schedule, the number of rows the MLM head decodes, the NSP class
        # weights when they live on the device); row maps and the decoded rows' index lists are then built on the
        # device.  See csrc/rowops.hip: plan_lengths / plan_build.
        tt = inp.get("token_type_ids")                       # (input conversions go in front of the sync as well)
        ids32 = self._i32(ids.reshape(-1), dev)
        typ32 = self._i32(tt.reshape(-1), dev) if tt is not None else torch.zeros(B * T, dtype=torch.int32, device=dev)
        pos = inp.get("position_ids")
        pos32 = self._i32(pos.reshape(-1), dev) if pos is not None else \
            torch.arange(T, dtype=torch.int32, device=dev).repeat(B)
        labels, weights = inp.get("masked_lm_labels"), inp.get("lm_weight")
        want_sel = lm_rows == "labelled" and labels is not None
        lab32 = self._i32(labels.reshape(B, T), dev) if labels is not None else None
        w32 = self._i32(weights.reshape(B, T), dev) if (weights is not None and labels is not None) else None
        nw_in = inp.get("nsp_weight")
        nw_dev = nw_in.reshape(-1)[:2].to(F32).contiguous() if (torch.is_tensor(nw_in) and nw_in.is_cuda) else None
        il = inp.get("image_label")
        il32 = self._i32(il.reshape(B, R), dev) if (il is not None and inp.get("image_target") is not None) else None
        plan, sel, n_img = None, None, None
        dyn = None
        if self.unpad or want_sel or nw_dev is not None or il32 is not None:
            header = L.plan_lengths(tmask, comask, R, lab32, w32, nw_dev, B, T, image_label=il32)
            # the step's one host sync (the graph executor has read the same header in its pre-pass and injects the values:
            # a captured launch sequence cannot synchronise)
            hh = self._inject_header if self._inject_header is not None else header.tolist()
            n_img = sum(hh[2 * B + 2:3 * B + 2]) if il32 is not None else None
            lens_h, n_lm = hh[:B], (sum(hh[B:2 * B]) if want_sel else 0)
            Mv = sum(lens_h)
            # capacities: what the launches are sized for (= the real counts unless the graph executor set buckets)
            Mcap = min(_rup(Mv, self.row_bucket), B * T)
            ncap = _rup(n_lm, self.lm_bucket) if n_lm > 0 else 0
            # (with row capacities -- the graph executor -- the layout must not depend on whether the bucket happens to reach
            #  B * T: packed even then, with the identity as row map, so that every replay of a signature and the eager step of
            #  the same batch number their rows, and therefore draw their dropout masks, alike)
            unpadded = self.unpad and (Mv < B * T or self.row_bucket > 1)
            # the device words of THIS forward (never shared between forwards: a backward reads the counts of its own step
            # even when other forwards -- an eval pass, a second micro-batch -- ran in between); under a graph capture they come
            # from the graph's pool, i.e. they are static per captured entry
            di, df = torch.zeros(8, dtype=torch.int32, device=dev), torch.zeros(8, dtype=F32, device=dev)
            built = L.plan_build(header, lab32 if want_sel else None, w32 if want_sel else None, B, T, Mcap if unpadded else Mv,
                                 ncap, want_rows=unpadded, dims=(di, df))
            dyn = dict(m=di[0:1], n_lm=di[1:2], n_img=di[2:3], inv_lm=df[0:1], inv_img=df[1:2])
            if unpadded:
                plan = dict(Mv=Mcap, lens_h=lens_h, rows=built["rows"], inv=built["inv"],
                            var=(built["off"], built["lens"], dyn["m"], built["order"] if self.attn_longest_first else None))
            if want_sel:
                sel = dict(n=ncap, pos=built["lm_pos"], idx=built["lm_idx"] if unpadded else built["lm_pos"],
                           label=built["lm_label"], weight=built["lm_weight"])
            if nw_dev is not None:
                import struct
                st_nspw = struct.unpack("<2f", struct.pack("<2i", hh[2 * B], hh[2 * B + 1]))
            else:
                st_nspw = None
        else:
            st_nspw = None
        self.last_plan = plan
        var = plan["var"] if plan is not None else None
        Mt = plan["Mv"] if plan is not None else B * T      # text rows actually computed
        self._step_rows = Mt

        return dict(ids32=ids32, typ32=typ32, pos32=pos32, labels=labels, lab32=lab32, w32=w32, il32=il32, plan=plan, sel=sel,
                    n_img=n_img, dyn=dyn, st_nspw=st_nspw, var=var, Mt=Mt)

    def _forward(self, inp: dict, train: bool, save: bool, lm_rows: str, want_pred_v: bool):
        """Runs the trunk + heads.  Returns a dict of outputs and (when save) the tape for backward.
        lm_rows: 'labelled' (decode only rows that carry a label / weight), 'all', or 'none'."""
        cfg = self.cfg
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
        # ---- image embedding first: it does not depend on the plan, so the image stream already has work while the
        # host waits for the header below (models/vilbert_dialog.py:360-383)
        A = self.arena
        F = cfg.v_feature_size
        self._to_img()                        # masks are packed (and the previous step is behind us): the image side may start
        with self._img():                     # image embedding: beside the first text layers
            featd = feat.to(dev, dtype=F32, non_blocking=True)
            locd = inp["image_loc"].to(dev, dtype=F32, non_blocking=True)
            if img_idx is not None:             # one entry per image on the wire, expanded on the device (train.py:413-432)
                featd, locd = featd.index_select(0, img_idx), locd.index_select(0, img_idx)
            featd, locd = featd.contiguous().view(B * R, F), locd.contiguous().view(B * R, 5)
            packed = torch.empty((B * R, self.vemb_k), dtype=BF16, device=dev)
            L.pack_image(featd, locd, packed, B * R, F, self.vemb_k)
            prev = torch.empty((B * R, Hv), dtype=F32, device=dev)
            L.gemm_nt(packed, self.vemb_w, prev, bias=self.vemb_b, M=B * R, N=Hv, K=self.vemb_k)
            d_embv = self._drop("emb_v", cfg.hidden_dropout_prob, train)
            xv32, xv, mv, rv = self._layernorm(prev, "emb_v", save, drop=d_embv)
            if save:
                v = "bert.v_embeddings."

                def bwd_embv(dxv):
                    dbias = A.grad(v + "image_embeddings.bias")
                    before = dbias.clone()
                    dpre, _ = self._layernorm_bwd(dxv, prev, mv, rv, "emb_v", dbias=dbias, out_drop=d_embv, defer=False)
                    A.grad(v + "image_location_embeddings.bias").add_(dbias - before)
                    self._wgrad(dpre, packed, A.grad(v + "image_embeddings.weight"), B * R, Hv, F, sole=True)
                    self._wgrad(dpre, packed[:, F:], A.grad(v + "image_location_embeddings.weight"), B * R, Hv, 5, sole=True)

        pl = self._prep_plan(inp, B, T, R, tmask, comask, lm_rows, dev)
        ids32, typ32, pos32, labels = pl["ids32"], pl["typ32"], pl["pos32"], pl["labels"]
        il32, plan, sel, n_img, dyn, st_nspw, var, Mt = (pl[k] for k in ("il32", "plan", "sel", "n_img", "dyn", "st_nspw", "var", "Mt"))
        # ---- embeddings --------------------------------------------------------------------------
        erows = plan["rows"] if plan is not None else None      # packed row -> padded row (the kernels gather through it)
        emd = plan["var"][2] if plan is not None else None
        gmm, bta, ggm, gbt = self.ln["emb_t"]
        d_embt = self._drop("emb_t", cfg.hidden_dropout_prob, train)
        xt = torch.empty((Mt, H), dtype=BF16, device=dev)
        xt32 = torch.empty((Mt, H), dtype=F32, device=dev)
        tabs = (self.tab["word"], self.tab["pos"], self.tab["type"], self.tab["ext"])
        L.embed_fwd(ids32, pos32, typ32, *tabs, gmm, bta, xt32, xt, Mt, H, cfg.type_vocab_size, drop=d_embt, m_dev=emd, rows=erows)
        A = self.arena
        e = "bert.embeddings."
        if save:
            def bwd_embt(dxt):
                L.embed_bwd(ids32, pos32, typ32, *tabs, gmm, bta, dxt, A.grad(e + "word_embeddings.weight"),
                            A.grad(e + "position_embeddings.weight"), A.grad(e + "token_type_embeddings.weight"),
                            A.grad(e + "token_type_embeddings_extension.weight"), ggm, gbt, self.part[H], Mt, H,
                            cfg.type_vocab_size, drop=d_embt, m_dev=emd, rows=erows)

        # ---- encoder (schedule of models/vilbert_dialog.py:842-929) ------------------------------
        # Two streams: the image stream (embedding, image layers, the image half of every connection layer) and the
        # text stream (= the caller's current stream).  They exchange data only inside the connection layers (the
        # co-attention needs the other side's K/V) and meet again before the heads.  Within a segment the image
        # layers are enqueued first so that both queues are fed; the tape keeps each stream's order for backward.
        sched = PM.encoder_schedule(cfg)
        st_frozen = dict(train=train, tape=None)
        pos = 0
        while pos < len(sched):
            seg = []
            while pos < len(sched) and sched[pos][0] != "c":
                seg.append(sched[pos])
                pos += 1
            # fixed_t_layer / fixed_v_layer (models/vilbert_dialog.py:850-869): the first layers of a stream run under no_grad in
            # the reference -- here: nothing of them is saved and no backward step is taped, so no gradient reaches them or the
            # stream's embeddings.  with_coattention = False (:901): the connection layers are skipped.
            for kind, i in seg:
                if kind == "v":
                    fz = i < cfg.fixed_v_layer
                    with self._img():
                        xv32, xv = self._self_block(f"v{i}", xv32, xv, vmask, B, R, cfg.v_num_attention_heads,
                                                    f"bert.encoder.v_layer.{i}.", cfg.v_attention_probs_dropout_prob,
                                                    cfg.v_hidden_dropout_prob, st_frozen if fz else st)
                    if save and not fz:
                        tape[-1] = ("v", tape[-1][0], tape[-1][1])
            for kind, i in seg:
                if kind == "t":
                    fz = i < cfg.fixed_t_layer
                    xt32, xt = self._self_block(f"t{i}", xt32, xt, tmask, B, T, cfg.num_attention_heads, f"bert.encoder.layer.{i}.",
                                                cfg.attention_probs_dropout_prob, cfg.hidden_dropout_prob, st_frozen if fz else st, var=var)
                    if save and not fz:
                        tape[-1] = ("t", tape[-1][0], tape[-1][1])
            if pos < len(sched):
                i = sched[pos][1]
                pos += 1
                if cfg.with_coattention:
                    with self._conn_tag():
                        xv32, xv, xt32, xt = self._conn_block(f"c{i}", i, xv32, xv, xt32, xt, B, R, T, vmask, comask, st, var=var)
                    if save:
                        tape[-1] = ("c", tape[-1][0], tape[-1][1])
        # ---- image head (:1001-1005, :1085-1088; its masked KL :1569-1574 is in _losses): on the image stream, beside the
        # text side's last layers and the MLM head
        img = None
        pred_v_out = None
        if want_pred_v or inp.get("image_target") is not None:
            with self._img_if(self.image_head_side):
                itr, idec = self.lin["imgtr"], self.lin["imgdec"]
                C = cfg.v_target_size
                if save:
                    tv, uvh = self._linear(xv, itr, L.EPI_BIAS_GELU, want_u=True, out_f32=True)
                else:
                    tv, uvh = self._linear(xv, itr, L.EPI_BIAS_GELU, out_f32=True), None
                _, hvn, mh, rh = self._layernorm(tv, "imgtr", save, want32=False)
                pred_v = self._linear(hvn, idec, out_f32=True, ldo=_rup(C, 4))
            pred_v_out = pred_v.view(B, R, -1)[:, :, :C]
            img = dict(tv=tv, u=uvh, hn=hvn, mean=mh, rstd=rh, pred=pred_v)
        self._to_txt(xv32, xv, img["pred"] if img is not None else None)                # the heads read both streams
        seq_t, seq_v = xt, xv

        xt32, xv32 = self._dense32(xt32), self._dense32(xv32)      # the final residual stream is an output
        out = dict(seq_out_t=seq_t, seq_out_v=seq_v, seq32_t=xt32, seq32_v=xv32, B=B, T=T, R=R, plan=plan, Mt=Mt,
                   nsp_weight_host=st_nspw, n_img=n_img, img_label32=il32, dyn=dyn, img=img)
        if pred_v_out is not None:
            out["pred_v"] = pred_v_out
        # ---- poolers + NSP (models/vilbert_dialog.py:946-967, 1064-1070) -------------------------
        cls_idx_t = var[0] if var is not None else torch.arange(0, B * T, T, dtype=torch.int32, device=dev)
        cls_idx_v = torch.arange(0, B * R, R, dtype=torch.int32, device=dev)
        # Everything above the encoder's last LayerNorm runs in fp32 from the fp32 residual stream and the fp32 master
        # weights (unimm_linear_f32): 0.4 GFLOP, but with bf16 operands ReLU units of the B pooled rows switched with the
        # last bit of the forward and the pooler gradients were 8-16 % off the reference's (round 2).
        cls_t = torch.empty((B, H), dtype=F32, device=dev)         # first-token rows (:949, :964)
        cls_v = torch.empty((B, Hv), dtype=F32, device=dev)
        L.gather_rows(xt32.view(BF16), cls_idx_t, cls_t.view(BF16), B, 2 * H)      # fp32 rows moved as 2 H 16-bit elements
        L.gather_rows(xv32.view(BF16), cls_idx_v, cls_v.view(BF16), B, 2 * Hv)
        pooled_t = self._linear32(cls_t, "tpool", relu=True)
        pooled_v = self._linear32(cls_v, "vpool", relu=True)
        d_fuse = self._drop("fuse", 0.1, train)
        fused = torch.empty_like(pooled_t)
        L.mul_dropout(pooled_t, pooled_v, fused, fused.numel(), d_fuse, fusion_sum=cfg.fusion_method == "sum")
        nsp = torch.zeros((B, 4), dtype=F32, device=dev)
        self._linear32(fused, "nsp", out=nsp)
        out["nsp"] = nsp[:, :2]

        # ---- MLM head: transform + tied decoder on the selected rows (:982-986, :1023-1026) -------
        V = cfg.vocab_size
        Vp = _rup(V, 64)
        lmtr, dec = self.lin["lmtr"], self.lin["dec"]
        lm = None
        if lm_rows == "labelled" and labels is not None:
            n = sel["n"]                                          # rows chosen by the plan kernels at the start of forward (a capacity)
            if n > 0:
                xs = torch.empty((n, H), dtype=BF16, device=dev)
                L.gather_rows(seq_t, sel["idx"], xs, n, H, n_dev=dyn["n_lm"])
                lm = self._lm_head(xs, n, sel["label"], sel["weight"], save, n_dev=dyn["n_lm"])
                lm.update(idx=sel["idx"], pos_idx=sel["pos"], n=n, n_dev=dyn["n_lm"], inv_dev=dyn["inv_lm"])
            out["lm"] = lm
        elif lm_rows == "all":
            out["pred_t"] = self.decode_rows(self.padded(out, seq_t), B * T).view(B, T, Vp)[:, :, :V]

        if save:
            out["bwd"] = dict(tape=tape, embt=bwd_embt, embv=bwd_embv, pooled_t=pooled_t, pooled_v=pooled_v, fused=fused,
                              d_fuse=d_fuse, nsp_pad=nsp, cls_t=cls_t, cls_v=cls_v, cls_idx_t=cls_idx_t,
                              cls_idx_v=cls_idx_v)
        return out

    @staticmethod
    def padded(out, x):
        """[rows computed, H] -> the reference's [B*T, H] layout (rows that were never computed are zero)."""
        plan = out.get("plan")
        if plan is None:
            return x
        full = torch.zeros((out["B"] * out["T"], x.shape[1]), dtype=x.dtype, device=x.device)
        if "rows32" not in plan:
            plan["rows32"] = plan["rows"].to(torch.int32)
        w = 2 if x.dtype == F32 else 1                  # fp32 rows move as 2 H 16-bit elements
        # the row dimension of x is a capacity: only the step's real rows are scattered (device-side count)
        L.gather_rows(x.view(BF16) if w == 2 else x, plan["rows32"], full.view(BF16) if w == 2 else full, x.shape[0],
                      w * x.shape[1], scatter=True, n_dev=plan["var"][2])
        return full

    def _lm_head(self, xs, n, lab_sel, w_sel, save, n_dev=None):
        cfg = self.cfg
        V = cfg.vocab_size
        Vp = _rup(V, 64)
        lmtr, dec = self.lin["lmtr"], self.lin["dec"]
        if save:
            t1, u = self._linear(xs, lmtr, L.EPI_BIAS_GELU, want_u=True, out_f32=True)
        else:
            t1, u = self._linear(xs, lmtr, L.EPI_BIAS_GELU, out_f32=True), None
        _, hn, mean, rstd = self._layernorm(t1, "lmtr", save, want32=False)
        logits = self._linear(hn, dec, out_f32=True, ldo=Vp)
        rowloss, rownll, lse = (torch.empty(n, dtype=F32, device=xs.device) for _ in range(3))
        L.lm_loss_fwd(logits, lab_sel, w_sel, rowloss, rownll, lse, n, V, n_dev=n_dev)
        return dict(xs=xs, t1=t1, u=u, hn=hn, mean=mean, rstd=rstd, logits=logits, rowloss=rowloss, rownll=rownll,
                    lse=lse, labels=lab_sel, weights=w_sel)

    def decode_rows(self, x, n):
        """MLM transform + decoder for n rows of x, fp32 logits [n, Vpad] (no loss, nothing saved)."""
        t1 = self._linear(x, self.lin["lmtr"], L.EPI_BIAS_GELU, M=n, out_f32=True)
        _, hn, _, _ = self._layernorm(t1, "lmtr", False, want32=False)
        return self._linear(hn, self.lin["dec"], out_f32=True, ldo=_rup(self.cfg.vocab_size, 64))

    # ------------------------------------------------------------------------------------------
    # losses + backward
    # ------------------------------------------------------------------------------------------
    def _losses(self, out, inp):
        """Three shape-[1] fp32 losses from the forward state (models/vilbert_dialog.py:1559-1621)."""
        cfg = self.cfg
        dev = self.arena.device
        B, R = out["B"], out["R"]
        res = {}
        lm = out.get("lm")
        lm_loss = torch.empty(1, dtype=F32, device=dev)
        if lm is None:
            lm_loss.fill_(float("nan"))        # 0 / 0 in the reference when nothing is labelled
        else:
            L.reduce_sum(lm["rowloss"], lm["n"], lm_loss, 1.0 / lm["n"], n_dev=lm.get("n_dev"), scale_dev=lm.get("inv_dev"))
        res["lm_loss"] = lm_loss
        # image KL
        img = out["img"]
        C = cfg.v_target_size
        label = inp["image_label"]
        n_img = out["n_img"] if out.get("n_img") is not None else int((label == 1).sum())     # counted by the step plan
        tgt = inp["image_target"].to(dev, dtype=F32, non_blocking=True)
        if inp.get("image_index") is not None and tgt.shape[0] != B:
            tgt = tgt.index_select(0, inp["image_index"].to(dev, dtype=torch.int64).reshape(-1))
        tgt = tgt.contiguous().view(B * R, C)
        lab32 = out["img_label32"].reshape(-1) if out.get("img_label32") is not None else self._i32(label.reshape(-1), dev)
        rl, lse = torch.empty(B * R, dtype=F32, device=dev), torch.empty(B * R, dtype=F32, device=dev)
        img_loss = torch.empty(1, dtype=F32, device=dev)
        if cfg.predict_feature:               # MSE on the labelled regions / max(#selected elements, 1) (:1562-1566)
            L.mse_loss_fwd(img["pred"], tgt, lab32, rl, B * R, C)
            inv_img, inv_img_dev = 1.0 / max(n_img, 1), None
        else:
            L.kl_loss_fwd(img["pred"], tgt, lab32, rl, lse, B * R, C)
            inv_img = 1.0 / n_img if n_img > 0 else float("inf")
            # (the plan kernels counted the regions: the divisor is then read from the device word, not from a launch argument)
            inv_img_dev = out["dyn"]["inv_img"] if (out.get("dyn") is not None and out.get("n_img") is not None) else None
        L.reduce_sum(rl, B * R, img_loss, inv_img, scale_dev=inv_img_dev)
        img.update(target=tgt, label=lab32, lse=lse, inv=inv_img, inv_dev=inv_img_dev)
        res["img_loss"] = img_loss
        # NSP
        nw = inp.get("nsp_weight")
        if nw is None:
            w0, w1 = 1.0, 1.0
        else:
            w = out["nsp_weight_host"] if out.get("nsp_weight_host") is not None else \
                [float(x) for x in nw.reshape(-1, 2)[0].tolist()]          # host tensor: no device round trip
            w0, w1 = 1.0, w[1] / w[0]
        nlab = self._i32(inp["next_sentence_label"].reshape(-1), dev)
        nsp_loss = torch.empty(1, dtype=F32, device=dev)
        L.nsp_loss_fwd(out["bwd"]["nsp_pad"] if "bwd" in out else out["nsp"], nlab, w0, w1, nsp_loss, B)
        out["nsp_state"] = (nlab, w0, w1)
        res["nsp_loss"] = nsp_loss
        return res

    def _backward(self, out, g_lm, g_img, g_nsp, g_nsp_scores=None):
        """Accumulates every parameter gradient into the arena (+=)."""
        cfg = self.cfg
        dev = self.arena.device
        bw = out["bwd"]
        B, T, R = out["B"], out["T"], out["R"]
        H, Hv = cfg.hidden_size, cfg.v_hidden_size
        seq_t, seq_v = out["seq_out_t"], out["seq_out_v"]
        self.arena.attach_grads()
        self._bwd_fresh = bool(self.arena.fresh)   # gradients known to be zero: sole contributors may write instead of add (_wgrad)
        self._reported = set()                     # gradient buckets reported done in this pass (_bucket_done)
        if self._bwd_fresh and self.debug_fresh and bool(self.arena.grad_flat.any()):
            raise RuntimeError("the gradient arena is marked fresh (zeroed since the last backward) but holds non-zero values: "
                               "something wrote into .grad / grad_flat between zero_grad() and backward() without clearing "
                               "`model.engine.arena.fresh` (INTEGRATION.md, 'Gradient arena')")
        self._step_rows = out["Mt"]              # the tile rule follows THIS step's rows (another forward may have run since)

        def gvec(g):
            return torch.zeros(1, dtype=F32, device=dev) if g is None else g.detach().to(F32).reshape(1).contiguous()

        dseq_t = torch.zeros((out["Mt"], H), dtype=BF16, device=dev)
        # ---- image head: on the image stream, beside the MLM head's backward -------------------------
        img = out["img"]
        C = cfg.v_target_size
        itr, idec = self.lin["imgtr"], self.lin["imgdec"]
        Cp = idec.wt.shape[1]
        gimg = gvec(g_img)
        self._to_img(gimg, img["target"], img["lse"], img["label"])
        with self._img_if(self.image_head_side):
            dpred = torch.empty((B * R, Cp), dtype=BF16, device=dev)
            if cfg.predict_feature:
                L.mse_loss_bwd(img["pred"], img["target"], img["label"], gimg, img["inv"], dpred, B * R, C)
            else:
                L.kl_loss_bwd(img["pred"], img["target"], img["label"], img["lse"], gimg, img["inv"], dpred, B * R, C,
                              inv_dev=img.get("inv_dev"))
            dhn_v = self._linear_bwd(dpred, img["hn"], idec, M=B * R, N=C)
            dtv, _ = self._layernorm_bwd(dhn_v, img["tv"], img["mean"], img["rstd"], "imgtr")
            duv = torch.empty_like(dtv)
            L.gelu_bwd(dtv, img["u"], duv, duv.numel())
            dseq_v = self._linear_bwd(duv, seq_v, itr)
        # ---- MLM head ---------------------------------------------------------------------------
        lm = out.get("lm")
        if lm is not None:
            n, V = lm["n"], cfg.vocab_size
            Vp = _rup(V, 64)
            lmtr, dec = self.lin["lmtr"], self.lin["dec"]
            dlog = torch.empty((n, Vp), dtype=BF16, device=dev)
            nd = lm.get("n_dev")                                              # n is a capacity: the real count lives on the device
            L.lm_loss_bwd(lm["logits"], lm["labels"], lm["weights"], lm["lse"], gvec(g_lm), 1.0 / n, dlog, n, V, n_dev=nd,
                          inv_dev=lm.get("inv_dev"))
            # dE += dlog^T hn ; dbias ; dhn = dlog @ E
            if n <= self.skinny_dx_rows:
                self._linear_bwd(dlog, lm["hn"], dec, M=n, N=V, m_dev=nd, need_dx=False)
                dhn = self._decoder_dx(dlog, dec, n, V)
            else:
                dhn = self._linear_bwd(dlog, lm["hn"], dec, M=n, N=V, m_dev=nd)
            dt1, _ = self._layernorm_bwd(dhn, lm["t1"], lm["mean"], lm["rstd"], "lmtr", m_dev=nd)
            du = torch.empty_like(dt1)
            L.gelu_bwd(dt1, lm["u"], du, du.numel())
            dxs = self._linear_bwd(du, lm["xs"], lmtr, m_dev=nd)
            L.gather_rows(dxs, lm["idx"], dseq_t, n, H, scatter=True, n_dev=nd)
        # ---- NSP + poolers ------------------------------------------------------------------------
        nlab, w0, w1 = out["nsp_state"]
        dnsp = torch.empty((B, 2), dtype=F32, device=dev)
        extra = None                      # gradient arriving through the returned NSP scores (dense fine-tune ranking loss)
        if g_nsp_scores is not None:
            extra = g_nsp_scores.detach().to(device=dev, dtype=F32).reshape(B, 2).contiguous()
        L.nsp_loss_bwd(bw["nsp_pad"], nlab, w0, w1, gvec(g_nsp), dnsp, B, extra=extra)
        dfused = self._linear32_bwd(dnsp, bw["fused"], "nsp")
        dpt, dpv = torch.empty_like(dfused), torch.empty_like(dfused)
        L.mul_dropout_bwd(bw["pooled_t"], bw["pooled_v"], dfused, dpt, dpv, dfused.numel(), bw["d_fuse"],
                          fusion_sum=cfg.fusion_method == "sum")
        # pooler input gradients land on the first-token rows
        dcls_t = self._linear32_bwd(dpt, bw["cls_t"], "tpool")
        L.rows_add_f32(dseq_t, bw["cls_idx_t"], dcls_t, B, dcls_t.shape[1])
        dcls_v = self._linear32_bwd(dpv, bw["cls_v"], "vpool")
        self._to_img(dcls_v)                     # the image pooler's input gradient joins the image head's on the image stream
        with self._img_if(self.image_head_side):
            L.rows_add_f32(dseq_v, bw["cls_idx_v"], dcls_v, B, dcls_v.shape[1])
        self._bucket_done("heads")
        # ---- encoder blocks in reverse -------------------------------------------------------------
        gt, gv = dseq_t, dseq_v
        entries = list(reversed(bw["tape"]))
        if not self.image_head_side:
            self._to_img(gv)                     # the heads' gradient of the image stream was produced on the main stream
        self._backward_encoder(bw, entries, gt, gv)
        self._to_txt()                                       # everything joined before the caller continues
        self._bucket_done("text_embeddings")
        self.arena.fresh = False                             # the arena holds this pass's gradients now
        self._bwd_fresh = False

    def _backward_encoder(self, bw, entries, gt, gv):
        """The encoder blocks of the tape in reverse (`entries`), then the two embedding backward passes; every block's bucket
        is reported as it completes (`_bucket_done`).  Shared by the bf16 and the fp32x3 engine."""
        pos = 0
        embv_done = False

        def image_tail():
            # The image stream's last work -- the image embedding's backward -- is enqueued as soon as no image or connection
            # layer is left on the tape (the full config: after c0, with text layers 5..0 still to come), and the image side's
            # queued weight gradients are launched with it: under data parallelism the image-side buckets (v0, c0's image half,
            # the image embedding) then travel while the remaining text layers run, instead of after the end of backward behind
            # everything else (bucket_plan.py: the last collective of an 8-rank step shrinks from 392 MB to 181 MB).
            nonlocal embv_done
            if embv_done or any(e[0] in ("c", "v") for e in entries[pos:]):
                return
            embv_done = True
            with self._img():
                if self.cfg.fixed_v_layer == 0:              # (frozen lower image layers: no gradient reaches the image embedding)
                    bw["embv"](gv)
                self._bucket_done("image_embeddings", force_img=True)

        while pos < len(entries):
            image_tail()
            seg = []
            while pos < len(entries) and entries[pos][0] != "c":
                seg.append(entries[pos])
                pos += 1
            for kind, key, fn in seg:                        # image layers of this segment beside its text layers
                if kind == "v":
                    with self._img():
                        gv = fn(gv)
                        self._bucket_done(key)
            for kind, key, fn in seg:
                if kind == "t":
                    gt = fn(gt)
                    self._bucket_done(key)
            if pos < len(entries):
                _, key, fn = entries[pos]
                pos += 1
                with self._conn_tag():
                    gv, gt = fn(gv, gt)                      # exchanges between the streams happen inside
                self._bucket_done(key)
        image_tail()
        if self.cfg.fixed_t_layer == 0:
            bw["embt"](gt)
        # blocks without a backward step (frozen layers, connection layers under with_coattention = False) still own a gradient
        # bucket: report them, so that a data-parallel wrapper sees every bucket once per step (their gradients are zero / None)
        for g, _, _ in self.arena.buckets:
            if g not in self._reported and g != "text_embeddings":
                self._bucket_done(g)

    def _bucket_done(self, group, force_img=False):
        """A block's backward is enqueued.  Its weight gradients may stay queued for a later grouped launch (`_flush_due`);
        the data-parallel hook of a bucket fires once the launches that cover it are enqueued.  Buckets are handed over as
        they complete, not in the order they were finished: a bucket whose launches are out is not held back by an older one
        that still waits for the other stream's queue (bucket_plan.py restates this rule on the host)."""
        last = group == "text_embeddings"
        force = last or group == "heads"          # the decoder's gradient has its own row count: a launch of its own
        self._reported.add(group)
        if self.grad_bucket_hook is not None:
            self._pending.append((group, self._nq[0], self._nq[1]))
        self._flush_wgrad(force=force, force_img=force_img)
        if self.grad_bucket_hook is not None:
            ready = [] if self._on_side else [p for p in self._pending if self._nf[0] >= p[1] and self._nf[1] >= p[2]]
            if ready:
                self._join_wgrad()                # the exchange reads them
                self._to_txt()                    # ... including the ones the image side produced for these buckets
                self._pending = [p for p in self._pending if p not in ready]
                for j, (g, _, _) in enumerate(ready):
                    self.grad_bucket_hook(g, j + 1 < len(ready))    # more: the next bucket follows at once (adjacent slices travel together)
            assert not (last and self._pending), self._pending
        elif last:                                # last bucket: everything joined before the caller continues
            self._join_wgrad()
            self._to_txt()                        # ... including the image side's last grouped launch
