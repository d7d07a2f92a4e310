"""Drop-in model surface: `BertForMultiModalPreTraining(config)` and `VisualDialogEncoder(config_path)`
with the reference's constructor / forward() signatures, return tuples and `state_dict` names
(models/vilbert_dialog.py:1496-1626, models/visual_dialog_encoder.py:8-50), computed by the HIP
engine (unimm_amd.engine).  The modules below only own parameters; all arithmetic is in the kernels.
"""
from __future__ import annotations

import logging
import os
import warnings

import torch
from torch import nn

from . import params as PM
from .config import BertConfig
from .engine import Engine

logger = logging.getLogger(__name__)


def _build_param_tree(root: nn.Module, cfg) -> None:
    """Register every parameter under nested plain containers so that state_dict keys equal the
    reference's ("bert.encoder.layer.0.attention.self.query.weight", ...).  Init follows
    init_bert_weights (models/vilbert_dialog.py:1110-1121): N(0, initializer_range) for Linear /
    Embedding weights, zero biases, LayerNorm weight 1 / bias 0."""
    for name, shape in PM.state_dict_names(cfg).items():
        if name == PM.TIED_DECODER:
            continue
        parts = name.split(".")
        mod = root
        for part in parts[:-1]:
            if not hasattr(mod, part):
                mod.add_module(part, nn.Module())
            mod = getattr(mod, part)
        if "LayerNorm" in name:
            t = torch.ones(shape) if parts[-1] == "weight" else torch.zeros(shape)
        elif parts[-1] == "weight":
            t = torch.empty(shape).normal_(mean=0.0, std=cfg.initializer_range)
        else:
            t = torch.zeros(shape)
        mod.register_parameter(parts[-1], nn.Parameter(t))
    # tie the decoder to the word embeddings (models/vilbert_dialog.py:1020, :1504-1506)
    root.cls.predictions.add_module("decoder", nn.Module())
    root.cls.predictions.decoder.register_parameter("weight", root.bert.embeddings.word_embeddings.weight)


def load_pretrained_state_dict(model: nn.Module, state_dict, default_gpu=True):
    """models/vilbert_dialog.py:1232-1296 on a plain key -> tensor mapping.  Returns (missing_keys, unexpected_keys)."""
    sd = {}
    for key, v in state_dict.items():
        new_key = key
        if "gamma" in new_key:
            new_key = new_key.replace("gamma", "weight")
        if "beta" in new_key:
            new_key = new_key.replace("beta", "bias")
        if new_key.startswith("bert_pretrained."):          # a VisualDialogEncoder checkpoint (train.py:503-505)
            new_key = new_key[len("bert_pretrained."):]
        sd[new_key] = v
    start_prefix = ""
    if not hasattr(model, "bert") and any(k.startswith("bert.") for k in sd):
        start_prefix = "bert."
    own = model.state_dict()
    missing, errors, usable = [], [], {}
    for k, t in own.items():
        src = sd.get(start_prefix + k)
        if src is None:
            missing.append(k)
        elif tuple(src.shape) != tuple(t.shape):
            errors.append(f"size mismatch for {k}: copying a param with shape {tuple(src.shape)} from checkpoint, "
                          f"the shape in current model is {tuple(t.shape)}.")
        else:
            usable[k] = src
    expected = {start_prefix + k for k in own}
    unexpected = [k for k in sd if k not in expected]
    if errors and default_gpu:
        raise RuntimeError("Error(s) in loading state_dict for {}:\n\t{}".format(model.__class__.__name__, "\n\t".join(errors)))
    model.load_state_dict(usable, strict=False)
    if missing and default_gpu:
        logger.info("Weights of {} not initialized from pretrained model: {}".format(model.__class__.__name__, missing))
    if unexpected and default_gpu:
        logger.info("Weights from pretrained model not used in {}: {}".format(model.__class__.__name__, unexpected))
    return missing, unexpected


class _HotPath(torch.autograd.Function):
    """One autograd node for the whole forward/backward hot path.  Parameter gradients are
    accumulated in place into the flat gradient arena (every Parameter's .grad is a view of it), so
    backward returns no tensors; `anchor` only makes autograd call us."""

    @staticmethod
    def forward(ctx, anchor, engine, inp, opts):
        ctx.set_materialize_grads(False)
        ctx.graph_entry = None
        if engine.graphs is not None and engine.graphs.eligible(inp, opts):     # the step as two replayed hipGraphs
            res = engine.graphs.forward(inp, opts)
            if res is not None:
                ctx.engine, ctx.out, ctx.graph_entry = engine, None, res[4]
                ctx.graph_token = res[4].out.pop("_token", None)   # alive until backward ran: the entry is not replayed meanwhile
                engine.last_seq_t = (None, None)
                return res[0], res[1], res[2], res[3]
        out = engine.forward(inp, train=opts["train"], save=True, lm_rows="labelled", want_pred_v=True)
        losses = engine.losses(out, inp)
        ctx.engine, ctx.out = engine, out
        engine.last_seq_t = (engine.padded(out, out["seq32_t"]) if opts["want_seq"] else None,
                             engine.padded(out, out["seq_out_t"]) if opts["want_seq"] else None)
        nsp = out["nsp"].clone()
        return losses["lm_loss"], losses["img_loss"], losses["nsp_loss"], nsp

    @staticmethod
    def backward(ctx, g_lm, g_img, g_nsp, g_scores):
        engine, out = ctx.engine, ctx.out
        ctx.out = None
        if ctx.graph_entry is not None:
            engine.graphs.backward(ctx.graph_entry, g_lm, g_img, g_nsp, g_scores)
            ctx.graph_token = None
            return None, None, None, None
        engine.backward(out, g_lm, g_img, g_nsp, g_scores)
        return None, None, None, None


class BertForMultiModalPreTraining(nn.Module):
    """BERT model with multi modal pre-training heads (drop-in for models/vilbert_dialog.py:1496)."""

    def __init__(self, config, compute_dtype="bf16"):
        """`compute_dtype` (an extension of the reference's one-argument constructor): "bf16" = the reference's autocast
        class (train.py:445: half-precision matmuls, fp32 LayerNorm / softmax / losses) on the bf16 MFMA engine; "fp32x3" =
        fp32-grade arithmetic end to end for callers that run WITHOUT autocast (dense_annotation_finetuning.py:253), on the
        same GEMM kernels over split operands (unimm_amd/engine_x3.py)."""
        super().__init__()
        if not isinstance(config, BertConfig):
            raise ValueError(
                "Parameter config in `{}(config)` should be an instance of class `BertConfig`. ".format(
                    self.__class__.__name__))
        self.config = config
        _build_param_tree(self, config)
        self.predict_feature = config.predict_feature
        if compute_dtype not in ("bf16", "fp32x3"):
            raise ValueError(f"compute_dtype must be 'bf16' or 'fp32x3', got {compute_dtype!r}")
        self.compute_dtype = compute_dtype
        if compute_dtype == "fp32x3":
            from .engine_x3 import EngineX3
            self._engine = EngineX3(self, config)
        else:
            self._engine = Engine(self, config)
        # values loaded into the Parameters must reach the engine's bf16 / transposed weight copies
        self.register_load_state_dict_post_hook(lambda module, incompatible: module._engine.invalidate_weights())

    # -- construction helpers -----------------------------------------------------------------
    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, config, default_gpu=True, state_dict=None, *inputs, **kwargs):
        """Key handling of models/vilbert_dialog.py:1232-1296.  The reference downloads 'bert-base-uncased' here
        (:1123-1231); there is no network in this build, so weights come from `state_dict` or from a local
        checkpoint file / directory at the given path, otherwise the model keeps its fresh init (with a warning).
        A found state dict is loaded as the reference loads it: old LayerNorm names (`gamma` / `beta`) are renamed to
        `weight` / `bias` (:1233-1245), the `bert.` start prefix is detected (:1270-1274), every key of the model that
        the checkpoint lacks is reported as missing and every checkpoint key the model lacks as unexpected
        (:1276-1287, logged when `default_gpu`), and a shape mismatch raises (:1288-1294)."""
        model = cls(config, *inputs, **kwargs)
        if state_dict is None and isinstance(pretrained_model_name_or_path, str):
            path = pretrained_model_name_or_path
            if os.path.isdir(path):
                path = os.path.join(path, "pytorch_model.bin")
            if os.path.isfile(path):
                state_dict = torch.load(path, map_location="cpu")
        if state_dict is None:
            warnings.warn(f"from_pretrained({pretrained_model_name_or_path!r}): no local weights found, "
                          "keeping random initialisation (no network access)")
            return model
        missing, unexpected = load_pretrained_state_dict(model, state_dict, default_gpu=default_gpu)
        model.pretrained_missing_keys, model.pretrained_unexpected_keys = missing, unexpected
        return model

    @property
    def engine(self) -> Engine:
        return self._engine

    def set_dropout_seed(self, seed: int, step: int = 0):
        self._engine.seed, self._engine.step = int(seed), int(step)

    # -- forward ---------------------------------------------------------------------------------
    def forward(self, input_ids, image_feat, image_loc, sep_indices=None, sep_len=None, token_type_ids=None,
                position_ids=None, attention_mask=None, image_attention_mask=None, co_attention_mask=None,
                masked_lm_labels=None, image_label=None, image_target=None, next_sentence_label=None,
                output_all_attention_masks=False, nsp_weight=None, lm_weight=None,
                _want_lm_scores=True, _want_pred_v=True, image_index=None):
        """Same contract as models/vilbert_dialog.py:1519-1626.  `sep_indices` / `sep_len` are accepted and
        ignored exactly as the reference's embeddings do (:326-356).  Train branch (labels, NSP label and
        image target all given) -> (lm_loss[1], img_loss[1], nsp_loss[1], seq_out_t, pred_t, nsp[B,2]);
        otherwise -> (pred_t, pred_v, nsp, seq_out_t, attention-lists).
        `_want_lm_scores=False` (used by VisualDialogEncoder when the caller does not ask for LM scores)
        skips materialising the dense [B,T,vocab] logits; the loss only ever needs the labelled rows.
        Extensions (unimm_amd/inputs.py): `attention_mask` may be a `DialogMaskSpec`
        (then `co_attention_mask` must be None) and `image_feat` / `image_loc` / `image_target` may hold one
        entry per IMAGE with `image_index` [B] mapping sequences to them."""
        eng = self._engine
        dev = self._device()
        eng.ensure(dev)
        inp = dict(input_ids=input_ids, image_feat=image_feat, image_loc=image_loc, token_type_ids=token_type_ids,
                   position_ids=position_ids, attention_mask=attention_mask, image_attention_mask=image_attention_mask,
                   co_attention_mask=co_attention_mask, masked_lm_labels=masked_lm_labels, image_label=image_label,
                   image_target=image_target, next_sentence_label=next_sentence_label, nsp_weight=nsp_weight,
                   lm_weight=lm_weight, image_index=image_index)
        eng.stage_host_inputs(inp)             # the reference's callers pass CPU tensors (train.py:113-161): pinned ring + copy stream
        B, T = input_ids.shape
        H, V = self.config.hidden_size, self.config.vocab_size
        train_branch = masked_lm_labels is not None and next_sentence_label is not None and image_target is not None
        if self.training:
            eng.step += 1
        if train_branch:
            if torch.is_grad_enabled():
                lm_loss, img_loss, nsp_loss, nsp = _HotPath.apply(eng._anchor, eng, inp,
                                                                  dict(train=self.training, want_seq=_want_lm_scores))
                seq_raw = eng.last_seq_t
            else:
                out = eng.forward(inp, train=self.training, save=False, lm_rows="labelled", want_pred_v=True)
                ls = eng.losses(out, inp)
                lm_loss, img_loss, nsp_loss, nsp = ls["lm_loss"], ls["img_loss"], ls["nsp_loss"], out["nsp"]
                seq_raw = (eng.padded(out, out["seq32_t"]), eng.padded(out, out["seq_out_t"])) if _want_lm_scores else None
            eng.last_seq_t = None
            seq_t = pred_t = None
            if _want_lm_scores:
                with torch.no_grad():   # dense scores are an output only; the loss path is row-sparse
                    seq_t = seq_raw[0].view(B, T, H)
                    pred_t = eng.decode_rows(seq_raw[1], B * T).view(B, T, -1)[:, :, :V]
            return lm_loss, img_loss, nsp_loss, seq_t, pred_t, nsp
        att = ([], [], [])
        with torch.no_grad():
            if output_all_attention_masks:     # :855-929, :1626 -- diagnostic output, padded one-stream schedule
                out, att = eng.forward_with_attention(inp, train=self.training, lm_rows="all" if _want_lm_scores else "none",
                                                      want_pred_v=_want_pred_v)
            else:
                out = eng.forward(inp, train=self.training, save=False, lm_rows="all" if _want_lm_scores else "none",
                                  want_pred_v=_want_pred_v)
        seq_t = eng.padded(out, out["seq32_t"]).view(B, T, H)
        return out.get("pred_t"), out.get("pred_v"), out["nsp"], seq_t, att

    def forward_backward(self, input_ids, image_feat, image_loc, loss_weights, sep_indices=None, sep_len=None, token_type_ids=None,
                         position_ids=None, attention_mask=None, image_attention_mask=None, co_attention_mask=None,
                         masked_lm_labels=None, image_label=None, image_target=None, next_sentence_label=None, nsp_weight=None,
                         lm_weight=None, image_index=None, plan_header=None):
        """The training step's forward AND backward in one call (an extension; the reference writes
        `loss = c_lm * lm.mean() + c_nsp * nsp.mean() + c_img * img.mean(); loss.backward()`, train.py:164-168, :315).
        loss_weights = (c_lm, c_nsp, c_img).  With the weights known up front the backward does not wait for the host to see
        the losses: both halves are enqueued back to back (two graph replays under the step executor) -- no autograd round trip
        (a hand-over to the autograd thread and ~8 one-element launches between the halves: 0.2-0.4 ms during which a
        30-sequence step's GPU idles).  Parameter gradients accumulate into `.grad` exactly as with `loss.backward()`.
        -> (loss, lm_loss, img_loss, nsp_loss, nsp_logits).  plan_header: `engine.count_rows(...)` of this batch if the caller
        has it already (a prefetcher: the step then starts without its host sync)."""
        if masked_lm_labels is None or next_sentence_label is None or image_target is None:
            raise ValueError("forward_backward needs the training inputs (masked_lm_labels, next_sentence_label, image_target)")
        eng = self._engine
        dev = self._device()
        eng.ensure(dev)
        inp = dict(input_ids=input_ids, image_feat=image_feat, image_loc=image_loc, token_type_ids=token_type_ids,
                   position_ids=position_ids, attention_mask=attention_mask, image_attention_mask=image_attention_mask,
                   co_attention_mask=co_attention_mask, masked_lm_labels=masked_lm_labels, image_label=image_label,
                   image_target=image_target, next_sentence_label=next_sentence_label, nsp_weight=nsp_weight,
                   lm_weight=lm_weight, image_index=image_index)
        eng.stage_host_inputs(inp)
        if plan_header is not None:
            inp["_plan_header"] = plan_header
        if self.training:
            eng.step += 1
        c_lm, c_nsp, c_img = (float(c) for c in loss_weights)
        key = (c_lm, c_nsp, c_img, dev)
        seeds = self._fb_seeds.get(key) if hasattr(self, "_fb_seeds") else None
        if seeds is None:
            if not hasattr(self, "_fb_seeds"):
                self._fb_seeds = {}
            w = torch.tensor([c_lm, c_img, c_nsp], dtype=torch.float32, device=dev)
            seeds = self._fb_seeds[key] = (w[0:1], w[1:2], w[2:3], w)
        g_lm, g_img, g_nsp, w = seeds
        opts = dict(train=self.training, want_seq=False)
        with torch.no_grad():
            res = None
            if eng.graphs is not None and eng.graphs.eligible(inp, opts):
                res = eng.graphs.forward(inp, opts, defer_outputs=True)
            if res is not None:
                ent = res[4]
                tok = ent.out.pop("_token", None)
                eng.graphs.backward(ent, g_lm, g_img, g_nsp, None)
                del tok
                lm_loss, img_loss, nsp_loss, nsp = eng.graphs.outputs(ent)
            else:
                inp.pop("_plan_header", None)
                out = eng.forward(inp, train=self.training, save=True, lm_rows="labelled", want_pred_v=True)
                ls = eng.losses(out, inp)
                lm_loss, img_loss, nsp_loss, nsp = ls["lm_loss"], ls["img_loss"], ls["nsp_loss"], out["nsp"].clone()
                eng.backward(out, g_lm.reshape(lm_loss.shape), g_img.reshape(img_loss.shape), g_nsp.reshape(nsp_loss.shape), None)
            eng.last_seq_t = None
            loss = torch.cat((lm_loss.reshape(1), img_loss.reshape(1), nsp_loss.reshape(1))).dot(w).reshape(())
        return loss, lm_loss, img_loss, nsp_loss, nsp

    def _device(self):
        return self.bert.embeddings.word_embeddings.weight.device

    # generative scoring without the [B,T,vocab] tensor (val_lm.py:121-136 / val_avg_lm.py:135)
    @torch.no_grad()
    def sequence_log_likelihood(self, input_ids, image_feat, image_loc, masked_lm_labels, average=False, shared_context=None, **kw):
        """-sum_t CE(pred_t, labels, ignore_index=-1) per sequence, computed on the labelled rows only by
        the fused decoder + log-softmax kernels.  Returns (scores[B] fp32, nsp[B,2]).
        shared_context (an extension): one group id per sequence -- sequences of a group share image and dialog context and
        differ only in the candidate answer, as the 100 options of a round in val_lm.py:52-121 do (pass the round index).  The
        context rows and the image stream are then computed once per group (unimm_amd/scoring.py); a sequence whose context does
        not match its group comes back as NaN.  bf16 engine; the fp32x3 engine takes the per-sequence path."""
        if shared_context is not None and self.compute_dtype == "bf16":
            from .scoring import sequence_log_likelihood_shared
            return sequence_log_likelihood_shared(self, input_ids, image_feat, image_loc, masked_lm_labels, shared_context,
                                                  average=average, **kw)
        eng = self._engine
        eng.ensure(self._device())
        inp = dict(input_ids=input_ids, image_feat=image_feat, image_loc=image_loc, masked_lm_labels=masked_lm_labels,
                   lm_weight=None, **kw)
        eng.stage_host_inputs(inp)                 # val_lm.py hands CPU tensors over chunk by chunk (val_lm.py:86-121)
        out = eng.forward(inp, train=False, save=False, lm_rows="labelled", want_pred_v=False)
        input_ids = inp["input_ids"]
        B, T = input_ids.shape
        scores = torch.zeros(B, dtype=torch.float32, device=eng.arena.device)
        lm = out.get("lm")
        if lm is not None:
            from . import lib as L
            seg = (lm["pos_idx"] // T).to(torch.int32)
            L.segment_sum(lm["rownll"], seg, scores, lm["n"], -1.0)
            if average:
                cnt = torch.bincount(seg.long(), minlength=B).clamp_min(1)
                scores = scores / cnt
        return scores, out["nsp"]


class VisualDialogEncoder(nn.Module):
    """Drop-in for models/visual_dialog_encoder.py:8-50 (what train.py / val_lm.py instantiate)."""

    def __init__(self, config_path, pretrained="bert-base-uncased", compute_dtype="bf16"):
        super().__init__()
        config = BertConfig.from_json_file(config_path)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            self.bert_pretrained = BertForMultiModalPreTraining.from_pretrained(pretrained, config, compute_dtype=compute_dtype)
        self.bert_pretrained.train()

    def forward(self, input_ids, image_feat, image_loc, sep_indices=None, sep_len=None, token_type_ids=None,
                token_position_ids=None, attention_mask=None, masked_lm_labels=None, next_sentence_label=None,
                head_mask=None, random_round_indices=None, output_nsp_scores=False, output_lm_scores=False,
                image_attention_mask=None, co_attention_mask=None, image_label=None, image_target=None,
                nsp_weight=None, lm_weight=None, image_index=None):
        masked_lm_loss = masked_img_loss = nsp_loss = None
        kw = dict(sep_indices=sep_indices, sep_len=sep_len, token_type_ids=token_type_ids,
                  position_ids=token_position_ids, attention_mask=attention_mask, masked_lm_labels=masked_lm_labels,
                  next_sentence_label=next_sentence_label, image_attention_mask=image_attention_mask,
                  co_attention_mask=co_attention_mask, image_label=image_label, image_target=image_target,
                  nsp_weight=nsp_weight, lm_weight=lm_weight, _want_lm_scores=output_lm_scores, image_index=image_index)
        if next_sentence_label is not None and masked_lm_labels is not None and image_target is not None:
            masked_lm_loss, masked_img_loss, nsp_loss, _, prediction_scores_t, seq_relationship_score = \
                self.bert_pretrained(input_ids, image_feat, image_loc, **kw)
        else:
            prediction_scores_t, _, seq_relationship_score, _, _ = \
                self.bert_pretrained(input_ids, image_feat, image_loc, _want_pred_v=False, **kw)
        out = (masked_lm_loss, masked_img_loss, nsp_loss)
        if output_nsp_scores:
            out = out + (seq_relationship_score,)
        if output_lm_scores:
            out = out + (prediction_scores_t,)
        return out

    def forward_backward(self, input_ids, image_feat, image_loc, loss_weights, sep_indices=None, sep_len=None, token_type_ids=None,
                         token_position_ids=None, attention_mask=None, masked_lm_labels=None, next_sentence_label=None,
                         image_attention_mask=None, co_attention_mask=None, image_label=None, image_target=None,
                         nsp_weight=None, lm_weight=None, image_index=None, plan_header=None):
        """forward + `(c_lm * lm + c_nsp * nsp + c_img * img).backward()` in one call, loss_weights = (c_lm, c_nsp, c_img)
        (BertForMultiModalPreTraining.forward_backward).  -> (loss, lm_loss, img_loss, nsp_loss, nsp_logits)."""
        return self.bert_pretrained.forward_backward(
            input_ids, image_feat, image_loc, loss_weights, sep_indices=sep_indices, sep_len=sep_len, token_type_ids=token_type_ids,
            position_ids=token_position_ids, attention_mask=attention_mask, image_attention_mask=image_attention_mask,
            co_attention_mask=co_attention_mask, masked_lm_labels=masked_lm_labels, image_label=image_label, image_target=image_target,
            next_sentence_label=next_sentence_label, nsp_weight=nsp_weight, lm_weight=lm_weight, image_index=image_index,
            plan_header=plan_header)
