import os, sys
sys.path.insert(0, os.getcwd())
import torch
from unimm_amd import VisualDialogEncoder, synth, lib as L
dev = torch.device("cuda", 0)
from unimm_amd.config import BertConfig
cfg = BertConfig.from_json_file("unimm_amd/config/bert_base_6layer_6conect.json")
for n in (30, 240):
    b = synth.make_batch(n_seq=n, cfg=cfg, seed=1234, device=dev)
    am, cm, labels, weights = b["attention_mask"], b["co_attention_mask"], b["masked_lm_labels"], b["lm_weight"]
    B, T = labels.shape
    valid = am.ne(0).any(dim=1) | am.ne(0).any(dim=2)
    valid = valid | cm.ne(0).any(dim=1) | labels.ne(-1) | weights.ne(0)
    idx = torch.arange(1, T + 1, device=dev, dtype=torch.int32)
    lens = (valid.to(torch.int32) * idx).amax(dim=1).clamp_min(1)
    tw, cw = L.mask_pack(am), L.mask_pack(cm)
    nw = tw.shape[-1]
    R = cm.shape[1]
    h = L.plan_lengths((tw, nw, T * nw), (cw, nw, R * nw), R, labels.int(), weights.int(), None, B, T).tolist()
    print(n, "lens equal", h[:B] == lens.tolist(), sum(h[:B]), int(lens.sum()), "weights dtype", weights.dtype, "labels", labels.dtype,
          "sel", sum(h[B:2*B]), int(weights.ne(0).sum()))
    if h[:B] != lens.tolist():
        d = [(i, a, c) for i, (a, c) in enumerate(zip(h[:B], lens.tolist())) if a != c]
        print(d[:10])
