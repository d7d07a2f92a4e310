"""Full-config check of the two-stream schedule: the same steps with engine.dual_stream on and off must give
bit-identical losses and gradients equal up to the order of the weight-gradient atomics.  Varying batches and
batch sizes, several repeats (a missing stream dependency shows up as a mismatch here)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import VisualDialogEncoder, synth

dev = torch.device("cuda", 0)
COMPUTE = sys.argv[2] if len(sys.argv) > 2 else "bf16"      # or fp32x3
enc = VisualDialogEncoder(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unimm_amd", "config",
                                       "bert_base_6layer_6conect.json"), compute_dtype=COMPUTE).to(dev).train()
model = enc.bert_pretrained
eng = model.engine
eng.ensure(dev)
eng.arena.attach_grads()
worst = 0.0
bad = 0
for rep in range(int(sys.argv[1]) if len(sys.argv) > 1 else 3):
    for n_seq in (240, 30, 96, 12):
        b = synth.make_batch(n_seq=n_seq, cfg=model.config, seed=100 * rep + n_seq, device=dev)
        nw = b.pop("nsp_weight")
        res = []
        for dual in (False, True):
            eng.dual_stream = dual
            model.set_dropout_seed(9, step=rep)
            eng.arena.zero_grads()
            lm, img, nsp = enc(b["input_ids"], b["image_feat"], b["image_loc"], sep_indices=b["sep_indices"], sep_len=b["sep_len"],
                               token_type_ids=b["token_type_ids"], token_position_ids=b["token_position_ids"],
                               attention_mask=b["attention_mask"], masked_lm_labels=b["masked_lm_labels"],
                               next_sentence_label=b["next_sentence_label"], image_attention_mask=b["image_attention_mask"],
                               co_attention_mask=b["co_attention_mask"], image_label=b["image_label"], image_target=b["image_target"],
                               nsp_weight=nw, lm_weight=b["lm_weight"])
            (lm.mean() + nsp.mean() + img.mean()).backward()
            torch.cuda.synchronize()
            res.append(((float(lm), float(img), float(nsp)), eng.arena.grad_flat.clone()))
        (l0, g0), (l1, g1) = res
        rel = float((g0 - g1).abs().max() / g0.abs().max())
        worst = max(worst, rel)
        ok = l0 == l1 and rel <= 1e-5 and bool(torch.isfinite(g1).all())
        bad += 0 if ok else 1
        print(f"rep {rep} n_seq {n_seq:4d}: losses {'equal' if l0 == l1 else f'DIFFER {l0} {l1}'}, grad rel diff {rel:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
# second phase: steps back to back WITHOUT a host sync in between (gradients accumulate over 4 different batches):
# catches a missing dependency between one step's image-side work and the next step's main-stream work
batches = []
for i in range(4):
    b = synth.make_batch(n_seq=(60, 24, 120, 30)[i], cfg=model.config, seed=900 + i, device=dev)
    batches.append((b, b.pop("nsp_weight")))
for rep in range(3):
    acc = []
    for dual in (False, True):
        eng.dual_stream = dual
        model.set_dropout_seed(3, step=10 * rep)
        eng.arena.zero_grads()
        for b, nw in batches:
            lm, img, nsp = enc(b["input_ids"], b["image_feat"], b["image_loc"], sep_indices=b["sep_indices"], sep_len=b["sep_len"],
                               token_type_ids=b["token_type_ids"], token_position_ids=b["token_position_ids"],
                               attention_mask=b["attention_mask"], masked_lm_labels=b["masked_lm_labels"],
                               next_sentence_label=b["next_sentence_label"], image_attention_mask=b["image_attention_mask"],
                               co_attention_mask=b["co_attention_mask"], image_label=b["image_label"], image_target=b["image_target"],
                               nsp_weight=nw, lm_weight=b["lm_weight"])
            (lm.mean() + nsp.mean() + img.mean()).backward()
        torch.cuda.synchronize()
        acc.append(eng.arena.grad_flat.clone())
    rel = float((acc[0] - acc[1]).abs().max() / acc[0].abs().max())
    ok = rel <= 1e-5 and bool(torch.isfinite(acc[1]).all())
    bad += 0 if ok else 1
    worst = max(worst, rel)
    print(f"back-to-back rep {rep}: accumulated grad rel diff {rel:.2e} {'ok' if ok else 'MISMATCH'}", flush=True)
print("worst", worst, "mismatches", bad)
sys.exit(1 if bad else 0)
