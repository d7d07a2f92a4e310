import sys, os, time; sys.path.insert(0, os.getcwd())
import torch
from unimm_amd import VisualDialogEncoder, synth
from unimm_amd.optim import FusedAdamW, WarmupLinearScheduleNonZero, default_language_weights, reference_param_groups
dev = torch.device("cuda", 0)
COMPUTE = sys.argv[4] if len(sys.argv) > 4 else "bf16"      # or fp32x3
enc = VisualDialogEncoder("unimm_amd/config/bert_base_6layer_6conect.json", compute_dtype=COMPUTE).to(dev); enc.train()
opt = FusedAdamW(reference_param_groups(enc, 2e-5, 2e-5, default_language_weights(enc)), enc.bert_pretrained.engine, lr=2e-5)
sch = WarmupLinearScheduleNonZero(opt, 100, 1000)
losses = []
t0 = time.time()
# python tools/soak.py [steps=40] [sequences=240] [graphs: 0 | 1] [bf16 | fp32x3] [host]
STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 40
NSEQ = int(sys.argv[2]) if len(sys.argv) > 2 else 240
if len(sys.argv) > 3 and sys.argv[3] == "1":
    enc.bert_pretrained.engine.ensure(dev)
    enc.bert_pretrained.engine.enable_graphs()
HOST = len(sys.argv) > 5 and sys.argv[5] == "host"           # fifth argument "host": the batches stay in host memory (CPU tensors into forward())
if HOST:
    torch.set_num_threads(8)
batches = [synth.make_batch(n_seq=NSEQ, cfg=enc.bert_pretrained.config, seed=s, device="cpu" if HOST else dev,
                            **(dict(mask_dtype=torch.int64) if HOST else {})) for s in range(4)]
for it in range(STEPS):
    b = dict(batches[it % 4])                                                                  # 4 batches cycled: loss must fall
    nw = b.pop("nsp_weight")
    opt.zero_grad()
    lm, img, nsp = enc(b["input_ids"], b["image_feat"], b["image_loc"], sep_indices=b["sep_indices"], sep_len=b["sep_len"],
                       token_type_ids=b["token_type_ids"], token_position_ids=b["token_position_ids"], attention_mask=b["attention_mask"],
                       masked_lm_labels=b["masked_lm_labels"], next_sentence_label=b["next_sentence_label"],
                       image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
                       image_label=b["image_label"], image_target=b["image_target"], nsp_weight=nw, lm_weight=b["lm_weight"])
    loss = lm.mean() + nsp.mean() + img.mean()
    loss.backward(); opt.step(); sch.step()
    if it % max(1, STEPS // 8) == 0 or it == STEPS - 1:
        torch.cuda.synchronize()
        print(it, round(float(loss.detach()), 4), "alloc GB", round(torch.cuda.memory_allocated() / 2**30, 2), "reserved GB", round(torch.cuda.memory_reserved() / 2**30, 2), flush=True)
print("done", round(time.time() - t0, 1), "s")
