"""List the small torch ops (copies, fills, elementwise) one training step launches besides the HIP library
calls, grouped by the Python line that issued them (torch.profiler with stacks)."""
import sys, os, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from torch.profiler import profile, ProfilerActivity
from unimm_amd import VisualDialogEncoder, synth

dev = torch.device("cuda", 0)
enc = VisualDialogEncoder(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unimm_amd", "config", "bert_base_6layer_6conect.json")).to(dev)
enc.train()
model = enc.bert_pretrained
B = int(sys.argv[1]) if len(sys.argv) > 1 else 240
batch = synth.make_batch(n_seq=B, cfg=model.config, seed=1, device=dev)
nsp_w = batch.pop("nsp_weight")
model.engine.ensure(dev); model.engine.arena.attach_grads()

def step():
    model.engine.arena.zero_grads()
    lm, img, nsp = enc(batch["input_ids"], batch["image_feat"], batch["image_loc"], sep_indices=batch["sep_indices"], sep_len=batch["sep_len"],
                       token_type_ids=batch["token_type_ids"], token_position_ids=batch["token_position_ids"], attention_mask=batch["attention_mask"],
                       masked_lm_labels=batch["masked_lm_labels"], next_sentence_label=batch["next_sentence_label"],
                       image_attention_mask=batch["image_attention_mask"], co_attention_mask=batch["co_attention_mask"],
                       image_label=batch["image_label"], image_target=batch["image_target"], nsp_weight=nsp_w, lm_weight=batch["lm_weight"])
    (lm.mean() + nsp.mean() + img.mean()).backward()

for _ in range(2): step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
by = collections.Counter(); dur = collections.Counter()
for ev in prof.events():
    if ev.device_type.name != "CPU" or not ev.name.startswith("aten::") or ev.cpu_parent is not None and ev.cpu_parent.name.startswith("aten::"):
        continue
    st = [f for f in (ev.stack or []) if "unimm_amd" in f or "tools/" in f]
    where = st[0].split("/")[-1] if st else "?"
    by[(ev.name, where)] += 1; dur[(ev.name, where)] += ev.device_time_total
print("top-level aten ops in one step:", sum(by.values()))
for (k, n) in by.most_common(45):
    print(f"{n:5d} x {k[0]:28s} {dur[k] / 1e3:8.3f} ms gpu   {k[1]}")
