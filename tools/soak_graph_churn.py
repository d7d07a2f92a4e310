"""Soak of the step executor under churn: signatures that keep evicting each other (max_entries = 2 against five batch sizes), a
second model with an executor of its own built and dropped every 40 steps, optimizer steps in between.  What it watches: the
process survives (destroyed graph execs used to leave the HIP runtime open to a fault in hipGraphLaunch, see
unimm_amd/graphs.py), reserved memory stays bounded (the pools of dead entries are captured into again), the loss falls.
python tools/soak_graph_churn.py [steps=300] [full | small]"""
import gc, os, sys, time
sys.path.insert(0, os.getcwd())
import json
import torch
from unimm_amd import BertConfig, BertForMultiModalPreTraining, synth
from unimm_amd import graphs as G
from unimm_amd.optim import FusedAdamW

STEPS = int(sys.argv[1]) if len(sys.argv) > 1 else 300
FULL = (sys.argv[2] if len(sys.argv) > 2 else "full") == "full"
dev = torch.device("cuda", 0)
cfg = BertConfig.from_json_file("unimm_amd/config/bert_base_6layer_6conect.json") if FULL else \
    BertConfig.from_dict(json.load(open("tests/golden/small_config.json")))
T = 256 if FULL else 64


def build():
    m = BertForMultiModalPreTraining(cfg).to(dev)
    m.train()
    m.engine.ensure(dev)
    gx = m.engine.enable_graphs(capture_after=0, max_entries=2)
    return m, gx


def step(m, b, opt=None):
    if opt is not None:
        opt.zero_grad()
    else:
        m.engine.arena.zero_grads()
    lm, img, nsp_l, _, _, _ = m(b["input_ids"], b["image_feat"], b["image_loc"], token_type_ids=b["token_type_ids"],
                                position_ids=b["token_position_ids"], attention_mask=b["attention_mask"],
                                image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
                                masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
                                next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"],
                                _want_lm_scores=False)
    loss = (lm + img + nsp_l).sum()
    loss.backward()
    if opt is not None:
        opt.step()
    return loss


model, gx = build()
opt = FusedAdamW([dict(params=list(model.parameters()), lr=2e-5, weight_decay=0.01)], model.engine, lr=2e-5)
sizes = [6, 8, 10, 12, 14]
batches = {n: synth.make_batch(n_seq=n, T=T, R=37, cfg=cfg, seed=n, device=dev) for n in sizes}
side = None
t0 = time.time()
first = None
for it in range(STEPS):
    n = sizes[(it * 7 + it // 3) % len(sizes)]
    loss = step(model, batches[n], opt)
    if it % 40 == 5:                                     # a second model + executor lives for 20 steps, then goes away
        side = build()
    if side is not None:
        step(side[0], batches[sizes[it % 2]])
        if it % 40 == 25:
            side = None
            gc.collect()
    if it % 25 == 0 or it == STEPS - 1:
        torch.cuda.synchronize()
        lv = float(loss.detach())
        first = lv if first is None else first
        print(f"step {it}: loss {lv:.4f}  execs kept {len(G._KEPT)}  free pools {sum(len(v) for v in G._FREE_POOLS.values())}  "
              f"captures {gx.stats['captures']} replays {gx.stats['replays']}  allocated {torch.cuda.memory_allocated() / 2**30:.2f} GB  "
              f"reserved {torch.cuda.memory_reserved() / 2**30:.2f} GB  {time.time() - t0:.0f} s", flush=True)
print("done: loss", first, "->", lv, "in", round(time.time() - t0, 1), "s")
# where the reserved memory sits: segments by memory pool (pool (0, 0) = the caching allocator's own)
import collections
tot, act, nseg = collections.Counter(), collections.Counter(), collections.Counter()
for seg in torch.cuda.memory_snapshot():
    k = tuple(seg.get("segment_pool_id", (0, 0)))
    tot[k] += seg["total_size"]; act[k] += seg["allocated_size"]; nseg[k] += 1
for k, v in sorted(tot.items(), key=lambda kv: -kv[1])[:12]:
    print(f"  pool {k}: reserved {v / 2**30:6.2f} GB in {nseg[k]} segments, allocated {act[k] / 2**30:6.2f} GB")
print("  pools:", len(tot))
