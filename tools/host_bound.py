"""Is a step at a given per-GPU batch limited by the host's launch rate?  Times the enqueue of K steps (no sync) and
the same K steps to completion."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from unimm_amd import VisualDialogEncoder, synth

B = int(sys.argv[1]) if len(sys.argv) > 1 else 30
COMPUTE = sys.argv[2] if len(sys.argv) > 2 else "bf16"      # or fp32x3
K = 20
dev = torch.device("cuda", 0)
cfgp = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unimm_amd", "config", "bert_base_6layer_6conect.json")
torch.manual_seed(0)
enc = VisualDialogEncoder(cfgp, compute_dtype=COMPUTE).to(dev).train()
model = enc.bert_pretrained
batch = synth.make_batch(n_seq=B, cfg=model.config, seed=1234, device=dev)
nsp_w = batch.pop("nsp_weight")
model.engine.ensure(dev)
model.engine.arena.attach_grads()


def step():
    model.engine.arena.zero_grads()
    lm, img, nsp = enc(batch["input_ids"], batch["image_feat"], batch["image_loc"], sep_indices=batch["sep_indices"],
                       sep_len=batch["sep_len"], token_type_ids=batch["token_type_ids"], token_position_ids=batch["token_position_ids"],
                       attention_mask=batch["attention_mask"], masked_lm_labels=batch["masked_lm_labels"],
                       next_sentence_label=batch["next_sentence_label"], image_attention_mask=batch["image_attention_mask"],
                       co_attention_mask=batch["co_attention_mask"], image_label=batch["image_label"],
                       image_target=batch["image_target"], nsp_weight=nsp_w, lm_weight=batch["lm_weight"])
    (lm.mean() + nsp.mean() + img.mean()).backward()


_wait = [0.0]
_orig_tolist = torch.Tensor.tolist


def _timed_tolist(self):
    a = time.perf_counter()
    r = _orig_tolist(self)
    if self.is_cuda:
        _wait[0] += time.perf_counter() - a
    return r


torch.Tensor.tolist = _timed_tolist
for _ in range(5):
    step()
torch.cuda.synchronize()
_wait[0] = 0.0
t0 = time.perf_counter()
for _ in range(K):
    step()
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"B={B}: enqueue {1e3 * (t1 - t0) / K:.2f} ms/step of which {1e3 * _wait[0] / K:.2f} ms waiting in the plan's device->host copy, "
      f"complete {1e3 * (t2 - t0) / K:.2f} ms/step "
      f"(dual_stream={model.engine.dual_stream})")
import cProfile, pstats
pr = cProfile.Profile()
eng = model.engine
orig_b, orig_f = eng._backward, eng._forward


def prof_call(fn):
    def w(*a, **k):
        pr.enable()
        try:
            return fn(*a, **k)
        finally:
            pr.disable()
    return w


eng._backward, eng._forward = prof_call(orig_b), prof_call(orig_f)
for _ in range(5):
    step()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(int(os.environ.get("HOST_BOUND_ROWS", "40")))
