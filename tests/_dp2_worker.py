"""Child process of tests/test_gpu_dp2.py: one data-parallel rank of the HIP engine.  Every rank sits on device 0
(a one-GPU box; RCCL refuses two ranks on one device, so the exchange runs over gloo) and runs the REAL engine path:
`DataParallelRCCL` -> `Engine.grad_bucket_hook` -> per-bucket all-reduce issued from inside backward, on the comm
stream, with the dual-stream joins of `Engine._bucket_done`.

    python tests/_dp2_worker.py <rank> <world> <port> <out.pt> [bf16|fp32] [allreduce|rs_ag] [gloo|nccl] [eager|graphs] [bf16|fp32x3]
(nccl = RCCL, rank r on device r: the path bench.py --gpus N takes; needs as many GPUs as ranks.
 graphs = the step executor of unimm_amd/graphs.py: the backward replayed as a chain of graphs cut at the bucket hand-overs)"""
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, out = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    wire = sys.argv[5] if len(sys.argv) > 5 else "fp32"
    algo = sys.argv[6] if len(sys.argv) > 6 else "allreduce"
    backend = sys.argv[7] if len(sys.argv) > 7 else "gloo"
    graphs = len(sys.argv) > 8 and sys.argv[8] == "graphs"
    compute = sys.argv[9] if len(sys.argv) > 9 else "bf16"          # "fp32x3": the fp32-accuracy engine (unimm_amd/engine_x3.py)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port)
    if backend == "nccl":                                 # real RCCL: one device per rank (needs >= `world` GPUs)
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    else:
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import vilbert_ref as R
    from unimm_amd import BertConfig, BertForMultiModalPreTraining
    from unimm_amd.parallel import DataParallelRCCL, shard_range
    gold = os.path.join(ROOT, "tests", "golden")
    cfgd = json.load(open(os.path.join(gold, "small_config.json")))
    model = BertForMultiModalPreTraining(BertConfig.from_dict(cfgd), compute_dtype=compute)
    # a different init on every rank: the wrapper's broadcast must make the replicas equal to rank 0's
    model.load_state_dict(R.init_state_dict(R.make_config(cfgd), seed=11 + 5 * rank), strict=True)
    model = model.cuda().eval()
    # a forward BEFORE wrapping: the bf16 weight copies exist, so the broadcast must invalidate them
    g = np.load(os.path.join(gold, "small_mixed.npz"))
    batch = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in::")}
    n = batch["input_ids"].shape[0]
    lo, hi = shard_range(n, rank, world)
    sh = {k: (v if k == "nsp_weight" else v[lo:hi]) for k, v in batch.items()}
    if graphs:                                            # the executor takes device-resident inputs
        sh = {k: (v if k == "nsp_weight" else v.cuda()) for k, v in sh.items()}
    args = (sh["input_ids"], sh["image_feat"], sh["image_loc"])
    kw = dict(token_type_ids=sh["token_type_ids"], position_ids=sh["position_ids"], attention_mask=sh["attention_mask"],
              image_attention_mask=sh["image_attention_mask"], co_attention_mask=sh["co_attention_mask"],
              masked_lm_labels=sh["masked_lm_labels"], image_label=sh["image_label"], image_target=sh["image_target"],
              next_sentence_label=sh["next_sentence_label"], nsp_weight=sh["nsp_weight"], lm_weight=sh["lm_weight"],
              _want_lm_scores=False)
    with torch.no_grad():
        model(*args, **kw)
    dp = DataParallelRCCL(model, wire_dtype=wire, algorithm=algo)
    seed_after = model.engine.seed
    gx = model.engine.enable_graphs(capture_after=0, row_bucket=64, lm_bucket=16) if graphs else None
    model.zero_grad(set_to_none=True)
    lm, img, nsp_l, _, _, _ = dp(*args, **kw)
    (lm + img + nsp_l).sum().backward()
    torch.cuda.synchronize()
    g1 = model.engine.arena.grad_flat.clone()
    with dp.no_sync():                                   # accumulation micro-step: local gradient on top, no exchange
        lm2, img2, nsp2, _, _, _ = dp(*args, **kw)
        (lm2 + img2 + nsp2).sum().backward()
    torch.cuda.synchronize()
    acc = model.engine.arena.grad_flat.clone()
    gstats = None
    if graphs:                                           # a third step: replayed forward + chain, exchange on again
        model.engine.arena.zero_grads()
        dp.comm_stats(reset=True)
        lm3, img3, nsp3, _, _, _ = dp(*args, **kw)
        (lm3 + img3 + nsp3).sum().backward()
        torch.cuda.synchronize()
        g3 = model.engine.arena.grad_flat.clone()
        gstats = dict(gx.stats, segments=[sum(1 for i in e.gB if not isinstance(i, tuple)) for e in gx.entries.values()],
                      replay_err=float((g3 - g1).abs().max()), replay_loss=[float(lm3), float(img3), float(nsp3)])
    torch.save(dict(grad=g1.cpu(), acc=acc.cpu(), losses=[float(lm), float(img), float(nsp_l)], shard=(lo, hi),
                    flat=model.engine.arena.flat.detach().cpu(), seed=seed_after, stats=dp.comm_stats(), graphs=gstats), out)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
