"""Data-parallel path on CPU: 2 processes, gloo.  The wrapper (unimm_amd.parallel.DataParallelRCCL)
is device-agnostic, so the same bucketed all-reduce / flat-arena code that runs over RCCL on the GPUs
is exercised here with the CPU oracle as the module.

Checked: N-rank result == 1-rank result on the concatenated batch under the reference's semantics
(per-replica mean losses, then mean over replicas: train.py:164-166, utils/data_parallel.py:129),
parameter broadcast at construction, `no_sync()` accumulation, and the even shard split."""
import json
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn

from oracle import vilbert_ref as R

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class OracleNet(nn.Module):
    """The CPU oracle wrapped as an nn.Module (parameters named with '.' -> '__')."""

    def __init__(self, cfg, sd):
        super().__init__()
        self.cfg = cfg
        self.names = [k for k in sd if k != R.TIED[0]]
        for k in self.names:
            self.register_parameter(k.replace(".", "__"), nn.Parameter(sd[k].clone()))

    def forward(self, b):
        sd = {k: getattr(self, k.replace(".", "__")) for k in self.names}
        sd[R.TIED[0]] = sd[R.TIED[1]]
        out = R.forward(sd, self.cfg, b["input_ids"], b["image_feat"], b["image_loc"], token_type_ids=b["token_type_ids"],
                        position_ids=b["position_ids"], attention_mask=b["attention_mask"],
                        image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
                        masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
                        next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"])
        return out["lm_loss"] + out["img_loss"] + out["nsp_loss"]


def _load():
    cfg = R.make_config(json.load(open(os.path.join(GOLD, "small_config.json"))))
    g = np.load(os.path.join(GOLD, "small_mixed.npz"))
    batch = {k[4:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("in::")}
    return cfg, batch


def _shard(batch, lo, hi):
    return {k: (v if k == "nsp_weight" else v[lo:hi]) for k, v in batch.items()}


def _worker(rank, world, port, q, wire, algo):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    from unimm_amd.parallel import DataParallelRCCL, shard_range
    cfg, batch = _load()
    # different init per rank: the wrapper's broadcast must make the replicas identical to rank 0's
    net = OracleNet(cfg, R.init_state_dict(cfg, seed=11 + 5 * rank))
    dp = DataParallelRCCL(net, wire_dtype=wire, algorithm=algo)
    lo, hi = shard_range(batch["input_ids"].shape[0], rank, world)
    dp.arena.attach_grads()
    loss = dp(_shard(batch, lo, hi))
    loss.sum().backward()
    dp.sync_gradients()
    g1 = dp.arena.grad_flat.clone()
    # accumulation: a second micro-step under no_sync, then a synced one
    with dp.no_sync():
        dp(_shard(batch, lo, hi)).sum().backward()
        dp.sync_gradients()
    local_acc = dp.arena.grad_flat.clone()
    if rank == 0:
        q.put(dict(grad=g1.numpy(), acc=local_acc.numpy(), loss=float(loss.detach().sum()), flat=dp.arena.flat.detach().numpy().copy(),
                   shard=(lo, hi)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("wire,algo,tol", [("fp32", "allreduce", 1e-5), ("bf16", "rs_ag", 6e-3)])
def test_two_rank_gloo_matches_single_process(wire, algo, tol):
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q, wire, algo)) for r in range(2)]
    for p in procs:
        p.start()
    res = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0

    # single-process reference: mean over the two shards of the per-shard (local-mean) losses
    from unimm_amd.arena import FlatArena
    from unimm_amd.parallel import shard_range
    cfg, batch = _load()
    net = OracleNet(cfg, R.init_state_dict(cfg, seed=11))
    named = dict(net.named_parameters())
    arena = FlatArena(named, [("all", [(n, tuple(p.shape)) for n, p in named.items()])])
    assert np.array_equal(res["flat"], arena.flat.detach().numpy())        # rank 0's weights were broadcast
    arena.attach_grads()
    n = batch["input_ids"].shape[0]
    assert shard_range(n, 0, 2) == tuple(res["shard"]) == (0, 3) and shard_range(n, 1, 2) == (3, 6)
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    total = 0.5 * (net(_shard(batch, 0, 3)).sum() + net(_shard(batch, 3, 6)).sum())
    total.backward()
    want = arena.grad_flat.numpy().copy()
    scale = np.abs(want).max()
    assert np.abs(res["grad"] - want).max() <= tol * scale
    # no_sync: the second backward only accumulated rank 0's local gradient on top of the averaged one
    arena.zero_grads()
    net(_shard(batch, 0, 3)).sum().backward()
    local = arena.grad_flat.numpy().copy()
    assert np.abs(res["acc"] - (res["grad"] + local)).max() <= 1e-5 * scale
