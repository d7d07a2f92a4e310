#!/usr/bin/env python
"""Headline benchmark: dialog-sequences/sec of one UniMM-UL training step (forward + backward,
dropout on, all three losses) at batch 240 x 256 tokens x 37 regions on N MI355X.

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus N ...          (no launcher: starts the N ranks itself as child processes, see self_launch)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \\
           --master-port P bench.py --gpus N --steps K --warmup W

Prints ONE JSON line on rank 0 (contract: see the task statement).  A "step" = zero the gradient
arena, forward, the reference's loss combination, backward into the flat gradient arena, and (N > 1)
the bucketed RCCL gradient all-reduce.  N > 1 keeps the GLOBAL batch of 240 (BASELINE configs[2]: "same config
data-parallel over 8 x MI355X" = 30 sequences per GPU; `"scaling": "strong"`); `--scaling weak` runs 240 per GPU.  The engine runs the text stream on the valid token rows only
(padding rows are inert), so `roofline` prices the FLOPs actually executed
(2*M*N*K of every launch) -- the padded-equivalent figure is reported separately and never used for
`achieved`.  Inputs are synthetic (unimm_amd.synth) and resident in HBM
before the timed region; weights are random-init at the full bert_base_6layer_6conect.json config.
The optimizer step is outside the metric ("fwd+bwd", BASELINE.json) and is not run.

Extra blocks:
  roofline     - the dominant kernel (largest summed launch time among the GEMM variants), timed live
                 with HIP events on its launch stream over the timed region, priced in algorithmic
                 FLOPs (2*M*N*K per launch) against the dense bf16 MFMA peak.
  cpu_baseline - the CPU oracle (PyTorch fp32 restatement pinned to the reference's goldens) on a
                 bounded sample: BASELINE config 1 (1 image x 6 sequences), fwd+bwd, on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

GRAPHS_AUTO_MAX_SEQ = 90           # --graphs auto: replayed hipGraphs up to this many sequences per GPU (DESIGN.md thresholds table)
PEAK_BF16_TFLOPS = 2500.0          # dense bf16 MFMA peak, MI355X (MI355X_MICROARCH.md: ~2.5 PF dense)
F_FWD_BASE_GF = 76.808             # GFLOP / sequence forward without the vocab decoder
F_DEC_ROW_GF = 0.04688             # GFLOP per decoded row (768 x 30522)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--batch", type=int, default=240, help="sequences per step: the GLOBAL batch, split evenly over the ranks "
                                                           "(--scaling strong, default) or per GPU (--scaling weak)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="strong",
                    help="strong (default): BASELINE configs[1]/[2], batch_size=240 in total = 30 per GPU on 8 (SURVEY 8d/8e); "
                         "weak: 240 per GPU")
    ap.add_argument("--workload", choices=["train", "dense", "scoring"], default="train",
                    help="train = BASELINE configs[1]/[2] (the headline metric); dense = configs[3] micro-step "
                         "(100 sequences, nsp_loss_coeff 0, gradient accumulation: no exchange inside the step); "
                         "scoring = configs[4] (val_lm: 1 image = 10 rounds x 100 candidates in 4 chunks of 250, "
                         "forward + sequence log-likelihood + ranks)")
    ap.add_argument("--dense-objective", choices=["ranking", "proxy"], default="ranking",
                    help="dense workload: ranking = NeuralNDCG^T over the step's options + LM (+0 x NSP), the objective of "
                         "dense_annotation_finetuning.py:263-293 (gradient enters through the NSP scores); proxy = LM + region-KL only")
    ap.add_argument("--with-optimizer", action="store_true",
                    help="also run the fused AdamW step (train.py:322-348 grouping and schedule) inside the timed step; "
                         "NOT the headline metric, which is fwd+bwd (the config block says which was run)")
    ap.add_argument("--fused-step", choices=["on", "off"], default="off",
                    help="A/B: the step through `forward_backward(..., loss_weights)` (forward and backward enqueued back to back, no "
                         "autograd round trip between them) instead of `loss = combine(...); loss.backward()`: +0.2-1.2 %% at 30 "
                         "sequences per GPU, nothing at 60 (the host is ahead of the GPU either way)")
    ap.add_argument("--attn-order", choices=["on", "off"], default="on",
                    help="A/B: off = the attention launches take their (sequence, head) items in batch order (engine.attn_longest_first "
                         "= False) instead of longest sequence first")
    ap.add_argument("--order-by-length", choices=["off", "desc", "asc"], default="off",
                    help="experiment: permute the synthetic batch's sequences by valid length before the run (the same work; the "
                         "attention kernels take their (sequence, head) items in batch order)")
    ap.add_argument("--compact-inputs", action="store_true",
                    help="feed mask descriptors + per-image tensors + image_index (SURVEY 8 row F3) instead of the "
                         "reference-shaped dense masks and per-sequence image copies")
    ap.add_argument("--host-staging", choices=["on", "off"], default="on",
                    help="with --host-inputs direct: the engine's own pinned staging ring + copy stream for CPU tensors handed to "
                         "forward() (Engine.stage_host_inputs; default) or the plain .to(device) path (A/B)")
    ap.add_argument("--host-inputs", choices=["off", "direct", "prefetch"], default="off",
                    help="off (default, the contract's metric): the step's inputs are resident in HBM.  direct: every step takes a "
                         "fresh batch of CPU tensors handed straight to forward(), the reference's calling convention "
                         "(train.py:113-129) in the reference's layout (int64 dense masks, per-sequence image copies: ~270 MB "
                         "per 240 sequences; with --compact-inputs: descriptors + per-image tensors).  prefetch: the same "
                         "host batches through unimm_amd.inputs.DevicePrefetcher (pinned, copied one step ahead on a copy stream)")
    ap.add_argument("--gemm-profile", choices=["all", "dominant"], default="dominant",
                    help="HIP-event timing inside the timed region: dominant (default) = the weight-gradient kernel only, which is "
                         "what the roofline block needs; all = every GEMM launch (adds the all_gemm_* fields).  Two event records "
                         "around each of ~340 GEMM launches are not free: measured 1.3 ms of a 56 ms step at bs=240 (drain + "
                         "timestamp between back-to-back kernels) and 2 ms of 16 ms at 30 sequences per GPU (host launch rate)")
    ap.add_argument("--wgrad-rounds", type=int, default=0, metavar="R",
                    help="tuning: engine.wgrad_group_rounds (rounds of 256 tiles a grouped weight-gradient launch aims at; 0 = default)")
    ap.add_argument("--wgrad-stream", action="store_true",
                    help="A/B: the text side's grouped weight-gradient launches on a third stream (engine.wgrad_stream = True)")
    ap.add_argument("--single-stream", action="store_true",
                    help="everything on one HIP stream (engine.dual_stream = False): exclusive kernel durations for rocprofv3 "
                         "breakdowns; the production schedule runs the image side on a second stream")
    ap.add_argument("--decoder-dx-rows", type=int, default=-1,
                    help="A/B: decoded-row count up to which the decoder's input gradient runs as a split reduction over the "
                         "vocabulary (Engine._decoder_dx); 0 = always the NT GEMM; -1 = the engine's default")
    ap.add_argument("--image-head-main", action="store_true", help="A/B: image prediction head on the text stream (round-2 schedule)")
    ap.add_argument("--image-tile", type=int, default=-1, help="A/B: tile code of the image side's GEMMs at large batches "
                    "(0 = the kernel library's choice; -1 = the engine's default, the 256x256 ping-pong tile)")
    ap.add_argument("--graphs", choices=["auto", "on", "off"], default="auto",
                    help="run the step as replayed hipGraphs (unimm_amd/graphs.py: two graph launches per step instead of ~650 "
                         "host calls; auto = on for <= %d sequences per GPU, where the host would otherwise bound the " % GRAPHS_AUTO_MAX_SEQ +
                         "step). The timed region then has no per-launch events: the roofline block comes from eager steps after it.")
    ap.add_argument("--compute", choices=["bf16", "fp32x3"], default="bf16",
                    help="bf16 (default, the headline: the reference's autocast class) or fp32x3 = the fp32-accuracy engine "
                         "(unimm_amd/engine_x3.py: bf16 MFMA GEMMs over split operands hi/lo, fp32 attention and gradient "
                         "stream): the arithmetic dense_annotation_finetuning.py:253 runs in (no autocast)")
    ap.add_argument("--x3-attn", choices=["mfma", "valu"], default="mfma", help="A/B (fp32x3): attention cores on the fp32 matrix "
                    "instruction (default) or the vector-ALU kernels of the first version (unimm_x3_attn_set_impl)")
    ap.add_argument("--plain-loss", action="store_true", help="A/B: combine the three losses with the written-out torch arithmetic "
                    "(c * x.mean() + ...) instead of harness.combine_losses (one autograd node)")
    ap.add_argument("--no-splitk", action="store_true", help="A/B: engine.splitk = False (no split-K for the long reductions of small batches)")
    ap.add_argument("--shared-context", choices=["on", "off"], default="on",
                    help="scoring workload: compute the context rows and the image stream once per dialog round (default; bf16 engine) "
                         "or per candidate as the reference does")
    ap.add_argument("--scoring-chunk", type=int, default=250, metavar="N",
                    help="scoring workload: sequences per forward call (250 = the reference's val_lm.log; 1000 = one image per call)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--rank-probe", action="store_true",
                    help="launch check only: every rank joins a gloo group on the CPU, all-reduces a one, rank 0 prints what it saw")
    ap.add_argument("--gemm-tile", type=int, default=0, metavar="CODE",
                    help="tuning: unimm_gemm_nt_args.tile = CODE for every launch of the run (1000 x tile columns per group + 100 x {1 persistent, "
                         "2 one workgroup per tile} + tile configuration; 0 = the library's defaults)")
    ap.add_argument("--tile-table", default="", metavar="SPEC",
                    help="tuning: per-shape tile codes of the small-batch regime, e.g. t:3072:768=8,i:1024:1024=6 (side:N:K=code)")
    ap.add_argument("--host-profile", default=None, metavar="FILE",
                    help="after warm-up, cProfile 5 untimed steps of host-side enqueue work into FILE (text, by own time)")
    ap.add_argument("--no-padded", action="store_true", help="skip the 3 extra steps that time the padded schedule beside the default")
    ap.add_argument("--wire", choices=["fp32", "bf16"], default="fp32", help="N > 1: dtype of the gradient exchange")
    ap.add_argument("--exchange", choices=["allreduce", "rs_ag"], default="allreduce",
                    help="N > 1: all-reduce per bucket, or reduce-scatter + all-gather per bucket")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--config", default=os.path.join(ROOT, "unimm_amd", "config", "bert_base_6layer_6conect.json"))
    return ap.parse_args()


def host_cores():
    """Cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = os.cpu_count() or 1
    try:
        n = len(os.sched_getaffinity(0))
    except Exception:
        pass
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(int(txt[0]) / int(txt[1]))))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, q // per))
        except Exception:
            pass
    return n


def log(msg):
    print(f"[bench] {msg}", file=sys.stderr, flush=True)


def cpu_baseline(cfg_path, steps):
    """Oracle fwd+bwd at BASELINE config 1 on the host cores (bounded: 1 warm-up + `steps` timed)."""
    from oracle import vilbert_ref as R
    from unimm_amd import synth
    from unimm_amd.config import BertConfig
    cores = host_cores()
    torch.set_num_threads(cores)
    log(f"cpu baseline on {cores} host cores ...")
    ocfg = R.make_config(cfg_path)
    sd = R.init_state_dict(ocfg, seed=1, perturb=False)
    leaves = {k: v.requires_grad_(True) for k, v in sd.items() if k != R.TIED[0]}
    leaves[R.TIED[0]] = leaves[R.TIED[1]]
    b = synth.make_batch(n_seq=6, cfg=BertConfig.from_json_file(cfg_path), seed=99, mask_dtype=torch.int64)

    def step():
        for v in leaves.values():
            v.grad = None
        out = R.forward(leaves, ocfg, b["input_ids"], b["image_feat"], b["image_loc"], token_type_ids=b["token_type_ids"],
                        position_ids=b["token_position_ids"], attention_mask=b["attention_mask"],
                        image_attention_mask=b["image_attention_mask"], co_attention_mask=b["co_attention_mask"],
                        masked_lm_labels=b["masked_lm_labels"], image_label=b["image_label"], image_target=b["image_target"],
                        next_sentence_label=b["next_sentence_label"], nsp_weight=b["nsp_weight"], lm_weight=b["lm_weight"])
        (out["lm_loss"] + out["img_loss"] + out["nsp_loss"]).sum().backward()

    step()
    log("cpu baseline warm-up done")
    t0 = time.perf_counter()
    for i in range(steps):
        step()
        log(f"cpu baseline step {i + 1}/{steps}: {(time.perf_counter() - t0) / (i + 1):.2f} s/step")
    dt = (time.perf_counter() - t0) / steps
    return {"value": round(6.0 / dt, 4), "unit": "dialog-sequences/sec", "cores": cores, "kind": "port",
            "sample": f"BASELINE config 1: 1 image x 6 sequences x 256 tokens x 37 regions, fp32 eager oracle, fwd+bwd (dense "
                      f"decoder on all 256 rows as the reference computes it), 1 warm-up + {steps} timed steps, {dt:.2f} s/step"}


def scoring(args, world, rank, dev, enc, lib, synth, dist):
    """BASELINE configs[4]: val_lm.py generative scoring.  One step = one image: 10 rounds x 100 candidate answers =
    1000 sequences, run as 4 chunks of 250 (val_lm.py:104-136, chunk size from the reference's val_lm.log), each chunk a
    forward + per-sequence log-likelihood on the labelled rows, then ranks per round (utils/visdial_metrics.py:21-39).
    The 100 candidates of a round share image, history and question (dataloader/dataloader_visdial.py builds them so;
    synth.make_scoring_batch); with --shared-context on (default, bf16 engine) the context rows and the image stream are
    computed once per round and chunk (unimm_amd/scoring.py) -- the per-candidate schedule, which recomputes them 100 times as
    the reference does, is timed beside it (`config.per_candidate_value`).  Ranks process different images (weak scaling, no
    collective)."""
    from unimm_amd import harness
    model = enc.bert_pretrained
    enc.eval()
    cfg = model.config
    host = args.host_inputs == "direct"       # the chunks stay in HOST memory, in the reference's layout (int64 masks), and are handed to
    #                                           sequence_log_likelihood as val_lm.py:86-121 hands them over: staged inside the call
    img = synth.make_scoring_batch(rounds=10, options=100, cfg=cfg, seed=4321 + 16 * rank, device="cpu" if host else dev,
                                   **(dict(mask_dtype=torch.int64) if host else {}))
    spec = img.pop("mask_spec")
    if host and args.host_staging == "off":
        model.engine.host_staging = False
    ck = args.scoring_chunk
    if 1000 % ck:
        raise SystemExit("--scoring-chunk must divide 1000")
    chunks = [{k: v[ck * c:ck * (c + 1)] for k, v in img.items()} for c in range(1000 // ck)]
    if args.compact_inputs:                       # row F3: mask descriptors instead of dense masks, one image entry + an index
        from unimm_amd.inputs import DialogMaskSpec
        for c, b in enumerate(chunks):
            sl = slice(ck * c, ck * (c + 1))
            b["attention_mask"] = DialogMaskSpec(spec.mode[sl], spec.length[sl], spec.answer[sl])
            b["co_attention_mask"] = None
            b["image_feat"], b["image_loc"] = b["image_feat"][:1].contiguous(), b["image_loc"][:1].contiguous()
            b["image_index"] = torch.zeros(ck, dtype=torch.int64, device="cpu" if host else dev)
    n_rows = int((img["masked_lm_labels"] != -1).sum())
    shared = args.shared_context == "on" and args.compute == "bf16"

    def step(shared=shared):
        sc = []
        for b in chunks:
            s, _ = model.sequence_log_likelihood(b["input_ids"], b["image_feat"], b["image_loc"], b["masked_lm_labels"],
                                                 token_type_ids=b["token_type_ids"], position_ids=b["token_position_ids"],
                                                 attention_mask=b["attention_mask"], image_attention_mask=b["image_attention_mask"],
                                                 co_attention_mask=b["co_attention_mask"], image_index=b.get("image_index"),
                                                 shared_context=b["context_group"] if shared else None)
            sc.append(s)
        return harness.scores_to_ranks(torch.cat(sc).view(1, 10, 100))

    log(f"scoring: {len(chunks)} chunks x {ck} sequences ready on {dev}, {n_rows} decoded rows per image, shared context {'on' if shared else 'off'}")
    for _ in range(max(args.warmup, 2) if host else args.warmup):
        step()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    lib.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ranks = step()
    fence()
    dt = time.perf_counter() - t0
    prof = lib.prof_collect()
    lib.prof_enable(False)
    per_cand = None
    if shared:                                    # the reference's schedule (every candidate recomputes context and image), same inputs
        step(False)
        fence()
        t1 = time.perf_counter()
        for _ in range(3):
            ranks_pc = step(False)
        fence()
        per_cand = 1000 * 3 / (time.perf_counter() - t1)
        rank_agree = float((ranks_pc == ranks).float().mean())
    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    value = 1000 * world * args.steps / dt
    if rank == 0:
        name, (ms, fl, cnt) = max(prof.items(), key=lambda kv: kv[1][0])
        gemm_ms = sum(v[0] for v in prof.values())
        gemm_fl = sum(v[1] for v in prof.values())
        achieved = fl / (ms * 1e-3) / 1e12
        f_fwd = F_FWD_BASE_GF + F_DEC_ROW_GF * n_rows / 1000
        out = {"metric": "candidate dialog-sequences/sec (forward + sequence log-likelihood + ranks) at chunk=250 seq=256 regions=36(+1 <IMG>)",
               "value": round(value, 2), "unit": "dialog-sequences/sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
               "dtype": "bf16", "data": "synthetic",
               "config": {"workload": "val_lm generative scoring (BASELINE configs[4]): 1 image = 10 rounds x 100 candidates per step (the "
                                      "candidates of a round share image and dialog context), 4 chunks of 250 sequences, forward only, "
                                      "row-sparse decoder on the answer-copy rows, per-sequence log-likelihood, ranks per round",
                          "schedule": ("shared context: context rows + image stream once per round and chunk, candidate rows per sequence "
                                       "(unimm_amd/scoring.py)" if shared else "per candidate (the reference's schedule: everything per sequence)"),
                          "per_candidate_value": round(per_cand, 2) if per_cand is not None else None,
                          "ranks_equal_to_per_candidate_schedule": round(rank_agree, 4) if per_cand is not None else None,
                          "global_batch": 1000 * world, "per_gpu_batch": 1000, "chunk": args.scoring_chunk,
                          "inputs": ("mask descriptors + one image entry + image_index (row F3)" if args.compact_inputs else "reference layout (dense masks, per-sequence image copies)")
                                    + ("; chunks start in HOST memory every call (CPU tensors handed to sequence_log_likelihood) -- NOT the contract's resident-input metric" if host else ""),
                          "seq_len": 256, "regions": 37,
                          "parallelism": f"dp{world}", "lm_rows_decoded_per_seq": round(n_rows / 1000, 2),
                          "gflop_per_seq_fwd_padded_equivalent": round(f_fwd, 3),
                          "gemm_gflop_per_seq_executed": round(gemm_fl / args.steps / 1000 / 1e9, 3),
                          "rank_checksum": int(ranks.sum())},
               "roofline": {"bound": "mfma", "kernel": name, "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                            "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": None,
                            "launches_per_step": cnt // args.steps, "avg_launch_us": round(ms * 1e3 / cnt, 2),
                            "all_gemm_tflops": round(gemm_fl / (gemm_ms * 1e-3) / 1e12, 1),
                            "gemm_share_of_step": round(gemm_ms * 1e-3 / dt, 3),
                            "padded_equivalent_tflops": round(f_fwd * 1e9 * value / 1e12, 1)}}
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


def self_launch(n):
    """`python bench.py --gpus N` run as a plain process (RANK / WORLD_SIZE unset): spawn the ranks the way the external launcher
    does -- python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port <free port>
    bench.py <the same arguments> -- as a child process, relay rank 0's JSON line(s) on stdout (everything else the ranks print
    goes to stderr) and return the child's exit code.  The parent never initialises the GPU and never exec()s (a process that
    has touched the device must not be replaced on this pool; this one has not, but a child is the form that is always allowed).
    Replaces the single-process fan-out of utils/data_parallel.py:120-129 at the command line."""
    import socket
    import subprocess
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")          # dmabuf IPC: RCCL across processes needs it on this image
    env.setdefault("OMP_NUM_THREADS", str(max(1, host_cores() // n)))
    env["UNIMM_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    child = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    for line in child.stdout:
        if line.lstrip().startswith("{"):
            sys.stdout.write(line)
            sys.stdout.flush()
        else:
            sys.stderr.write(line)
    return child.wait()


def rank_probe(world, rank, batch):
    """--rank-probe: the launch path without the engine -- every rank joins a gloo group (no GPU needed), all-reduces a one, and
    rank 0 prints a JSON line with what it saw.  What tests/test_bench_launch_cpu.py runs on the CPU-only build box."""
    import torch.distributed as dist
    if batch < world:                                          # the same refusal as the real path (shard_range below)
        raise SystemExit(f"--batch {batch} leaves rank {world - 1} of {world} without a sequence")
    seen = 1
    if world > 1:
        dist.init_process_group("gloo")
        ones = torch.ones(1)
        dist.all_reduce(ones)
        seen = int(ones.item())
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps({"probe": True, "n_gpus": world, "comm": {"backend": "gloo", "rccl_ranks": seen},
                          "launcher": "self" if os.environ.get("TORCHELASTIC_RUN_ID") is not None and os.environ.get("UNIMM_SELF_LAUNCHED") else "external"}),
              flush=True)
    if seen != world:
        raise SystemExit(f"process group reduced ones to {seen}, expected {world}")


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves as CHILD processes (this process has not
        # touched the GPU and never will) and exit with their return code
        raise SystemExit(self_launch(args.gpus))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch one rank per GPU (torch.distributed.run --nproc-per-node {args.gpus}), "
                         "or run `python bench.py --gpus N` without a launcher")
    if args.rank_probe:
        return rank_probe(world, rank, args.batch)
    # Host thread pools sized for the CPU share this process really has (affinity capped by the cgroup quota: 16 of the box's
    # 256 hardware threads per GPU).  torch's default is one intra-op thread per hardware thread (128 here): one parallel CPU op
    # then burns the cgroup's quota for the scheduling period and EVERY thread of the process stalls until the next one
    # (cpu.stat: seconds of throttled time per run) -- seen as 20 ms host copies in host-fed steps.
    if "OMP_NUM_THREADS" not in os.environ:
        torch.set_num_threads(max(1, min(host_cores() // max(1, world), 16)))
    # Rehearsal of the N > 1 path on a one-GPU box: UNIMM_BENCH_REHEARSAL=1 puts every rank on device 0 and
    # exchanges gradients over gloo (RCCL refuses two ranks on one device).  Never set by the driver.
    rehearsal = os.environ.get("UNIMM_BENCH_REHEARSAL", "0") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    import torch.distributed as dist
    if world > 1:
        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    rccl_ranks = 1
    if world > 1:                                 # "did the backend see N ranks": an all-reduce of ones, reported in `comm`
        ones = torch.ones(1, device=dev)
        dist.all_reduce(ones)
        rccl_ranks = int(ones.item())
        if rccl_ranks != world:
            raise SystemExit(f"process group reduced ones to {rccl_ranks}, expected {world}")

    from unimm_amd import VisualDialogEncoder, harness, lib, synth
    from unimm_amd.parallel import DataParallelRCCL

    torch.manual_seed(1234)                       # identical init on every rank (+ broadcast in the wrapper)
    enc = VisualDialogEncoder(args.config, compute_dtype=args.compute).to(dev)
    if args.compute != "bf16":
        from unimm_amd import lib as _lib
        _lib.x3_attn_set_impl(1 if args.x3_attn == "mfma" else 0)
        enc.bert_pretrained.engine.attn_planes = args.x3_attn == "mfma"
    enc.train()
    model = enc.bert_pretrained
    model.set_dropout_seed(1234 + rank)
    net = DataParallelRCCL(enc, device=dev, wire_dtype=args.wire, algorithm=args.exchange) if world > 1 else enc
    cfg = model.config

    if args.workload == "scoring":
        return scoring(args, world, rank, dev, enc, lib, synth, dist)
    if args.workload == "dense" and args.batch == 240:
        args.batch = 100                          # dense_annotation_finetuning.py: 1 image x 100 options per micro-step
    if args.scaling == "weak":
        per_gpu, global_batch = args.batch, args.batch * world
    else:                                         # even split of the global batch (unimm_amd.parallel.shard_range)
        from unimm_amd.parallel import shard_range
        lo, hi = shard_range(args.batch, rank, world)
        per_gpu, global_batch = hi - lo, args.batch
        if per_gpu < 1:
            raise SystemExit(f"--batch {args.batch} leaves rank {rank} of {world} without a sequence")
    if args.workload == "dense":                  # discriminative inputs, 2 sequences share an image (configs[3])
        batch = synth.make_batch(n_seq=per_gpu, cfg=cfg, seed=1234 + rank, device=dev, modes=["dis"] * per_gpu,
                                 sequences_per_image=2)
        coeff = dict(lm=1.0, nsp=0.0, img=1.0)
        g = torch.Generator().manual_seed(77 + rank)
        relevance = torch.tensor([0, 0, 0, 0, 0.2, 0.4, 0.6, 1.0])[torch.randint(0, 8, (1, per_gpu), generator=g)].to(dev)
    else:
        batch = synth.make_batch(n_seq=per_gpu, cfg=cfg, seed=1234 + rank, device=dev, compact=args.compact_inputs)
        coeff = dict(lm=1.0, nsp=1.0, img=1.0)    # options.py:68-70 defaults
    if args.order_by_length != "off" and torch.is_tensor(batch.get("attention_mask")):
        am = batch["attention_mask"]
        lens = am.ne(0).any(-1).sum(1) if am.dim() == 3 else am.ne(0).sum(1)
        perm = torch.argsort(lens, descending=args.order_by_length == "desc", stable=True)
        nb = am.shape[0]
        batch = {k: (v.index_select(0, perm.to(v.device)).contiguous() if torch.is_tensor(v) and v.dim() >= 1 and v.shape[0] == nb else v)
                 for k, v in batch.items()}
    nsp_w = batch.pop("nsp_weight")
    n_lm_rows = int((batch["lm_weight"] != 0).sum())
    feed = None
    h2d_bytes = None
    if args.host_inputs != "off":
        if args.workload != "train":
            raise SystemExit("--host-inputs prefetch is measured on the train workload")
        import itertools
        hbs = []
        for j in range(3):                          # three different host batches, cycled (other lengths, other row counts)
            hb = synth.make_batch(n_seq=per_gpu, cfg=cfg, seed=1234 + rank + 1000 * j, device="cpu", compact=args.compact_inputs,
                                  mask_dtype=torch.int64)        # the reference's discriminative masks are int64 (utils/data_utils.py:300)
            hb.pop("nsp_weight")
            if args.compact_inputs:                 # what a compact loader ships: no dense masks, no per-sequence image copies
                for k in ("attention_mask", "co_attention_mask", "image_feat", "image_loc", "image_target"):
                    hb.pop(k)
            hbs.append(hb)
        h2d_bytes = sum(v.numel() * v.element_size() for v in hbs[0].values() if torch.is_tensor(v))
        if args.host_inputs == "prefetch":
            from unimm_amd.inputs import DevicePrefetcher
            src = DevicePrefetcher(itertools.cycle(hbs), dev, cache_pinned=True)   # a cycled list of immutable batches: pinned once
        else:
            src = itertools.cycle(hbs)

        def feed():
            batch.update(next(src))

    fused_step = [False]                        # set below, once it is known whether the graph executor runs the step

    def combine(lm, nsp, img):                  # train.py:164-168, as unimm_amd.harness.forward combines them
        if args.plain_loss:
            return coeff["lm"] * lm.mean() + coeff["nsp"] * nsp.mean() + coeff["img"] * img.mean()
        return harness.combine_losses(lm, nsp, img, coeff["lm"], coeff["nsp"], coeff["img"])

    def fwd_bwd():
        if args.workload == "dense" and args.dense_objective == "ranking":
            from unimm_amd import ranking
            lm, img, nsp, nsp_scores = net(
                batch["input_ids"], batch["image_feat"], batch["image_loc"], sep_indices=batch["sep_indices"],
                sep_len=batch["sep_len"], token_type_ids=batch["token_type_ids"], token_position_ids=batch["token_position_ids"],
                attention_mask=batch["attention_mask"], masked_lm_labels=batch["masked_lm_labels"],
                next_sentence_label=batch["next_sentence_label"], image_attention_mask=batch["image_attention_mask"],
                co_attention_mask=batch["co_attention_mask"], image_label=batch["image_label"], image_target=batch["image_target"],
                nsp_weight=nsp_w, lm_weight=batch["lm_weight"], output_nsp_scores=True)
            loss, _ = ranking.dense_finetune_loss(nsp_scores, batch["next_sentence_label"], relevance, lm, coeff["nsp"],
                                                  num_options=per_gpu)
            loss.backward()
            return loss
        if args.compact_inputs and args.workload == "train":
            lm, img, nsp = net(batch["input_ids"], batch["image_feat_unique"], batch["image_loc_unique"],
                               sep_indices=batch["sep_indices"], sep_len=batch["sep_len"], token_type_ids=batch["token_type_ids"],
                               token_position_ids=batch["token_position_ids"], attention_mask=batch["mask_spec"],
                               masked_lm_labels=batch["masked_lm_labels"], next_sentence_label=batch["next_sentence_label"],
                               image_attention_mask=batch["image_attention_mask"], image_label=batch["image_label"],
                               image_target=batch["image_target_unique"], nsp_weight=nsp_w, lm_weight=batch["lm_weight"],
                               image_index=batch["image_index"])
            loss = combine(lm, nsp, img)
            loss.backward()
            return loss
        if fused_step[0]:
            return net.forward_backward(
                batch["input_ids"], batch["image_feat"], batch["image_loc"], (coeff["lm"], coeff["nsp"], coeff["img"]),
                sep_indices=batch["sep_indices"], sep_len=batch["sep_len"], token_type_ids=batch["token_type_ids"],
                token_position_ids=batch["token_position_ids"], attention_mask=batch["attention_mask"],
                masked_lm_labels=batch["masked_lm_labels"], next_sentence_label=batch["next_sentence_label"],
                image_attention_mask=batch["image_attention_mask"], co_attention_mask=batch["co_attention_mask"],
                image_label=batch["image_label"], image_target=batch["image_target"], nsp_weight=nsp_w,
                lm_weight=batch["lm_weight"])[0]
        lm, img, nsp = net(batch["input_ids"], batch["image_feat"], batch["image_loc"], sep_indices=batch["sep_indices"],
                           sep_len=batch["sep_len"], token_type_ids=batch["token_type_ids"],
                           token_position_ids=batch["token_position_ids"], attention_mask=batch["attention_mask"],
                           masked_lm_labels=batch["masked_lm_labels"], next_sentence_label=batch["next_sentence_label"],
                           image_attention_mask=batch["image_attention_mask"], co_attention_mask=batch["co_attention_mask"],
                           image_label=batch["image_label"], image_target=batch["image_target"], nsp_weight=nsp_w,
                           lm_weight=batch["lm_weight"])
        loss = combine(lm, nsp, img)
        loss.backward()
        return loss

    opt = sched = None
    if args.with_optimizer:
        from unimm_amd.optim import FusedAdamW, WarmupLinearScheduleNonZero, default_language_weights, reference_param_groups
        groups = reference_param_groups(enc, lr=2e-5, image_lr=2e-5, language_weights=default_language_weights(enc))
        opt = FusedAdamW(groups, model.engine, lr=2e-5)
        sched = WarmupLinearScheduleNonZero(opt, warmup_steps=10000, t_total=200000)

    micro = [0]

    def step():
        loss = step_fb()
        if opt is not None:
            if world > 1:
                net.sync_gradients()        # no-op when every bucket was reduced during backward
            opt.step()
            sched.step()
        return loss

    def step_fb():
        if feed is not None:
            feed()
        if args.workload == "dense":
            # batch_multiply = 16 (dense_annotation_finetuning.py:299): gradients accumulate over 16 micro-steps
            # and are exchanged on the 16th only; one bench step = one micro-step.
            micro[0] += 1
            last = micro[0] % 16 == 0
            if micro[0] % 16 == 1:
                model.engine.arena.zero_grads()
            if world > 1 and not last:
                with net.no_sync():
                    return fwd_bwd()
            return fwd_bwd()
        model.engine.arena.zero_grads()
        return fwd_bwd()

    model.engine.gemm_tile = args.gemm_tile      # per-call tuning code of every unimm_gemm_nt launch (0 = automatic)
    if args.tile_table:
        for ent in args.tile_table.split(","):
            k, code = ent.split("=")
            side, n_, k_ = k.split(":")
            model.engine.tile_table[(side, int(n_), int(k_))] = int(code)
    if args.no_splitk:
        model.engine.splitk = False
    if args.host_staging == "off":
        model.engine.host_staging = False
    if args.single_stream:
        model.engine.dual_stream = False
    if args.wgrad_stream:
        model.engine.wgrad_stream = True
    if args.attn_order == "off":
        model.engine.attn_longest_first = False
    if args.wgrad_rounds > 0:
        model.engine.wgrad_group_rounds = args.wgrad_rounds
    if args.image_head_main:
        model.engine.image_head_side = False
    if args.image_tile >= 0:
        model.engine.image_tile = args.image_tile
    if args.decoder_dx_rows >= 0:
        model.engine.skinny_dx_rows = args.decoder_dx_rows
    model.engine.ensure(dev)
    model.engine.arena.attach_grads()
    log(f"model + batch ready on {dev}: {per_gpu} sequences/GPU, {n_lm_rows} decoded MLM rows")
    n_warm = args.warmup
    if args.host_inputs == "direct" and args.host_staging == "on":
        n_warm = max(n_warm, 4)                 # every set of the pinned staging ring (3) is allocated before the timed region
    for _ in range(n_warm):
        step()
    log("warm-up done")
    if args.host_profile and rank == 0:
        import cProfile, io, pstats
        torch.cuda.synchronize()
        pr = cProfile.Profile()
        # backward in the calling thread, so that the profile sees the engine's backward too
        with torch.autograd.set_multithreading_enabled(False):
            pr.enable()
            for _ in range(5):
                step()
            pr.disable()
        torch.cuda.synchronize()
        buf = io.StringIO()
        pstats.Stats(pr, stream=buf).sort_stats("tottime").print_stats(60)
        open(args.host_profile, "w").write(buf.getvalue())

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    use_graphs = (args.graphs == "on" or (args.graphs == "auto" and per_gpu <= GRAPHS_AUTO_MAX_SEQ)) and args.workload in ("train", "dense") \
        and not args.compact_inputs and not args.host_profile and (args.host_inputs == "off" or (args.host_inputs == "direct" and args.host_staging == "on"))
    gx = None
    fused_step[0] = args.workload == "train" and not args.compact_inputs and \
        args.fused_step == "on"
    if use_graphs:
        gx = model.engine.enable_graphs(True)
        done = 0
        try:
            n_pre = 9 if feed is not None else 3    # eager once more, capture, first replay -- of EVERY signature (three host batches are cycled)
            for _ in range(n_pre):
                step()
                done += 1
            torch.cuda.synchronize()
        except Exception as e:                  # never lose a run to the executor: the eager path computes the same step
            log(f"graph executor unavailable ({type(e).__name__}: {e}); continuing with eager launches")
            model.engine.enable_graphs(False)
            use_graphs, gx = False, None
            for _ in range(n_pre - done):       # the other ranks' collectives of these steps still need their partners
                step()
            torch.cuda.synchronize()
    if world > 1:                               # every rank runs the same executor (their collectives pair up either way)
        flag = torch.tensor([1 if use_graphs else 0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if use_graphs and int(flag) == 0:
            model.engine.enable_graphs(False)
            use_graphs, gx = False, None
    fence()
    if world > 1:
        net.comm_stats(reset=True)
    prof_all = args.gemm_profile == "all"
    if not use_graphs:
        lib.prof_enable(1 if prof_all else 2)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    if world > 1:          # every gradient bucket was exchanged exactly once per timed step
        st_timed = net.comm_stats()
        assert st_timed["buckets"] == st_timed["n_buckets_expected"] * args.steps, st_timed
    graph_stats = dict(model.engine.graphs.stats) if use_graphs else None
    prof_steps = args.steps                    # steps the launch profile `prof` covers
    if use_graphs:                             # the per-kernel figures below come from eager steps
        model.engine.graphs = None
        lib.prof_enable(2)
        step(); step()
        torch.cuda.synchronize()
        prof_steps = 2
    prof = lib.prof_collect()
    lib.prof_enable(False)
    loss_val = float(loss.detach())
    log(f"timed region: {dt / args.steps * 1e3:.2f} ms/step")
    if getattr(model.engine, "_stager", None) is not None:
        st_ = model.engine._stager.stats
        log(f"host staging: {st_['steps']} steps, {st_.get('host_ms', 0.0) / max(1, st_['steps']):.2f} ms of host time per step, "
            f"{st_['bytes_h2d'] / max(1, st_['steps']) / 1e6:.1f} MB per step over PCIe, pinned {model.engine._stager.pinned_bytes() / 1e6:.0f} MB; "
            + ", ".join(f"{k} {st_.get(k, 0.0) / max(1, st_['steps']):.2f}" for k in ("wait_ms", "pack_ms", "copy_ms")))
    # Outside the timed region (N=1 only): the same kernel with the chip to itself.  In the production schedule the image
    # side runs on its own stream, so a weight-gradient launch shares CUs with image-layer kernels and its in-situ
    # duration (the `achieved` / `frac` above) is not an exclusive one; two single-stream steps give that figure.
    exclusive = None
    if world == 1 and model.engine.dual_stream:
        model.engine.dual_stream = False
        step()
        torch.cuda.synchronize()
        lib.prof_enable(2)
        step(); step()
        torch.cuda.synchronize()
        ex = lib.prof_collect()
        lib.prof_enable(False)
        model.engine.dual_stream = True
        if "gemm_tn_pp" in ex:
            ems, efl, ecnt = ex["gemm_tn_pp"]
            exclusive = dict(achieved=round(efl / (ems * 1e-3) / 1e12, 1), avg_launch_us=round(ems * 1e3 / ecnt, 2),
                             launches_per_step=ecnt // 2, frac=round(efl / (ems * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4),
                             note="2 extra steps after the timed region with everything on one stream")

    # Outside the timed region, every rank: two more steps of the production schedule with EVERY GEMM launch bracketed by
    # events.  They give (a) the GEMM FLOPs one step really executes (2*M*N*K of every launch with its real M: the engine
    # runs the text stream on the valid rows only) and (b) the ranking of the GEMM kernels by time, so that
    # `roofline.kernel` is the measured dominant kernel and not a presumed one.  (Timing all ~340 GEMM launches inside
    # the timed region costs 1.3 ms per step in event records; --gemm-profile all does that on request.)
    def coattention(nsteps):
        """GEMM launches of the six connection layers (models/vilbert_dialog.py:655-783: the Q/K/V projections of both
        directions, BertBiOutput, both FFNs; forward, input gradients and their share -- by FLOPs -- of the grouped
        weight-gradient launches) over `nsteps` profiled steps: the figure north_star's ">= 40 % MFMA utilisation on the
        co-attention GEMMs" is judged by (the attention cores themselves are 0.5 % of the FLOPs)."""
        pc = model.engine.prof_conn = {"in": False, "tn_conn": 0.0, "tn_other": 0.0}
        lib.prof_enable(1)
        for _ in range(nsteps):
            step()
        torch.cuda.synchronize()
        allv = lib.prof_collect()
        tag = lib.prof_tagged()[1]
        lib.prof_enable(False)
        model.engine.prof_conn = None
        tn_ms = sum(v[0] for k, v in allv.items() if k.startswith("gemm_tn"))
        tn_fl = sum(v[1] for k, v in allv.items() if k.startswith("gemm_tn"))
        share = pc["tn_conn"] / max(pc["tn_conn"] + pc["tn_other"], 1.0)
        fl = (tag[1] + share * tn_fl) / nsteps
        ms = (tag[0] + share * tn_ms) / nsteps
        nt_tf = tag[1] / (tag[0] * 1e-3) / 1e12 if tag[0] > 0 else 0.0
        un = (tag[3] + share * tn_ms) / nsteps       # wall time with >= 1 connection-layer GEMM executing (+ the weight-gradient share, taken as not overlapping)
        return allv, {"gflop_per_step": round(fl / 1e9, 1), "ms_per_step": round(ms, 3),
                      "wall_ms_per_step_with_a_coattention_gemm_running": round(un, 3),
                      "frac_over_that_wall_time": round(fl / (un * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if un > 0 else None,
                      "tflops": round(fl / (ms * 1e-3) / 1e12, 1) if ms > 0 else None,
                      "frac": round(fl / (ms * 1e-3) / 1e12 / PEAK_BF16_TFLOPS, 4) if ms > 0 else None,
                      "fwd_and_dgrad": {"launches_per_step": tag[2] // nsteps, "ms_per_step": round(tag[0] / nsteps, 3),
                                        "tflops": round(nt_tf, 1), "frac": round(nt_tf / PEAK_BF16_TFLOPS, 4)},
                      "wgrad_share_of_grouped_launches": round(share, 4)}

    ex_all, coatt = coattention(2)
    coatt["schedule"] = "two streams: in-situ durations (the image half runs beside the text half)" if model.engine.dual_stream else "single stream"
    if world == 1 and model.engine.dual_stream:          # ... and with the chip to itself (exclusive durations)
        model.engine.dual_stream = False
        step()
        _, coatt_ex = coattention(2)
        model.engine.dual_stream = True
        coatt_ex["schedule"] = "single stream: exclusive durations"
        coatt = dict(coatt_ex, in_situ=coatt)
    exec_fl_step = sum(v[1] for v in ex_all.values()) / 2
    ranking = sorted(((k, v[0] / 2, v[1] / 2, v[2] // 2) for k, v in ex_all.items()), key=lambda r: -r[1])

    # the padded schedule (every sequence computed on all 256 token rows, as the reference does) beside the default one
    padded_value = None
    plan_default = model.engine.last_plan          # (the padded steps below replace engine.last_plan)
    if world == 1 and args.workload == "train" and not args.no_padded:
        model.engine.unpad = False
        step()
        torch.cuda.synchronize()
        tp = time.perf_counter()
        step(); step()
        torch.cuda.synchronize()
        padded_value = 2 * per_gpu / (time.perf_counter() - tp)
        model.engine.unpad = True

    # data-parallel exchange (N > 1): bytes per bucket, the exchange alone against the xGMI budget, and how much of it
    # the step does not hide (same step with and without the exchange: gradients stay local under no_sync)
    comm = None
    if world > 1:
        st = net.comm_stats(reset=True)
        flat = model.engine.arena.grad_flat
        fence()
        tc = time.perf_counter()
        for _ in range(3):
            net._exchange(0, flat.numel())
        fence()
        t_ar = (time.perf_counter() - tc) / 3
        model.engine.graphs = gx                # the executor of the timed region (None = eager launches)
        with net.no_sync():
            step_fb()
        fence()
        tc = time.perf_counter()
        with net.no_sync():
            for _ in range(2):
                step_fb()
        fence()
        t_nosync = (time.perf_counter() - tc) / 2
        model.engine.graphs = None
        wire = flat.numel() * (2 if net.wire_dtype == "bf16" else 4)
        busbw = wire * 2 * (world - 1) / world / t_ar / 1e9
        comm = {"backend": dist.get_backend(), "rccl_ranks": rccl_ranks, "world_size": dist.get_world_size(),
                "wire_dtype": net.wire_dtype, "algorithm": net.algorithm, "bytes_per_step": wire,
                "bucket_bytes": st["bucket_bytes"], "buckets_per_step": len(st["bucket_bytes"]),
                "collectives_per_step": st_timed["calls"] // max(1, args.steps),
                "exchange_alone_ms": round(t_ar * 1e3, 3), "busbw_GBps": round(busbw, 1),
                "xgmi_budget_GBps": 7 * 153, "busbw_frac_of_xgmi": round(busbw / (7 * 153), 3),
                "step_ms_without_exchange": round(t_nosync * 1e3, 3),
                "exposed_exchange_ms": round(max(0.0, dt / args.steps - t_nosync) * 1e3, 3),
                "note": "exchange alone = 3 all-reduces of the whole gradient arena back to back; exposed = timed step - the same "
                        "step under no_sync (this rank's clock)"}

    t = torch.tensor([dt], device=dev, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t)
    total_seq = global_batch * args.steps
    value = total_seq / dt

    if rank == 0:
        dom = max(prof.items(), key=lambda kv: kv[1][0])
        name, (ms, fl, cnt) = dom
        achieved = fl / (ms * 1e-3) / 1e12
        gemm_ms = sum(v[0] for v in prof.values())
        gemm_fl = sum(v[1] for v in prof.values())
        measured_dominant = ranking[0][0] if ranking else name
        if measured_dominant != name and not prof_all:
            # the kernel timed in situ is not the one that takes the most time: report the measured dominant kernel
            # from the two extra steps instead (same schedule, outside the timed region)
            k, kms, kfl, kcnt = ranking[0]
            name, ms, fl, cnt = k, kms * prof_steps, kfl * prof_steps, kcnt * prof_steps
            achieved = fl / (ms * 1e-3) / 1e12
        traffic, traffic_src, traffic_commit = None, None, None
        tpath = os.path.join(ROOT, "profiles", "traffic_dominant_kernel.json")
        if os.path.exists(tpath) and args.workload == "train" and per_gpu == 240:   # HBM bytes per launch from the committed rocprofv3 --pmc passes
            tj = json.load(open(tpath))    # (PMC counters cannot be read from inside the process)
            import hashlib
            sha = hashlib.sha256(open(os.path.join(ROOT, "unimm_amd", "csrc", "gemm.hip"), "rb").read()).hexdigest()
            if tj.get("kernel") == name and tj.get("gemm_hip_sha256") == sha:   # a PMC figure of ANOTHER kernel source is stale: null
                traffic, traffic_src = round(tj["bytes_per_launch_corrected"]), tj["source"]
                traffic_commit = tj.get("commit")      # the commit the PMC passes were taken at
            elif tj.get("kernel") == name:
                traffic_src = "stale: csrc/gemm.hip changed since " + str(tj.get("commit")) + " (re-run tools/profile_round.sh)"
        f_fwd = F_FWD_BASE_GF + F_DEC_ROW_GF * n_lm_rows / per_gpu     # reference-equivalent (padded to 256 tokens)
        plan = plan_default
        valid_rows = plan["Mv"] if plan is not None else per_gpu * 256
        exec_gf_seq = exec_fl_step / per_gpu / 1e9          # GEMM FLOPs actually executed per sequence, fwd+bwd
        if args.workload == "dense":
            metric = f"dialog-sequences/sec (fwd+bwd) dense-annotation fine-tune micro-step at bs={per_gpu} seq=256 regions=36(+1 <IMG>)"
            wl = ("dense-annotation fine-tune micro-step (BASELINE configs[3]): bert_base_6layer_6conect, discriminative inputs, "
                  "sequences_per_image=2, nsp_loss_coeff=0, batch_multiply=16 (gradient exchange every 16th step), " + args.compute + ", "
                  "fwd+bwd, " + ("objective = NeuralNDCG^T over the step's options + LM loss (dense_annotation_finetuning.py:263-293), "
                              "optimizer not included" if args.dense_objective == "ranking"
                              else "LM + region-KL proxy objective, ranking loss and optimizer not included"))
        else:
            metric = "dialog-sequences/sec (fwd+bwd) at bs=240 seq=256 regions=36(+1 <IMG>)"
            wl = ("UniMM-UL sparse training step (BASELINE configs[1]): bert_base_6layer_6conect, "
                  "sequences_per_image=6, num_negative_samples=5, mask_prob=0.15, dropout on, "
                  "MLM+UL / NSP / region-KL losses, fwd+bwd, " +
                  ("PLUS the fused AdamW step and weight-copy refresh (--with-optimizer)" if args.with_optimizer
                   else "optimizer step not included") +
                  ("; compact inputs (mask descriptors, per-image tensors)" if args.compact_inputs else "") +
                  ("" if args.host_inputs == "off" else
                   f"; inputs start in HOST memory every step ({args.host_inputs}: "
                   + ("CPU tensors handed to forward()" + ("" if args.host_staging == "on" else ", plain .to(device) path") if args.host_inputs == "direct" else "pinned + copied one step ahead by DevicePrefetcher")
                   + f", {h2d_bytes / 1e6:.0f} MB per step, 3 batches cycled) -- NOT the contract's resident-input metric"))
        out = {
            "metric": metric,
            "value": round(value, 2), "unit": "dialog-sequences/sec", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True,
            "scaling": args.scaling, "vs_baseline": None,
            "dtype": "bf16" if args.compute == "bf16" else "fp32x3 (bf16 MFMA over split operands hi+lo, fp32 accumulate; fp32 attention / activations / gradients)",
            "data": "synthetic",
            "config": {"workload": wl + ("" if args.compute == "bf16" else "; compute_dtype=fp32x3 (the reference's no-autocast arithmetic class)"),
                       "global_batch": global_batch, "per_gpu_batch": per_gpu, "seq_len": 256, "regions": 37,
                       "parallelism": f"dp{world}", "lm_rows_decoded_per_seq": round(n_lm_rows / per_gpu, 2),
                       "executor": ("hipGraph replay (unimm_amd/graphs.py): " + json.dumps(graph_stats)) if graph_stats else "eager launches",
                       "step_api": ("forward_backward(batch, loss_weights): forward and backward enqueued back to back (--fused-step)"
                                    if fused_step[0] else "loss = combine(losses); loss.backward() (autograd, as train.py:164-168, :315)"),
                       "valid_token_rows": valid_rows, "token_rows_padded": per_gpu * 256,
                       "gflop_per_seq_fwd_padded_equivalent": round(f_fwd, 3),
                       "gemm_gflop_per_seq_executed_fwd_bwd": round(exec_gf_seq, 3),
                       "padded_schedule_value": round(padded_value, 2) if padded_value is not None else None,
                       "padded_schedule_note": "the same step with engine.unpad = False: every sequence computed on all 256 "
                                               "token rows as the reference does (2 steps after the timed region)",
                       "loss": round(loss_val, 4)},
            "roofline": {"bound": "mfma", "kernel": name, "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE)", "traffic_source": traffic_src,
                         "traffic_measured_at_commit": traffic_commit,
                         "launches_per_step": cnt // prof_steps, "avg_launch_us": round(ms * 1e3 / cnt, 2),
                         "event_timed_launches": "every GEMM" if prof_all else "the weight-gradient kernels only (in the timed region); every GEMM in 2 extra steps",
                         "whole_step_executed_gemm_tflops": round(exec_fl_step / (dt / args.steps) / 1e12, 1),
                         "whole_step_frac": round(exec_fl_step / (dt / args.steps) / 1e12 / PEAK_BF16_TFLOPS, 4),
                         "gemm_kernels_by_time": [{"kernel": k, "ms_per_step": round(kms, 3), "tflops": round(kfl / (kms * 1e-3) / 1e12, 1) if kms > 0 else None,
                                                   "launches_per_step": kcnt} for k, kms, kfl, kcnt in ranking[:6]],
                         "schedule": ("two streams (image side beside text side): in-situ durations are shared-chip durations"
                                      if model.engine.dual_stream else "single stream"),
                         "coattention_gemms": coatt,
                         "exclusive": exclusive,
                         "padded_equivalent_tflops": round(3 * f_fwd * 1e9 * value / 1e12, 1)},
        }
        if args.compute != "bf16":
            out["roofline"]["note"] = ("fp32x3: every GEMM runs over three bf16 planes (hi hi + lo hi + hi lo): the FLOPs priced here are the "
                                       "EXECUTED bf16 MFMA FLOPs, 3x the algorithmic fp32 FLOPs of the layer")
        if prof_all:
            out["roofline"].update(all_gemm_tflops=round(gemm_fl / (gemm_ms * 1e-3) / 1e12, 1),
                                   gemm_share_of_step=round(gemm_ms * 1e-3 / prof_steps / (dt / args.steps), 3),
                                   whole_step_executed_gemm_tflops=round(gemm_fl / prof_steps / (dt / args.steps) / 1e12, 1))
        if comm is not None:
            out["comm"] = comm
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.config, args.cpu_steps)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
