#!/usr/bin/env python
"""Self-check of a multi-GPU bench line against DESIGN.md section 6's predictions.

    python bench.py --gpus N ... > line.json          (starts its own ranks; or under python -m torch.distributed.run --nproc-per-node N)
    python tools/check_scale.py line.json [more.json ...]        (or: ... | python tools/check_scale.py -)

No N > 1 box was available to this build in any round, so the first real RCCL run has to check itself: the line must say
that RCCL really saw N ranks (`comm.rccl_ranks`, an all-reduce of ones), that every gradient bucket was exchanged once per
step in the predicted number of collectives, and that the step time and the exposed (non-overlapped) exchange time are
within 25 % (+ an absolute slack for sub-millisecond figures) of DESIGN.md 6's table.  Exit code 1 on any deviation, with
one line per finding -- a deviation is a finding about the model of section 6, not necessarily a bug.

Expected values (1-GPU figures measured in round 6 (profiles/r6z_*), strong scaling of the global batch of 240;
the number of collectives is computed from unimm_amd/bucket_plan.py, the rule the engine itself uses):
  N  per-GPU  step without exchange   collectives/step   exposed exchange (fp32 wire)
  1    240        40.0 ms                   0                  0
  2    120        22.3 ms (eager)           14                 <= 4.5 ms: ONE xGMI link pair carries the 1.0 GB (~18 ms at ~55 GB/s per direction:
                                                                 more than backward can hide); the tail alone (181 MB) is 3.3 ms.  `--wire bf16` halves both
  4     60        13.5 ms (graph replay)    14                 <= 1.8 ms (three links per GPU, ~150 GB/s bus bandwidth; tail 181 MB)
  8     30         9.0 ms (graph replay)    14                 ~1.0 ms (181 MB x 1.75 at ~320 GB/s; 2.1 ms with the 392 MB tail of rounds 3-5)
(step = the 1-GPU step of that share + 2 % for the 2-round weight-gradient grouping under N > 1)
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _collectives():
    """Collectives per step of the N-rank step, from the exchange planner (host arithmetic, no torch)."""
    try:
        from unimm_amd import bucket_plan as BP
        from unimm_amd.config import BertConfig
        cfg = BertConfig.from_json_file(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "unimm_amd", "config",
                                                     "bert_base_6layer_6conect.json"))
        return {n: len(BP.exchange_plan(cfg, 240 // n, 31162 // n, 5016 // n, wgrad_group_rounds=2)["collectives"]) for n in (2, 4, 8)}
    except Exception:
        return {2: 14, 4: 14, 8: 14}


_C = _collectives()
EXPECT = {   # n_gpus: (step_ms_without_exchange, collectives_per_step, exposed_exchange_ms)
    1: (40.0, 0, 0.0),
    2: (22.3, _C[2], 4.5),
    4: (13.5, _C[4], 1.8),
    8: (9.0, _C[8], 1.0),
}
REL = 0.25
ABS_MS = 0.6          # slack for the sub-millisecond exposed-exchange figures


def check(line):
    msgs = []
    n = int(line["n_gpus"])
    if line.get("scaling") != "strong" or line.get("config", {}).get("global_batch") != 240:
        return [f"n_gpus={n}: not the strong-scaling bs=240 line (scaling={line.get('scaling')}, global_batch="
                f"{line.get('config', {}).get('global_batch')}): nothing to compare against"]
    exp = EXPECT.get(n)
    if exp is None:
        return [f"n_gpus={n}: no prediction in DESIGN.md 6 (have {sorted(EXPECT)})"]
    step_exp, coll_exp, exposed_exp = exp
    comm = line.get("comm")
    if n > 1:
        if comm is None:
            return [f"n_gpus={n}: the line has no `comm` block"]
        if int(comm.get("rccl_ranks", -1)) != n:
            msgs.append(f"n_gpus={n}: RCCL saw {comm.get('rccl_ranks')} ranks, not {n} (backend {comm.get('backend')})")
        if comm.get("backend") != "nccl":
            msgs.append(f"n_gpus={n}: process-group backend is {comm.get('backend')!r}, not 'nccl' (= RCCL); rehearsal runs are not measurements")
        c = int(comm.get("collectives_per_step", -1))
        if abs(c - coll_exp) > REL * coll_exp:
            msgs.append(f"n_gpus={n}: {c} collectives per step, predicted {coll_exp}")
        ex = float(comm.get("exposed_exchange_ms", -1.0))
        if ex > exposed_exp * (1 + REL) + ABS_MS:
            msgs.append(f"n_gpus={n}: exposed exchange {ex:.2f} ms, predicted <= {exposed_exp:.2f} ms (+25 % + {ABS_MS} ms)")
        base = float(comm.get("step_ms_without_exchange", -1.0))
        if abs(base - step_exp) > REL * step_exp:
            msgs.append(f"n_gpus={n}: step without exchange {base:.2f} ms, predicted {step_exp:.2f} ms")
        if float(comm.get("busbw_frac_of_xgmi", 0.0)) > 1.0:
            msgs.append(f"n_gpus={n}: bus bandwidth above the 7 x 153 GB/s xGMI budget: timing is wrong")
    total = float(line["ms_per_step"])
    pred = step_exp + (exposed_exp if n > 1 else 0.0)
    if abs(total - pred) > REL * pred + (ABS_MS if n > 1 else 0.0):
        msgs.append(f"n_gpus={n}: {total:.2f} ms per step, predicted {pred:.2f} ms")
    val = float(line["value"])
    if abs(val - 240.0 / (total * 1e-3)) > 0.02 * val:
        msgs.append(f"n_gpus={n}: value {val:.1f} seq/s is not global_batch / ms_per_step ({240.0 / (total * 1e-3):.1f})")
    return msgs


def main(argv):
    lines = []
    for a in argv or ["-"]:
        txt = sys.stdin.read() if a == "-" else open(a).read()
        for ln in txt.splitlines():
            ln = ln.strip()
            if ln.startswith("{") and '"metric"' in ln:
                lines.append(json.loads(ln))
    if not lines:
        print("check_scale: no bench line found", file=sys.stderr)
        return 2
    bad = 0
    for line in lines:
        msgs = check(line)
        for m in msgs:
            print("DEVIATION " + m)
        bad += len(msgs)
        if not msgs:
            print(f"ok n_gpus={line['n_gpus']}: {line['value']} seq/s, {line['ms_per_step']} ms per step")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
