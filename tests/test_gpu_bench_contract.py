"""bench.py prints ONE JSON line with the fields the contract names (a small batch, one step; run as a child process).  N > 1 is
run in both launch forms: `python bench.py --gpus 2` (bench.self_launch starts the ranks) and under an external
torch.distributed.run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_the_contract_fields():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "12",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 2 and j["warmup"] == 1 and j["higher_is_better"] is True
    assert j["unit"] == "dialog-sequences/sec" and j["dtype"] == "bf16" and j["data"] == "synthetic" and j["vs_baseline"] is None
    assert abs(j["value"] - 12 / (j["ms_per_step"] * 1e-3)) <= 0.01 * j["value"]
    assert "workload" in j["config"] and "model" not in j["config"]
    rf = j["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in rf, k
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert rf["exclusive"] is None or rf["exclusive"]["achieved"] > 0
    # executed work, whole-step fraction, measured kernel ranking and the padded schedule are always in the line
    assert j["config"]["gemm_gflop_per_seq_executed_fwd_bwd"] > 0
    assert 0 < rf["whole_step_frac"] < 1 and abs(rf["whole_step_frac"] - rf["whole_step_executed_gemm_tflops"] / rf["peak"]) < 1e-3
    assert rf["gemm_kernels_by_time"] and rf["gemm_kernels_by_time"][0]["kernel"] == rf["kernel"]
    assert j["config"]["padded_schedule_value"] > 0


def test_bench_two_ranks_rehearsal_with_the_graph_executor():
    """The N > 1 path of bench.py under an external torch.distributed.run (one JSON line from rank 0), rehearsed on
    one GPU: UNIMM_BENCH_REHEARSAL=1 puts both ranks on device 0 and exchanges over gloo (RCCL refuses two ranks on one
    device).  A small global batch, the graph executor forced on: the data-parallel hooks between the replayed segments of
    backward, the bucket bookkeeping assert of the timed region, the `comm` block."""
    import socket
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    env = dict(os.environ, UNIMM_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "24", "--graphs", "on", "--no-cpu-baseline", "--no-padded"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip().startswith("{")]
    assert len(lines) == 1, lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 24 and j["config"]["per_gpu_batch"] == 12
    assert j["config"]["executor"].startswith("hipGraph replay"), j["config"]["executor"]
    c = j["comm"]
    assert c["buckets_per_step"] > 0 and 0 < c["collectives_per_step"] <= c["buckets_per_step"]
    assert c["bytes_per_step"] > 0 and c["exchange_alone_ms"] > 0


def test_bench_two_ranks_rehearsal_launched_as_plain_python():
    """`python bench.py --gpus 2` with no launcher around it: bench.self_launch spawns the two ranks as child processes and relays
    rank 0's line (rehearsal: both ranks on device 0 over gloo).  tools/check_scale.py reads the same line."""
    env = dict(os.environ, UNIMM_BENCH_REHEARSAL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "24",
                        "--no-cpu-baseline", "--no-padded"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and lines[0].startswith("{"), lines
    j = json.loads(lines[0])
    assert j["n_gpus"] == 2 and j["config"]["global_batch"] == 24 and j["comm"]["rccl_ranks"] == 2
    import importlib.util
    spec = importlib.util.spec_from_file_location("check_scale", os.path.join(ROOT, "tools", "check_scale.py"))
    cs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cs)
    msgs = cs.check(j)                       # a rehearsal is gloo on one device at 12 sequences per rank: only the structural checks apply
    assert not any("is not global_batch / ms_per_step" in m or "no `comm` block" in m or "RCCL saw" in m for m in msgs), msgs
