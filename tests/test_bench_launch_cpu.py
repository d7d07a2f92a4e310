"""`python bench.py --gpus N` as a plain process starts its own ranks (bench.self_launch) and relays rank 0's line; the form
under an external torch.distributed.run keeps working.  --rank-probe exercises exactly that path without the engine, so it
runs on the CPU-only build box (gloo); the engine's N = 2 line through the same launcher is tests/test_gpu_bench_contract.py."""
import json
import os
import socket
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.strip().startswith("{")]


def _clean_env():
    env = dict(os.environ)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "UNIMM_SELF_LAUNCHED"):
        env.pop(k, None)
    return env


def test_plain_python_launch_spawns_the_ranks_and_relays_one_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rank-probe"], capture_output=True, text=True, timeout=600,
                       cwd=ROOT, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1, r.stdout
    assert lines[0]["n_gpus"] == 2 and lines[0]["comm"]["rccl_ranks"] == 2 and lines[0]["launcher"] == "self"
    assert all(l.strip().startswith("{") for l in r.stdout.splitlines() if l.strip()), r.stdout     # nothing but the line on stdout


def test_external_launcher_form_still_works():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2", "--rank-probe"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT, env=_clean_env())
    assert r.returncode == 0, r.stderr[-2000:]
    lines = _json_lines(r.stdout)
    assert len(lines) == 1 and lines[0]["n_gpus"] == 2 and lines[0]["comm"]["rccl_ranks"] == 2 and lines[0]["launcher"] == "external"


def test_a_failing_rank_fails_the_parent():
    """The parent exits with the children's return code: a batch of 1 sequence over 2 ranks is refused INSIDE the ranks."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--rank-probe", "--batch", "1"], capture_output=True,
                       text=True, timeout=600, cwd=ROOT, env=_clean_env())
    assert r.returncode != 0
    assert _json_lines(r.stdout) == []
    assert "without a sequence" in r.stderr
