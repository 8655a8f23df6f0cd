"""CPU: bench.py's N > 1 launch path exactly as the driver starts it (`python -m torch.distributed.run --nnodes=1
--nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...`), 8 ranks over gloo, before any
GPU call: `--dry-run` swaps PSMNet for a stand-in model (the HIP kernels have no CPU fallback) and keeps everything
else -- rendezvous from the environment, the world-size / backend assertions, per-rank data, DDP wrap, barrier +
max-over-ranks timing, rank 0's single JSON line."""
import json
import os
import socket
import subprocess
import sys

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(600)
@pytest.mark.parametrize("n", [8])
def test_bench_launch_path_eight_ranks_gloo(n):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(REPO, "bench.py"), "--gpus", str(n),
           "--steps", "3", "--warmup", "1", "--dry-run", "--dist-backend", "gloo"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=REPO, env=env, timeout=540)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout  # ONE JSON line, from rank 0
    out = json.loads(lines[0])
    assert out["n_gpus"] == n and out["steps"] == 3 and out["warmup"] == 1
    assert out["config"]["global_batch"] == 4 * n and out["config"]["parallelism"] == f"dp{n}"
    assert out["config"]["dist_backend"] == "gloo" and out["scaling"] == "weak"
    assert out["replicas_in_sync"] and out["ranks_drew_distinct_data"]
    assert out["vs_baseline"] is None and "dry_run" in out and out["value"] > 0


def test_bench_refuses_a_world_size_that_does_not_match_gpus():
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--dry-run", "--dist-backend", "gloo"],
                       capture_output=True, text=True, cwd=REPO, env={k: v for k, v in os.environ.items()
                                                                      if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")})
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)
