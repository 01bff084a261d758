"""World-size-2 `gloo` test of the multi-GPU drivers' host logic (no GPU): track partition and result
gather for file batches (no data-path collective), and the neighbour exchange + all-reduce of a
time-sharded stream.  The GPU variant is tests/test_gpu_distributed.py."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_workers(mode, world, timeout=300):
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(HERE, "dist_worker.py"), mode]
    p = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    assert f"dist_worker {mode}: ok (world {world})" in p.stdout


@pytest.mark.parametrize("world", [2, 3])
def test_sharding_drivers_gloo(world):
    run_workers("cpu", world)
