"""RCCL with more than one rank under pytest: runs only where at least two GPUs are visible (the build's GPU box has one:
skipped there; the first multi-GPU box exercises the collective here, not only under bench.py)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 4])
def test_data_parallel_bank_step_over_rccl(world):
    """N ranks on N GPUs (torch.distributed.run, backend nccl = RCCL): the all-reduced gradient bucket equals the sum of the
    ranks' buckets, the graph-replayed step (collective captured inside the graph when every rank's probe agrees, else two
    graphs with the eager all-reduce between them) equals the host-launched step bit for bit, every rank ends with the same
    parameters -- tests/mgpu_worker.py."""
    if not torch.cuda.is_available() or torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs (this box has {torch.cuda.device_count() if torch.cuda.is_available() else 0})")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    port = 29600 + (os.getpid() + world) % 300
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "mgpu_worker.py")]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=ROOT)
    assert r.returncode == 0 and "MGPU_OK" in r.stdout, (r.stdout[-3000:], r.stderr[-3000:])
