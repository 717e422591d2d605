"""N>1 path on CPU: world_size-2 (and 4) gloo jobs launched exactly like the driver launches bench.py."""
import os
import socket
import subprocess
import sys

import pytest

from cnn_amd import dp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("world", [2, 4])
def test_sharded_step_equals_full_batch_step(world):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr",
           "127.0.0.1", "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "dp_worker.py")]
    env = dict(os.environ, OMP_NUM_THREADS="1")
    out = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    assert f"DP_OK world={world}" in out.stdout
    assert f"SYNCBN_OK world={world}" in out.stdout
    assert f"TWO_COMMS_OK world={world}" in out.stdout


def test_shard_bounds():
    assert [dp.shard_bounds(2048, r, 8) for r in (0, 7)] == [(0, 256), (1792, 2048)]
    with pytest.raises(ValueError):
        dp.shard_bounds(10, 0, 4)
    assert dp.allreduce_grads(None, None, 1) == 1.0
