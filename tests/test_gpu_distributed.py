"""N > 1 through the HIP library: three processes (torch.distributed.run, gloo rendezvous on 127.0.0.1) share the test box's
one GPU, each scores its shard with libiefvad, the gathered vector must equal a single-process pass bit for bit
(tests/dist_gpu_worker.py).  The RCCL transport itself needs one GPU per rank and is the round driver's N > 1 bench run;
its one-rank form is tests/test_gpu_gather.py.  This process never touches the GPU."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_scoring_on_the_hip_library_equals_single_process(world):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["OMP_NUM_THREADS"] = "2"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dist_gpu_worker.py")]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert "DIST_GPU_OK 2533" in r.stdout, r.stdout[-2000:]
