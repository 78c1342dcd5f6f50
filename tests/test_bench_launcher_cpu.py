"""`python bench.py --gpus N` must start N ranks by itself (the round driver launches it from a plain shell) and
print ONE JSON line from rank 0.  No GPU here, so the launcher runs in `--plumbing-only` mode: no forward, each
rank fabricates the scores of its shard as global snippet indices, and the launch -> strong split of config 4
(rank r takes chunks [r*B/N, (r+1)*B/N)) -> gather -> max-over-ranks -> JSON path is what is checked.
The reference fixes only the ORDER of the score vector (/root/reference/test.py:123-129,153)."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["OMP_NUM_THREADS"] = "2"
    return env


def _one_json_line(stdout: str) -> dict:
    lines = [ln for ln in stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("chunks,expect", [(8, [4, 4]), (11, [5, 6])])
def test_plain_shell_launch_of_two_ranks(chunks, expect):
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only", "--steps", "3", "--warmup", "1",
                        "--chunks", str(chunks)], env=_env(), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["scaling"] == "strong" and line["value"] is None and "plumbing_only" in line
    assert line["config"]["chunks_per_gpu"] == expect and line["config"]["chunks_total"] == chunks
    assert line["gathered_scores"] == chunks * 256 and line["gathered_in_order"] is True
    assert len(line["per_rank_ms_per_step"]) == 2 and line["ms_per_step"] >= max(line["per_rank_ms_per_step"]) * 0.999
    assert line["dist_backend"] == "gloo"


def test_launch_under_torch_distributed_run_as_the_driver_does():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(port), BENCH, "--gpus", "2",
                        "--plumbing-only", "--steps", "2", "--warmup", "1", "--chunks", "6", "--scaling", "weak"],
                       env=_env(), capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["scaling"] == "weak" and line["config"]["chunks_per_gpu"] == [6, 6]
    assert line["gathered_scores"] == 2 * 6 * 256


def test_world_size_mismatch_is_an_error():
    env = _env()
    env.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--plumbing-only"], env=env, capture_output=True,
                       text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE" in (r.stderr + r.stdout)


def test_eight_ranks_line_has_what_the_first_real_run_will_be_checked_for():
    """N = 8 on CPU ranks (gloo): the strong split of BASELINE config 4 -- 8192 chunks, rank r holds [1024 r, 1024 (r + 1)) -- and
    every field the driver's 8-GPU record is read for: `rccl_ranks`, `gather`, eight `per_rank_ms_per_step` entries, the gathered
    vector in list order."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--plumbing-only", "--steps", "2", "--warmup", "1"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["n_gpus"] == 8 and line["config"]["chunks_per_gpu"] == [1024] * 8 and line["config"]["chunks_total"] == 8192
    assert len(line["per_rank_ms_per_step"]) == 8 and all(t > 0 for t in line["per_rank_ms_per_step"])
    assert "rccl_ranks" in line and isinstance(line["gather"], str) and line["gather"]
    assert line["gathered_scores"] == 8192 * 256 and line["gathered_in_order"] is True
    assert line["scaling"] == "strong" and line["ms_per_step"] >= max(line["per_rank_ms_per_step"]) * 0.999


@pytest.mark.parametrize("mode", ["fail", "hang"])
def test_comm_create_failure_on_one_rank_moves_every_rank_to_the_torch_transport(mode):
    """harness.ScoreComm's handshake with the library's communicator create failing -- or never returning -- on ONE rank (injected on
    CPU ranks; on hardware: a rank whose ncclCommInitRank fails, or a peer that never enters it): the outcome is all-reduced, so
    every rank takes torch.distributed's all-gather, the line says why, rc is 0, and the deadline on the collective create keeps
    the job from hanging."""
    import time
    t0 = time.time()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--plumbing-only", "--steps", "2", "--warmup", "1", "--chunks", "9",
                        "--plumbing-comm", mode, "--plumbing-comm-timeout", "4"], env=_env(), capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = _one_json_line(r.stdout)
    assert line["gather"].startswith("torch.distributed all_gather_into_tensor (library gather unavailable:")
    assert ("did not return within" in line["gather"]) or ("failed on another rank" in line["gather"]) or ("injected failure" in line["gather"])
    assert line["rccl_ranks"] == 0 and line["gathered_scores"] == 9 * 256 and line["gathered_in_order"] is True
    assert time.time() - t0 < 200
