"""N > 1 path on the CPU: two gloo ranks each score a contiguous, snippet-balanced shard of the test
list and all-gather the scores; the result must equal the single-process order bit for bit."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from iefvad_amd import harness, synth
from oracle import iefvad_oracle as orc

LENGTHS = [40, 300, 17, 256, 90, 520, 33]


def _videos():
    return [synth.make_video(3, i, n) for i, n in enumerate(LENGTHS)]


def _model():
    sd = synth.make_state_dict(9, 768, 1, 1)
    return orc.OracleMMFMIL(sd, orc.OracleConfig(num_layers=1, num_refinement_steps=1))


def _items(videos):
    for img, ev in videos:
        ci, n = harness.process_split(img, 256)
        ce, _ = harness.process_split(ev, 256)
        yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([n])


def _worker(rank, world, port, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(2)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    videos = _videos()
    a, b = harness.partition_by_snippets(LENGTHS, world)[rank]
    scores, _, _, _ = harness.score_loader(_model(), _items(videos[a:b]), 256, "cpu", "ucfcrime", batch_chunks=4)
    local = torch.from_numpy(np.concatenate(scores) if scores else np.zeros(0, np.float32))
    full = harness.gather_scores(local)
    if rank == 0:
        np.save(out_path, full.numpy())
    dist.destroy_process_group()


def test_two_rank_sharded_scores_equal_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "gathered.npy")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    torch.set_num_threads(4)
    scores, _, _, _ = harness.score_loader(_model(), _items(_videos()), 256, "cpu", "ucfcrime", batch_chunks=4)
    single = np.concatenate(scores)
    got = np.load(out)
    assert got.shape == single.shape == (sum(LENGTHS),)
    assert np.abs(got - single).max() < 1e-6
