"""Worker of tests/test_gpu_distributed.py (started by torch.distributed.run, one process per rank, all on cuda:0 of the
one-GPU test box): every rank scores its contiguous, snippet-balanced shard of a synthetic test list with the HIP library
(f32 mode: bit-reproducible across batch compositions), the shards are gathered in rank order, and rank 0 checks the result
against ITS OWN single-process pass over the whole list, bit for bit.  Prints "DIST_GPU_OK <n>" on success."""
import argparse
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iefvad_amd                                   # noqa: E402
from iefvad_amd import harness, synth               # noqa: E402

LENGTHS = [40, 300, 17, 256, 90, 520, 33, 1, 700, 255, 257, 64]


def items(videos):
    for img, ev in videos:
        ci, n = harness.process_split(img, 256)
        ce, _ = harness.process_split(ev, 256)
        yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([n])


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo")                 # the ranks share one GPU here: RCCL refuses that, gloo carries the scores
    torch.cuda.set_device(0)
    margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=3, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores")
    model.load_state_dict(synth.make_state_dict(9, 768, 2, 3))
    model = model.to("cuda:0").eval()
    videos = [synth.make_video(3, i, n) for i, n in enumerate(LENGTHS)]
    a, b = harness.partition_by_snippets(LENGTHS, world)[rank]
    scores, _, _, _ = harness.score_loader(model, items(videos[a:b]), 256, "cuda:0", "ucfcrime", batch_chunks=3, lanes=2)
    local = torch.from_numpy(np.concatenate(scores) if scores else np.zeros(0, np.float32))
    counts = harness.shard_counts(LENGTHS, world)
    assert counts[rank] == local.numel(), (counts, local.numel())
    full = harness.gather_scores(local, counts=counts)
    if rank == 0:
        single, _, _, _ = harness.score_loader(model, items(videos), 256, "cuda:0", "ucfcrime")       # one forward per video
        single = np.concatenate(single)
        got = full.numpy()
        assert got.shape == single.shape == (sum(LENGTHS),), (got.shape, single.shape)
        assert np.array_equal(got, single), float(np.abs(got - single).max())
        print(f"DIST_GPU_OK {got.size}", flush=True)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
