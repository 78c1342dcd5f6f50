#!/usr/bin/env python3
"""BASELINE config 5 on one GPU, for the profiler: the ShanghaiTech + MSAD test lists (438 videos, 17,732 snippets: real gt and
label order, synthetic features -- synth.config5_lists; every video its own mostly-empty chunks: ~440 chunks), K = 5
refinement steps, bf16 projections with fp32 state, packed evaluation loop (harness.score_loader -> iefvad_forward_videos, all
videos in one call per pass).  Prints snippets/s; run it under `rocprofv3 --kernel-trace --stats` / `--pmc ...`
(tools/collect_profiles.sh) for the per-kernel roofline counters."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iefvad_amd
from iefvad_amd import harness, synth

T, D, L, K = 256, 768, 2, 5
import numpy as np
lists = synth.config5_lists(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
items, lens = [], []
for seed, d in ((51, "shang"), (52, "msad")):
    for i, (n, c) in enumerate(zip(lists[d][0], lists[d][1])):
        img, ev = synth.make_video(seed, i, int(n))
        ci, _ = harness.process_split(img, T)
        ce, _ = harness.process_split(ev, T)
        items.append((torch.from_numpy(ci).unsqueeze(0), torch.from_numpy(ce).unsqueeze(0), (c,), torch.tensor([int(n)])))
        lens.append(int(n))
lengths, nvid = np.array(lens), len(lens)
torch.set_num_threads(harness.host_cpu_share())
margs = argparse.Namespace(visual_layers=L, visual_head=8, num_refinement_steps=K, lambda_ref=0.5, noise_model="StudentT", nu=8)
model = iefvad_amd.MMFMIL(14, D, T, D, 8, L, 8, 10, 10, "cuda", margs, outputs="scores", compute="bf16")
model.load_state_dict(synth.make_state_dict(19, D, L, K))
model = model.to("cuda:0").eval()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
harness.score_loader(model, items, T, "cuda:0", "msad", batch_chunks=4096)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(reps):
    harness.score_loader(model, items, T, "cuda:0", "msad", batch_chunks=4096)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / reps
chunks = sum(int(it[0].shape[1]) if it[0].dim() == 4 else 1 for it in items)
print(f"config 5 (K=5, bf16, {nvid} videos, {int(lengths.sum())} snippets in {chunks} chunks, one iefvad_forward_videos call per pass): "
      f"{dt * 1e3:.2f} ms per pass wall clock (staging of {int(lengths.sum()) * D * 8 / 1e6:.0f} MB of valid rows into pinned memory and "
      f"the H2D copy included), {int(lengths.sum()) / dt:,.0f} snippets/s")
