#!/usr/bin/env python3
"""Where the host time of the per-video evaluation loop goes (cProfile of harness.score_loader on a UCF-sized synthetic
list, f32, lanes = 4).  Diagnostic only."""
import argparse, cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iefvad_amd
from iefvad_amd import harness, synth

T = 256
seed, nvid, total_target = 1, 290, 69500
lengths = synth.lognormal_lengths(seed, nvid, total_target)
items = []
for i, n in enumerate(lengths):
    img, ev = synth.make_video(seed, i, int(n))
    ci, _ = harness.process_split(img, T)
    ce, _ = harness.process_split(ev, T)
    items.append((torch.from_numpy(ci).unsqueeze(0), torch.from_numpy(ce).unsqueeze(0), ("Normal",), torch.tensor([int(n)])))
torch.set_num_threads(harness.host_cpu_share())
margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
model = iefvad_amd.MMFMIL(14, 768, T, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores")
model.load_state_dict(synth.make_state_dict(0, 768, 2, 10))
model = model.to("cuda:0").eval()
lanes = int(sys.argv[1]) if len(sys.argv) > 1 else 4
bc = int(sys.argv[2]) if len(sys.argv) > 2 else 0
if len(sys.argv) > 3:
    model = iefvad_amd.MMFMIL(14, 768, T, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute=sys.argv[3])
    model.load_state_dict(synth.make_state_dict(0, 768, 2, 10))
    model = model.to("cuda:0").eval()
harness.score_loader(model, items[:16], T, "cuda:0", "ucfcrime", lanes=lanes, batch_chunks=bc)
torch.cuda.synchronize()
for rep in range(2):
    t0 = time.perf_counter()
    harness.score_loader(model, items, T, "cuda:0", "ucfcrime", lanes=lanes, batch_chunks=bc)
    torch.cuda.synchronize()
    print(f"lanes={lanes} batch_chunks={bc} {sys.argv[3] if len(sys.argv) > 3 else 'f32'}: {time.perf_counter() - t0:.4f} s for {nvid} videos")
pr = cProfile.Profile()
pr.enable()
harness.score_loader(model, items, T, "cuda:0", "ucfcrime", lanes=lanes, batch_chunks=bc)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
