#!/usr/bin/env python3
"""B = 1 (one chunk per call, the reference's per-video pattern) forward in a loop, for rocprofv3 --kernel-trace."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iefvad_amd
from iefvad_amd import synth

ap = argparse.ArgumentParser()
ap.add_argument("--B", type=int, default=1)
ap.add_argument("--iters", type=int, default=200)
ap.add_argument("--compute", default="f32")
ap.add_argument("--graph", type=int, default=-1)
a = ap.parse_args()
margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute=a.compute, graph_chunks=a.graph)
m.load_state_dict(synth.make_state_dict(7))
m = m.to("cuda:0").eval()
x = torch.randn(a.B, 256, 768, device="cuda:0") * 0.45
y = torch.randn(a.B, 256, 768, device="cuda:0") * 0.45
with torch.no_grad():
    for _ in range(a.iters):
        m(x, y, None, None, None)
torch.cuda.synchronize()
print("done")
