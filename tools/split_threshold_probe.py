#!/usr/bin/env python3
"""bf16x6 mode, mid-size batches: forward time by the grid size from which the split kernels take a micro-batch over from the
fp32 kernels (IEFVAD_SPLIT_MIN_WGS, read when the library is loaded: one process per value).  python tools/split_threshold_probe.py"""
import os, subprocess, sys
CHILD = r'''
import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, iefvad_amd
from iefvad_amd import synth
margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute="bf16x6")
m.load_state_dict(synth.make_state_dict(7)); m = m.to("cuda:0").eval()
out = []
for B in (1, 2, 3, 4, 6, 8, 12, 16, 24, 32, 48):
    x = torch.randn(B, 256, 768, device="cuda:0") * 0.45; y = torch.randn(B, 256, 768, device="cuda:0") * 0.45
    with torch.no_grad():
        for _ in range(3): m(x, y, None, None, None)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(10): m(x, y, None, None, None)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 10 * 1e3)
print(" ".join(f"{v:8.3f}" for v in out))
'''
print("forward ms (bf16x6, K = 10, scores); columns B = 1 2 3 4 6 8 12 16 24 32 48")
for thr in (512, 96, 72, 48, 36, 24, 12):
    env = dict(os.environ, IEFVAD_SPLIT_MIN_WGS=str(thr))
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"min workgroups {thr:4d}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")
