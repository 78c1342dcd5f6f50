#!/usr/bin/env python3
"""f32 mode, small and mid-size batches: forward time by the grid-size rules that choose among the four fp32 tilings
(IEFVAD_F32_RULES = "tiny_max_blocks64,t256_min_blocks,small_max_blocks128", read when the library is loaded: one process per
setting).  python tools/f32_threshold_probe.py"""
import os, subprocess, sys
CHILD = r'''
import argparse, os, sys, time
sys.path.insert(0, os.getcwd())
import torch, iefvad_amd
from iefvad_amd import synth
margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute="f32")
m.load_state_dict(synth.make_state_dict(7)); m = m.to("cuda:0").eval()
out = []
for B in (8, 16, 24, 32, 48, 64, 96, 128, 192, 256):
    x = torch.randn(B, 256, 768, device="cuda:0") * 0.45; y = torch.randn(B, 256, 768, device="cuda:0") * 0.45
    with torch.no_grad():
        for _ in range(3): m(x, y, None, None, None)
        torch.cuda.synchronize(); t = time.perf_counter()
        for _ in range(5): m(x, y, None, None, None)
        torch.cuda.synchronize(); out.append((time.perf_counter() - t) / 5 * 1e3)
print(" ".join(f"{v:7.3f}" for v in out))
'''
print("forward ms (f32, K = 10, scores); columns B = 8 16 24 32 48 64 96 128 192 256")
for rules in ("320,256,256", "320,512,1024", "320,768,1024", "320,1024,1024", "320,1536,1024", "320,3072,1024", "320,100000,1024", "320,512,2048", "320,512,4096"):
    env = dict(os.environ, IEFVAD_F32_RULES=rules)
    r = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
    print(f"{rules:14s}: {r.stdout.strip().splitlines()[-1] if r.stdout.strip() else r.stderr[-300:]}")
