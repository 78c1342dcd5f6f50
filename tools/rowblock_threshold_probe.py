#!/usr/bin/env python3
"""bf16 mode, small and mid-size batches: forward time by the thresholds from which the row-block kernels (workgroups) and the
refinement chain kernel (64-row blocks) take over from the ring kernels / the 2K launches.  IEFVAD_ROWBLOCK_MIN_WGS and
IEFVAD_CHAIN_MIN_BLOCKS are read at model creation.  python tools/rowblock_threshold_probe.py"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iefvad_amd
from iefvad_amd import synth

margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
sd = synth.make_state_dict(7)
res = {}
COMBOS = ((256, 256), (128, 16), (128, 4), (192, 4), (128, 1000000))      # (row-block workgroups, chain blocks)
for thr in COMBOS:
    os.environ["IEFVAD_ROWBLOCK_MIN_WGS"] = str(thr[0])
    os.environ["IEFVAD_CHAIN_MIN_BLOCKS"] = str(thr[1])
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute="bf16")
    m.load_state_dict(sd)
    m = m.to("cuda:0").eval()
    for B in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64):
        x = torch.randn(B, 256, 768, device="cuda:0") * 0.45
        y = torch.randn(B, 256, 768, device="cuda:0") * 0.45
        with torch.no_grad():
            for _ in range(3):
                m(x, y, None, None, None)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = 20
            for _ in range(n):
                m(x, y, None, None, None)
            torch.cuda.synchronize()
            res[(thr, B)] = (time.perf_counter() - t) / n * 1e3
    del m
print("forward ms (bf16, K = 10, scores) by batch (chunks) and threshold (workgroups)")
print("   B " + "".join(f"{str(t[0]) + '/' + str(t[1]):>9s}" for t in COMBOS))
for B in (1, 2, 4, 6, 8, 12, 16, 24, 32, 48, 64):
    print(f"{B:4d} " + "".join(f"{res[(t, B)]:9.3f}" for t in COMBOS))
