#!/usr/bin/env python3
"""Phase breakdown of the persistent out_proj + LayerNorm kernel (outproj_ln_pchain_bf16.h) from in-kernel s_memtime sums (a -DOC_DIAG
build: make -C ief-vad_amd/csrc EXTRA=-DOC_DIAG OUT=../../build/libiefvad_ocdiag.so; IEFVAD_LIB=build/libiefvad_ocdiag.so)."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

B = 1024
buf = torch.zeros(2 * 128 * 8, dtype=torch.int64, device="cuda:0")
os.environ["IEFVAD_OC_DIAG_PTR"] = str(buf.data_ptr())
import iefvad_amd
from iefvad_amd import synth
margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute="bf16")
m.load_state_dict(synth.make_state_dict(7))
m = m.to("cuda:0").eval()
g = torch.Generator(device="cuda:0"); g.manual_seed(1)
x = torch.randn(B, 256, 768, device="cuda:0", generator=g) * 0.45
y = torch.randn(B, 256, 768, device="cuda:0", generator=g) * 0.45
with torch.no_grad():
    for _ in range(3):
        m(x, y, None, None, None)
torch.cuda.synchronize()
d = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)     # the LAST out_proj launch of the forward (layer 1: with whitening)
if os.environ.get("IEFVAD_OUTLN", "p")[0] == "p":      # round 4's kernel (outproj_ln_pchain_bf16.h)
    nb = d[:, 5]
    names = ["top of block -> image ready (wait + barrier)", "main loop (24 k-steps x 24 MFMAs) + ring drain", "image-free barrier + next image DMA issue",
             "4 x (park + barrier) + 4 x end barrier", "4 x (LayerNorm(s) of 2 rows per wave + stores)"]
else:                                                   # round 5's (outproj_ln_rchain_bf16.h: LayerNorm in the accumulator registers)
    nb = d[:, 6]
    names = ["top of block -> image ready (wait + barrier)", "main loop (24 k-steps x 24 MFMAs) + ring drain",
             "bias + residual of row group 0, requests of groups 1 - 3, image-free barrier, next image DMA issue", "wait for the residual rows, bias + residual of groups 1 - 3",
             "LayerNorm(s) in the accumulator layout (2 exchanges each)", "stores (fp32 / bf16 pieces)"]
print(f"{d.shape[0]} workgroups, {nb.mean():.1f} blocks each; s_memtime ticks (100 MHz) per BLOCK: median / p10 / p90")
tot = 0
for i, n in enumerate(names):
    v = d[:, i] / nb
    tot += np.median(v)
    print(f"  {n:100s} {np.median(v):9.1f} {np.percentile(v, 10):9.1f} {np.percentile(v, 90):9.1f}")
print(f"  {'sum':100s} {tot:9.1f}")
