#!/usr/bin/env python3
"""Phase breakdown of iefvad_outproj_ln_chain_bf16_kernel from in-kernel s_memtime stamps (a -DOC_DIAG build of the library,
IEFVAD_LIB=build/libiefvad_ocdiag.so).  The stamps go to a buffer of their own; no output depends on them."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

B = 1024
buf = torch.zeros(2 * (B * 256 // 64) * 8, dtype=torch.int64, device="cuda:0")
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
d = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)     # the LAST out_proj launch of the forward (layer 1)
names = ["entry -> image ready (A load, ds_write, barrier)", "main loop (24 k-steps x 24 MFMAs)", "ring drain + barrier",
         "park half 0 + barrier", "LayerNorm half 0 (4 rows per wave) + stores", "park half 1 + barriers", "LayerNorm half 1 + stores"]
dd = np.diff(d, axis=1)
print(f"{d.shape[0]} workgroups; cycles (s_memtime ticks at 100 MHz?) per phase: median / p10 / p90")
for i, n in enumerate(names):
    print(f"  {n:56s} {np.median(dd[:, i]):9.0f} {np.percentile(dd[:, i], 10):9.0f} {np.percentile(dd[:, i], 90):9.0f}")
print(f"  {'whole workgroup':56s} {np.median(d[:, 7] - d[:, 0]):9.0f}")
