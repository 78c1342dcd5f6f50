#!/usr/bin/env python3
"""Phase breakdown of a row-block kernel from in-kernel s_memtime stamps (a -DHC_DIAG -DIC_DIAG -DRC_DIAG build of the library,
IEFVAD_LIB=build/libiefvad_<x>diag.so).  Usage: rowblock_diag.py heads|heads_p|inproj|chain (heads_p: the persistent heads kernel, -DHC_DIAG).  The stamps go to a buffer of their own; no
output depends on them; the LAST launch of the kernel in the forward is what remains in the buffer."""
import argparse, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

which = sys.argv[1] if len(sys.argv) > 1 else "heads"
B = 1024
PH = {"heads": (3 * (B * 256 // 64), "IEFVAD_HC_DIAG_PTR",
                ["entry -> x_i image ready", "phase 1 main loop (mu_i, logvar_i)", "x_e image: loads, barrier, ds_write, barrier",
                 "phase 2 main loop (mu_e, logvar_e)", "epilogue (fusion in registers, stores)"]),
      # the persistent heads kernel (heads_pchain_bf16.h): per-workgroup SUMS over its blocks, eight phases
      "heads_p": (3 * 128, "IEFVAD_HC_DIAG_PTR",
                  ["top of block: ring requests, image wait, barrier", "phase 1 main loop (x_i: mu_i, logvar_i)", "x_i-free barrier + DMA of x_e's last third",
                   "phase 2 main loop, k < 512", "barrier in front of k = 512", "phase 2 main loop, k >= 512",
                   "ring drain, barrier, next block's image DMA issue", "fusion epilogue in registers + stores"]),
      "inproj": (2 * (B * 256 // 64), "IEFVAD_IC_DIAG_PTR",
                 ["entry -> image ready", "pass q main loop", "pass q epilogue", "pass k main loop", "pass k epilogue",
                  "pass v main loop", "pass v epilogue"]),
      "chain": (2 * (B * 256 // 64), "IEFVAD_RC_DIAG_PTR", None)}[which]
buf = torch.zeros(PH[0] * 8, dtype=torch.int64, device="cuda:0")
os.environ[PH[1]] = str(buf.data_ptr())
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
if which == "chain":
    d = buf.cpu().numpy().reshape(-1, 16)[: B * 256 // 64].astype(np.float64)
    names = ["pass 0 main loop", "pass 0 bias pieces", "pass 0 epilogue (park / z update) + zeroing", "pass 1 main loop", "bias + barrier (image free)",
             "pass 1 epilogue (image rewritten)", "lgkmcnt + barrier (image ready)"]
    for gi, gn in ((0, "projection 2 (W1: relu -> h)"), (1, "projection 3 (W2: z update)")):
        dd = np.diff(d[:, 8 * gi:8 * gi + 8], axis=1)
        print(f"chain, {gn}: {d.shape[0]} workgroups, wave 0's s_memtime ticks per phase: median / p10 / p90")
        for i, name in enumerate(names):
            print(f"  {name:44s} {np.median(dd[:, i]):9.0f} {np.percentile(dd[:, i], 10):9.0f} {np.percentile(dd[:, i], 90):9.0f}")
        print(f"  {'whole projection':44s} {np.median(d[:, 8 * gi + 7] - d[:, 8 * gi]):9.0f}")
    print(f"  W1 start -> W2 start {np.median(d[:, 8] - d[:, 0]):9.0f}")
    sys.exit(0)
if which == "heads_p":
    d = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
    d = d[d.sum(axis=1) > 0]
    nb = (B * 256 // 64) / (d.shape[0] / 3)
    print(f"heads (persistent): {d.shape[0]} workgroups, {nb:.1f} blocks each; s_memtime ticks per BLOCK: median / p10 / p90")
    tot = 0
    for i, name in enumerate(PH[2]):
        v = d[:, i] / nb
        tot += np.median(v)
        print(f"  {name:56s} {np.median(v):9.1f} {np.percentile(v, 10):9.1f} {np.percentile(v, 90):9.1f}")
    print(f"  {'sum':56s} {tot:9.1f}")
    sys.exit(0)
n = len(PH[2])
d = buf.cpu().numpy().reshape(-1, 8).astype(np.float64)
dd = np.diff(d[:, :n + 1], axis=1)
print(f"{which}: {d.shape[0]} workgroups; s_memtime ticks per phase: median / p10 / p90")
for i, name in enumerate(PH[2]):
    print(f"  {name:56s} {np.median(dd[:, i]):9.0f} {np.percentile(dd[:, i], 10):9.0f} {np.percentile(dd[:, i], 90):9.0f}")
print(f"  {'whole workgroup':56s} {np.median(d[:, n] - d[:, 0]):9.0f}")
