#!/usr/bin/env python3
"""Small-batch latency of MMFMIL.forward through libiefvad.so (host wall time per call, synchronised)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iefvad_amd
from iefvad_amd import synth

margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
sd = synth.make_state_dict(7)
for outputs in ("scores", "full"):
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs=outputs)
    m.load_state_dict(sd)
    m = m.to("cuda:0").eval()
    for B in (1, 2, 4, 8, 32):
        x = torch.randn(B, 256, 768, device="cuda:0") * 0.45
        y = torch.randn(B, 256, 768, device="cuda:0") * 0.45
        with torch.no_grad():
            for _ in range(3):
                m(x, y, None, None, None)
            torch.cuda.synchronize()
            t = time.perf_counter()
            n = 20
            for _ in range(n):
                o = m(x, y, None, None, None)
            torch.cuda.synchronize()
            dt = (time.perf_counter() - t) / n
            # host-only cost: enqueue without waiting
            t = time.perf_counter()
            for _ in range(n):
                o = m(x, y, None, None, None)
            host = (time.perf_counter() - t) / n
            torch.cuda.synchronize()
            m(x, y, None, None, None, timed=True)
            st = {k: round(v, 4) for k, v in m.last_stage_times.items() if k.endswith("_ms")}
        print(f"outputs={outputs:6s} B={B:3d}: {dt*1e3:8.3f} ms per forward (host enqueue {host*1e3:.3f} ms), {B*256/dt:,.0f} snippets/s  stages {st}")
