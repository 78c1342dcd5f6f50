#!/usr/bin/env python3
"""Does running two independent halves of the batch on two HIP streams (blocks of different kernels interleave
on the CUs, so one kernel's store phase overlaps the other's MFMA phase) beat one stream?  Timing probe only."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iefvad_amd
from iefvad_amd import synth

margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
sd = synth.make_state_dict(0)
B = 4096
img = torch.randn(B, 256, 768, device="cuda:0") * 0.45
ev = torch.randn(B, 256, 768, device="cuda:0") * 0.45
for compute, mb in (("f32", 256), ("bf16", 256), ("bf16", 1024)):
    models = []
    for _ in range(2):
        m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute=compute, micro_batch=mb)
        m.load_state_dict(sd)
        models.append(m.to("cuda:0").eval())
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def one_stream():
        with torch.no_grad():
            models[0](img, ev, None, None, None)
    def two_streams():
        h = B // 2
        cur = torch.cuda.current_stream()
        with torch.no_grad():
            for k, s in enumerate(streams):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    models[k](img[k * h:(k + 1) * h], ev[k * h:(k + 1) * h], None, None, None)
            for s in streams:
                cur.wait_stream(s)
    for name, fn in (("one stream ", one_stream), ("two streams", two_streams)):
        fn(); fn(); torch.cuda.synchronize()
        t = time.perf_counter()
        for _ in range(3): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t) / 3
        print(f"{compute} mb={mb} {name}: {dt*1e3:8.1f} ms per {B} chunks -> {B*256/dt/1e6:.2f} M snippets/s")
