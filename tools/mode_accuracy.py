#!/usr/bin/env python3
"""Error of every compute mode against the fp64 oracle on the same seeded inputs (B = 48 chunks, K = 10): max and rms over
each of the eight outputs.  Prints one JSON line."""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import iefvad_amd
from iefvad_amd import synth
from oracle import iefvad_oracle as orc

B = 48
sd = synth.make_state_dict(0)
img, ev = synth.make_inputs(7, B)
cfg = orc.OracleConfig(num_layers=2, num_refinement_steps=10, nu=8)
torch.set_num_threads(iefvad_amd.harness.host_cpu_share())
ref64 = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), cfg, dtype=torch.float64)
ref32 = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), cfg)
margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
res = {"B": B, "oracle_fp32_cpu": {}}
for k in ref64:
    d = (ref32[k].double() - ref64[k]).abs()
    res["oracle_fp32_cpu"][k] = [float(d.max()), float(d.pow(2).mean().sqrt())]
for mode in ("f32", "bf16x6", "fp16x3", "bf16"):
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, compute=mode)
    m.load_state_dict(sd)
    m = m.to("cuda:0").eval()
    with torch.no_grad():
        out = m(torch.from_numpy(img).cuda(), torch.from_numpy(ev).cuda(), None, None, None)
    res[mode] = {}
    for k in ref64:
        d = (out[k].double().cpu() - ref64[k]).abs()
        res[mode][k] = [float(d.max()), float(d.pow(2).mean().sqrt())]
print(json.dumps(res))
