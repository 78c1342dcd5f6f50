"""GPU probe for iefvad_forward_scaled on fp16 rows, scores-only outputs (the failing test's configuration)."""
import sys, os, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import iefvad_amd
from iefvad_amd import synth
a = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
for compute in ("f32", "bf16x6"):
  for outputs in ("scores", "full"):
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", a, outputs=outputs, compute=compute)
    m.load_state_dict(synth.make_state_dict(5))
    m = m.to("cuda:0").eval()
    B = 9
    img, ev = synth.make_inputs(77, B)
    img, ev = torch.from_numpy(img).half().cuda(), torch.from_numpy(ev).half().cuda()
    gen = torch.Generator().manual_seed(1)
    sc = torch.ones(B, 256)
    for b in range(B):
        sc[b, torch.randperm(256, generator=gen)[:77]] = 0.01
    sc = sc.reshape(-1).cuda()
    with torch.no_grad():
        got = m(img, ev, None, None, None, row_scale=(sc, None))
        got2 = m(img, ev, None, None, None, row_scale=(sc, None))
        scaled = img.clone(); rows = sc.reshape(B, 256) != 1; scaled[rows] = scaled[rows] * 0.01
        want = m(scaled, ev, None, None, None)
        want2 = m(scaled, ev, None, None, None)
    d = (got["logits"] - want["logits"]).abs().reshape(B, 256)
    print(compute, outputs, "run-to-run (scaled):", float((got["logits"] - got2["logits"]).abs().max()), "run-to-run (plain):",
          float((want["logits"] - want2["logits"]).abs().max()), "| scaled vs torch-scaled: max", float(d.max()), "rows differing per chunk", (d > 0).sum(dim=1).tolist())
