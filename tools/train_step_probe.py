#!/usr/bin/env python3
"""Time whole training steps (forward, loss, backward, AdamW) of the reference's UCF configuration on the MI355X:
B = 2 x batch_size chunks of [256, 768] (main.py:68: batch_size 64 -> 128 chunks), K = 10, L = 2, StudentT nu = 8, lr 2e-5.

    python3 tools/train_step_probe.py [--chunks 128] [--steps 5] [--compute f32|bf16x6] [--dropout 0.1]

Prints one JSON line: ms per step, snippets/s, the forward / backward+optimiser split (torch.cuda events on the current stream,
which is the stream every launch goes to), algorithmic FLOPs (3 x the forward's 50.33 MFLOP per snippet: dX and dW of every projection,
five products instead of two in attention) and the fraction of the fp32 MFMA peak.  Run under `rocprofv3 --kernel-trace --stats`
for the per-kernel picture (tools/collect_profiles.sh does)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

import iefvad_amd  # noqa: E402
from iefvad_amd import losses, synth  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--chunks", type=int, default=128)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--compute", default="f32", choices=["f32", "bf16x6"])
    p.add_argument("--dropout", type=float, default=0.1)
    p.add_argument("--K", type=int, default=10)
    a = p.parse_args()
    dev = torch.device("cuda", 0)
    args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=a.K, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args, compute=a.compute)
    model.load_state_dict(synth.make_state_dict(0, 768, 2, a.K))
    model = model.to(dev).train()
    for m in list(model.temporal.image_attn_layers) + list(model.temporal.event_attn_layers):
        m.dropout = a.dropout
    opt = losses.AdamW(model.parameters(), lr=2e-5)
    B = a.chunks
    gen = torch.Generator(device=dev).manual_seed(1)
    img = torch.randn(B, 256, 768, device=dev, generator=gen) * 0.45
    ev = torch.randn(B, 256, 768, device=dev, generator=gen) * 0.45
    labels = torch.zeros(B, 14, device=dev)
    labels[: B // 2, 0] = 1
    labels[B // 2:, 3] = 1
    lengths = torch.full((B,), 256, dtype=torch.int64, device=dev)
    ev_f = [torch.cuda.Event(enable_timing=True) for _ in range(3 * (a.steps + a.warmup))]
    fwd = bwd = 0.0
    t0 = None
    for s in range(a.warmup + a.steps):
        if s == a.warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        e0, e1, e2 = ev_f[3 * s: 3 * s + 3]
        e0.record()
        out = model(img, ev, None, None, lengths)
        total = losses.training_loss(out, labels, lengths, "StudentT", 8, 1.0, 1.0)
        e1.record()
        opt.zero_grad()
        total.backward()
        opt.step()
        e2.record()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    for s in range(a.warmup, a.warmup + a.steps):
        e0, e1, e2 = ev_f[3 * s: 3 * s + 3]
        fwd += e0.elapsed_time(e1) / a.steps
        bwd += e1.elapsed_time(e2) / a.steps
    snips = B * 256
    fwd_flops = (26_740_224 + a.K * 2_359_296) * snips
    # backward: dX + dW for every projection (2 x), attention 4 products for the forward's 2 (and the first layer's dX is skipped)
    step_flops = 3 * fwd_flops
    print(json.dumps({"workload": f"training step, {B} chunks x 256 x 768 (ucf_train.py: 2 x batch_size 64), K={a.K}, L=2, StudentT, "
                                  f"attention dropout {a.dropout}, AdamW lr 2e-5, compute={a.compute}",
                      "ms_per_step": dt * 1e3, "snippets_per_s": snips / dt, "forward_plus_loss_ms": fwd, "backward_plus_adamw_ms": bwd,
                      "algorithmic_tflop_per_step": step_flops / 1e12, "achieved_tflops": step_flops / dt / 1e12,
                      "frac_of_bf16_mfma_peak": step_flops / dt / 1e12 / 2500.0, "loss": float(total.detach()),
                      "train_buffer_gib": model._handle and iefvad_amd.lib.load_library().iefvad_train_workspace_bytes(model._handle, B) / 2**30}))


if __name__ == "__main__":
    main()
