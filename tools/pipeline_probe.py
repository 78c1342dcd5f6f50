#!/usr/bin/env python3
"""Packed evaluation loop, XD-sized bf16 list: each stage of the pipeline alone and in combination (which one bounds the wall
clock).  python tools/pipeline_probe.py"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iefvad_amd
from iefvad_amd import harness, synth
from tools.ragged_profile import build


def main():
    dev = torch.device("cuda:0")
    torch.set_num_threads(harness.host_cpu_share())
    lengths, items = build(753, 145000, 2)
    args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args, compute="bf16", outputs="scores")
    model.load_state_dict(synth.make_state_dict(17, 768, 2, 10))
    model = model.to(dev).eval()
    rows = [harness._unpack_rows(it, 256, "ucfcrime", None) for it in items]
    batches, cur, cc = [], [], 0
    for r in rows:
        cur.append(r)
        n = r[3]
        cc += (n // 256 + (1 if n % 256 else 0)) if n >= 256 else 1
        if cc >= 128:
            batches.append(cur); cur, cc = [], 0
    if cur:
        batches.append(cur)
    nb = len(batches)
    total = int(lengths.sum())
    # one pinned pair per batch, staged once
    stagers = [harness._RowStager(dev, slots=1) for _ in range(nb)]
    def stage_all():
        return [stagers[i].stage([r[0] for r in b], [r[1] for r in b], torch.float32, [r[3] for r in b]) for i, b in enumerate(batches)]
    staged = stage_all()
    for st in stagers:
        st.events = [None]
    t0 = time.perf_counter(); staged = stage_all(); t_gather = time.perf_counter() - t0
    lens = [[r[3] for r in b] for b in batches]
    lanes = model.lanes(2)
    streams = [torch.cuda.Stream(device=dev) for _ in range(2)]

    def h2d_only():
        out = []
        for i in range(nb):
            out.append(stagers[i].send(staged[i]))
        return out
    def timed(fn, reps=3):
        best = 1e9
        for _ in range(reps):
            torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(); torch.cuda.synchronize()
            best = min(best, time.perf_counter() - t0)
        return best, r
    t_h2d, devrows = timed(h2d_only)
    def fwd_only(nl):
        with torch.no_grad():
            for i in range(nb):
                k = i % nl
                with torch.cuda.stream(streams[k]):
                    lanes[k].forward_videos(devrows[i][0], devrows[i][1], lens[i])
    for s in streams:
        s.wait_stream(torch.cuda.current_stream(dev))
    t_f1, _ = timed(lambda: fwd_only(1))
    t_f2, _ = timed(lambda: fwd_only(2))
    def h2d_fwd(nl):
        with torch.no_grad():
            for i in range(nb):
                k = i % nl
                with torch.cuda.stream(streams[k]):
                    a, b = stagers[i].send(staged[i])
                    lanes[k].forward_videos(a, b, lens[i])
    t_p1, _ = timed(lambda: h2d_fwd(1))
    t_p2, _ = timed(lambda: h2d_fwd(2))
    def full(nl):
        harness.score_loader(model, items, 256, dev, "ucfcrime", batch_chunks=128, lanes=nl)
    t_full2, _ = timed(lambda: full(2), reps=4)
    t0 = time.perf_counter(); rows2 = [harness._unpack_rows(it, 256, "ucfcrime", None) for it in items]; t_unpack = time.perf_counter() - t0
    print(f"XD-sized list, {total} snippets in {nb} batches of >= 128 chunks (bf16):")
    print(f"  unpack (Python, per video)            {t_unpack * 1e3:6.1f} ms")
    print(f"  staging copies alone (all batches)    {t_gather * 1e3:6.1f} ms")
    print(f"  H2D alone (pre-staged pinned rows)    {t_h2d * 1e3:6.1f} ms")
    print(f"  forward alone, rows resident, 1 lane  {t_f1 * 1e3:6.1f} ms;  2 lanes {t_f2 * 1e3:6.1f} ms")
    print(f"  H2D + forward, pre-staged, 1 lane     {t_p1 * 1e3:6.1f} ms;  2 lanes {t_p2 * 1e3:6.1f} ms")
    print(f"  score_loader end to end, 2 lanes      {t_full2 * 1e3:6.1f} ms  ({total / t_full2 / 1e6:.2f} M snippets/s)")


if __name__ == "__main__":
    main()
