#!/usr/bin/env python3
"""Where the packed evaluation loop's wall clock goes (dataset-shaped configs): staging copy, H2D, forward_videos, per phase
and end to end, for an XD-sized bf16 list and a UCF-sized f32 list.  python tools/ragged_profile.py"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import iefvad_amd
from iefvad_amd import harness, synth


def build(nvid, total, seed, lo=16, hi=8000):
    lengths = synth.lognormal_lengths(seed, nvid, total, lo=lo, hi=hi)
    items = []
    for i, n in enumerate(lengths):
        img, ev = synth.make_video(seed, i, int(n))
        ci, _ = harness.process_split(img, 256)
        ce, _ = harness.process_split(ev, 256)
        items.append((torch.from_numpy(ci).unsqueeze(0), torch.from_numpy(ce).unsqueeze(0), ("Normal",), torch.tensor([int(n)])))
    return lengths, items


def main():
    dev = torch.device("cuda:0")
    torch.set_num_threads(harness.host_cpu_share())
    for tag, nvid, total, seed, compute, K, bc in (("xd", 753, 145000, 2, "bf16", 10, 128), ("ucf", 290, 69500, 1, "f32", 10, 64),
                                                   ("shang_msad", 438, 17732, 5, "bf16", 5, 32)):
        lengths, items = build(nvid, total, seed, *((4, 400) if tag == "shang_msad" else (16, 8000)))
        args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=K, lambda_ref=0.5, noise_model="StudentT", nu=8)
        model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args, compute=compute, outputs="scores")
        model.load_state_dict(synth.make_state_dict(17, 768, 2, K))
        model = model.to(dev).eval()
        n = int(lengths.sum())
        for bcs in (bc, 2 * bc, 4 * bc):
          for lanes in (1, 2, 3):
            best = 1e9
            for rep in range(4):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                harness.score_loader(model, items, 256, dev, "ucfcrime", batch_chunks=bcs, lanes=lanes)
                torch.cuda.synchronize()
                dt = time.perf_counter() - t0
                best = min(best, dt) if rep else 1e9
            print(f"{tag} {compute} batch_chunks={bcs} lanes={lanes}: {n / best / 1e6:.2f} M snippets/s ({best * 1e3:.1f} ms best of 3, last {dt * 1e3:.1f})", flush=True)
        # phases, one lane, everything synchronised in between
        st = harness._RowStager(dev)
        rows = [harness._unpack_rows(it, 256, "ucfcrime", None) for it in items]
        t0 = time.perf_counter()
        rows = [harness._unpack_rows(it, 256, "ucfcrime", None) for it in items]
        t_unpack = time.perf_counter() - t0
        imgs, evs, lens = [r[0] for r in rows], [r[1] for r in rows], [r[3] for r in rows]
        for rep in range(2):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            di, de = st.upload(imgs, evs, torch.float32, lens)
            t_stage = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_h2d = time.perf_counter() - t0
            t1 = time.perf_counter()
            with torch.no_grad():
                model.forward_videos(di, de, lens)
            t_enq = time.perf_counter() - t1
            torch.cuda.synchronize()
            t_fwd = time.perf_counter() - t1
        chunks = sum((n_ // 256 + (1 if n_ % 256 else 0)) if n_ >= 256 else 1 for n_ in lens)
        print(f"   phases (whole list as ONE batch): unpack {t_unpack * 1e3:.1f} ms, staging copy {t_stage * 1e3:.1f} ms, "
              f"copy + H2D {t_h2d * 1e3:.1f} ms, forward_videos enqueue {t_enq * 1e3:.1f} ms, forward done {t_fwd * 1e3:.1f} ms; "
              f"{n} valid rows in {chunks} chunks ({chunks * 256} chunk rows)", flush=True)
        del model


if __name__ == "__main__":
    main()
