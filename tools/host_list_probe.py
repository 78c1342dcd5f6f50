#!/usr/bin/env python3
"""Where the wall clock of the packed evaluation goes when the list walk is inside the library (iefvad_forward_videos_host):
BASELINE configs 3 (XD-sized, 753 videos, 145 k snippets) and 5 (Shang + MSAD lists, 438 videos, 17.7 k snippets, K = 5), bf16.

    python3 tools/host_list_probe.py [--threads N] [--batch-chunks C]

Per list: the whole score_loader pass (median of 5), and its phases measured separately -- Python collecting the items, the library
call (returns when every host row has been read and every pass is enqueued), the wait for the device, sigmoid + D2H + per-video split."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench  # noqa: E402
import iefvad_amd  # noqa: E402
from iefvad_amd import harness, synth  # noqa: E402


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--threads", type=int, default=0)
    p.add_argument("--batch-chunks", type=int, default=128)
    p.add_argument("--prelude", default="none", choices=["none", "ucf", "threads"],
                   help="what runs in the process first: bench.ucf_eval (as in a default bench run), or only torch.set_num_threads(host share)")
    p.add_argument("--wire-bf16", action="store_true", help="round the fp32 rows to bf16 while staging (wire_dtype = BF16)")
    a = p.parse_args()
    wire = torch.bfloat16 if a.wire_bf16 else None
    dev = torch.device("cuda", 0)
    if a.prelude == "ucf":
        ba = bench.parse([])
        m0 = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5, noise_model="StudentT", nu=8)
        bench.ucf_eval(synth.make_state_dict(0, 768, 2, 10), m0, dev, ba)
    elif a.prelude == "threads":
        torch.set_num_threads(harness.host_cpu_share())
    for tag, parts, wseed, K in (("config3_xd", bench.xd_parts(), 17, 10), ("config5_shang_msad", bench.config5_parts(), 19, 5)):
        items = []
        for dataset, lengths, classes, gt, seed, nk in parts:
            for i, n in enumerate(lengths):
                img, ev = synth.make_video(seed, i, int(n))
                ci, _ = harness.process_split(img, 256)
                ce, _ = harness.process_split(ev, 256)
                items.append((torch.from_numpy(ci).unsqueeze(0), torch.from_numpy(ce).unsqueeze(0), (classes[i],), torch.tensor([int(n)])))
        total = sum(int(it[3]) for it in items)
        margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=K, lambda_ref=0.5, noise_model="StudentT", nu=8)
        model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs="scores", compute="bf16")
        model.load_state_dict(synth.make_state_dict(wseed, 768, 2, K))
        model = model.to(dev).eval()
        for _ in range(2):
            harness.score_loader(model, items, 256, dev, "ucfcrime", batch_chunks=a.batch_chunks, wire_bf16=a.wire_bf16)
        torch.cuda.synchronize()
        whole = []
        for _ in range(5):
            t0 = time.perf_counter()
            harness.score_loader(model, items, 256, dev, "ucfcrime", batch_chunks=a.batch_chunks, wire_bf16=a.wire_bf16)
            torch.cuda.synchronize()
            whole.append(time.perf_counter() - t0)
        ph = []
        for _ in range(5):
            t0 = time.perf_counter()
            un = [harness._unpack_rows(it, 256, "ucfcrime", None) for it in items]
            t1 = time.perf_counter()
            out = model.forward_videos_host([u[0] for u in un], [u[1] for u in un], [u[3] for u in un], batch_chunks=a.batch_chunks,
                                            host_threads=a.threads, wire_dtype=wire)
            t2 = time.perf_counter()
            torch.cuda.synchronize()
            t3 = time.perf_counter()
            prob = torch.sigmoid(out["logits"]).cpu().numpy()
            offs = np.cumsum([0] + [u[3] for u in un])
            per = [prob[offs[i]:offs[i + 1]] for i in range(len(un))]
            t4 = time.perf_counter()
            ph.append((t1 - t0, t2 - t1, t3 - t2, t4 - t3))
        med = lambda xs: float(sorted(xs)[len(xs) // 2])
        print(json.dumps({"list": tag, "videos": len(items), "snippets": total, "score_loader_ms": med(whole) * 1e3,
                          "snippets_per_s": total / med(whole),
                          "phases_ms": {"python_collect": med([x[0] for x in ph]) * 1e3, "library_call": med([x[1] for x in ph]) * 1e3,
                                        "device_wait": med([x[2] for x in ph]) * 1e3, "sigmoid_d2h_split": med([x[3] for x in ph]) * 1e3},
                          "wire": "bf16" if a.wire_bf16 else "fp32", "host_threads": a.threads or "library default", "batch_chunks": a.batch_chunks, "prelude": a.prelude,
                          "torch_threads": torch.get_num_threads()}))
        del model


if __name__ == "__main__":
    main()
