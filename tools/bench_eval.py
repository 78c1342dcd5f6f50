#!/usr/bin/env python3
"""End-to-end evaluation throughput on a UCF-Crime-shaped synthetic .npy set (BASELINE config 2: 290 videos,
~69.5 k snippets, fp32 files on local disk): the callers either side of the forward,
measured on the GPU box.  Prints one JSON object.

  per_video      the reference's pattern: DataLoader(batch_size=1), one forward per video (test.py:76-117)
  batched        same loader, chunks of consecutive videos packed into forwards of >= 256 chunks
  streaming      harness.evaluate_files: threaded .npy reads -> pinned staging -> async H2D -> batched forward
                 -> device-side AUC/AP
  cpu_oracle     the CPU oracle in the per-video pattern on the first videos (bounded sample)
  metric_tail    sklearn roc_auc_score + average_precision_score on the x16 repeat vs harness.device_auc_ap
  sweep          robustness sweep levels (test2.py) with and without the clean-forward cache
"""
import argparse
import json
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

import iefvad_amd  # noqa: E402
from iefvad_amd import harness, synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--videos", type=int, default=290)
    ap.add_argument("--snippets", type=int, default=69500)
    ap.add_argument("--compute", default="f32")
    ap.add_argument("--cpu-videos", type=int, default=24)
    a = ap.parse_args()
    torch.set_num_threads(harness.host_cpu_share())     # the box exposes 256 hardware threads behind a 16-CPU quota
    seed = 1
    lengths = synth.lognormal_lengths(seed, a.videos, a.snippets)
    abnormal = [c for c in synth.UCF_CLASSES if c != 'Normal']
    classes = ['Normal' if i % 2 == 0 else abnormal[(i // 2) % 13] for i in range(a.videos)]
    total = int(lengths.sum())
    gt = synth.make_gt(seed, total)
    tmp = tempfile.mkdtemp(prefix="iefvad_eval_", dir="/tmp")
    rows = []
    t0 = time.perf_counter()
    for i, (n, c) in enumerate(zip(lengths, classes)):
        img, ev = synth.make_video(seed, i, int(n))
        d = os.path.join(tmp, "rgb", c)
        os.makedirs(d, exist_ok=True)
        os.makedirs(d.replace("rgb", "event_thr_10"), exist_ok=True)
        p = os.path.join(d, f"v{i:04d}__5.npy")
        np.save(p, img)
        np.save(p.replace("rgb", "event_thr_10"), ev)
        rows.append((p, c))
    csv = os.path.join(tmp, "test.csv")
    with open(csv, "w") as f:
        f.write("path,label\n" + "".join(f"{p},{c}\n" for p, c in rows))
    gen_s = time.perf_counter() - t0
    args = argparse.Namespace(dataset="ucfcrime", visual_length=256, test_list=csv, exp_name="eval")
    margs = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5,
                               noise_model="StudentT", nu=8)
    sd = synth.make_state_dict(7)

    def gpu_model(outputs):
        m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", margs, outputs=outputs, compute=a.compute)
        m.load_state_dict(sd)
        return m.to("cuda:0").eval()

    out = {"videos": a.videos, "snippets": total, "file_bytes": int(total * 768 * 4 * 2), "compute": a.compute,
           "dataset_write_s": gen_s}
    m_full, m_scores = gpu_model("full"), gpu_model("scores")
    # warm-up (library load, weights, workspace)
    harness.score_loader(m_scores, list(harness.get_test_loader(args))[:2], 256, "cuda:0")
    torch.cuda.synchronize()

    def timed(fn):
        torch.cuda.synchronize()
        t = time.perf_counter()
        r = fn()
        torch.cuda.synchronize()
        return r, time.perf_counter() - t

    (s_pv, _, _, _), t_pv = timed(lambda: harness.score_loader(m_full, harness.get_test_loader(args), 256, "cuda:0"))
    out["per_video_full_outputs"] = {"seconds": t_pv, "snippets_per_s": total / t_pv}
    (s_pv2, _, _, _), t_pv2 = timed(lambda: harness.score_loader(m_scores, harness.get_test_loader(args), 256, "cuda:0"))
    out["per_video_scores_only"] = {"seconds": t_pv2, "snippets_per_s": total / t_pv2}
    (s_b, _, _, _), t_b = timed(lambda: harness.score_loader(m_scores, harness.get_test_loader(args), 256, "cuda:0",
                                                              batch_chunks=256))
    out["batched"] = {"seconds": t_b, "snippets_per_s": total / t_b}
    res, t_s = timed(lambda: harness.evaluate_files(args, m_scores, gt, "cuda:0"))
    out["streaming"] = {"seconds": t_s, "snippets_per_s": total / t_s, "phases": res["seconds"], "roc": res["roc"], "ap": res["ap"]}
    res2, t_s2 = timed(lambda: harness.evaluate_files(args, m_scores, gt, "cuda:0"))
    out["streaming_second_pass"] = {"seconds": t_s2, "snippets_per_s": total / t_s2, "phases": res2["seconds"]}
    r3, t3 = timed(lambda: harness.evaluate_files(args, m_scores, gt, "cuda:0"))
    out["streaming_third_pass"] = {"seconds": t3, "snippets_per_s": total / t3, "phases": r3["seconds"]}
    a1, a2, a3 = np.concatenate(s_pv2), np.concatenate(s_b), np.concatenate(res["scores"])
    out["max_score_diff_between_patterns"] = float(max(np.abs(a1 - a2).max(), np.abs(a1 - a3).max()))

    # CPU oracle, per-video pattern, bounded sample
    from oracle import iefvad_oracle as orc
    oracle = orc.OracleMMFMIL(sd, orc.OracleConfig())
    sub = list(harness.get_test_loader(args))[: a.cpu_videos]
    nsub = int(sum(int(it[3]) for it in sub))
    t = time.perf_counter()
    s_cpu, _, _, _ = harness.score_loader(oracle, sub, 256, "cpu")
    t_cpu = time.perf_counter() - t
    out["cpu_oracle_per_video"] = {"seconds": t_cpu, "snippets": nsub, "snippets_per_s": nsub / t_cpu,
                                   "cores": torch.get_num_threads(),
                                   "max_score_diff_vs_gpu": float(np.abs(np.concatenate(s_cpu) - a1[:nsub]).max())}

    # metric tail
    from sklearn.metrics import average_precision_score, roc_auc_score
    t = time.perf_counter()
    r_sk = roc_auc_score(gt, np.repeat(a1, 16))
    ap_sk = average_precision_score(gt, np.repeat(a1, 16))
    t_sk = time.perf_counter() - t
    sd_dev, gt_dev = torch.from_numpy(a1).cuda(), torch.from_numpy(gt).cuda()
    harness.device_auc_ap(sd_dev, gt_dev)
    (r_dev, ap_dev), t_dev = timed(lambda: harness.device_auc_ap(sd_dev, gt_dev))
    out["metric_tail"] = {"sklearn_s": t_sk, "device_s": t_dev, "auc_diff": abs(r_sk - r_dev), "ap_diff": abs(ap_sk - ap_dev)}

    # robustness sweep: three levels of one modality, clean forward cached vs recomputed
    items = list(harness.get_test_loader(args))[:60]
    nsw = int(sum(int(it[3]) for it in items))
    for tag, use_cache in (("cached_clean", True), ("recomputed_clean", False)):
        cache = {} if use_cache else None
        torch.manual_seed(0)
        t = time.perf_counter()
        for lvl in (0.05, 0.2, 0.5):
            harness.run_perturbation_test(args, m_full, items, gt, "cuda:0", sigma_img=lvl, sigma_ev=0, clean_cache=cache)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t
        out["sweep_" + tag] = {"seconds": dt, "levels": 3, "videos": len(items), "snippets": nsw,
                               "forwarded_snippets_per_s": nsw * (3 + (1 if use_cache else 3)) / dt}
    print(json.dumps(out))
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
