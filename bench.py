#!/usr/bin/env python3
"""bench.py -- snippets/sec of the IEF-VAD fusion-inference forward on MI355X.

    python bench.py                         # N = 1
    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`python bench.py --gpus N` from a plain shell works for N > 1 as well: before anything touches a GPU the process starts
`python -m torch.distributed.run` with N ranks as a CHILD, relays its output and exits with its code.

A "step" is one pass of the hot path (MMFMIL.forward through libiefvad.so, scores-only outputs) over one batch of
synthetic [B, T=256, d=768] fp32 image + event feature blocks that are already resident in HBM, followed -- when
N > 1 -- by the single RCCL all-gather of per-snippet scores (`iefvad_gather_scores`, librccl over xGMI).
Workload: BASELINE.json config 4, B = 8192 chunks (2,097,152 snippets) IN TOTAL; rank r holds and scores chunks
[r*B/N, (r+1)*B/N), so scaling is "strong" and `value` is the whole-job rate.  `--scaling weak` keeps
8192 chunks per GPU instead; a default N > 1 run also measures that variant and reports it as `weak_scaling`.
Weights are seeded synthetic tensors of the reference architecture (K=10, nu=8, StudentT).

Arithmetic (--compute): the default, bf16x6, is fp32-ACCURATE: every fp32 operand of a dense projection is split
exactly into three bf16 terms and the product is accumulated in fp32 from six bf16 MFMA products (csrc/gemm_split.h);
it is held to the fp32 parity gates (tests/test_gpu_bf16x6.py).  The same workload is timed in the same run in the
other arithmetic modes and reported as `f32_mfma_mode` (fp32 MFMA instruction), `bf16_mode` (BASELINE configs 3 / 5:
bf16-rounded operands, fp32 accumulation and state) and `fp16x3_mode`.

The JSON line also carries
  roofline      the dense-projection GEMM kernel (~85 % of device time).  `achieved` = ALGORITHMIC FLOPs (SURVEY.md 8d:
                47,185,920 per snippet at K=10) of the rows one launch processes / that kernel's average launch
                duration, timed with hipEvents on the launch stream inside the timed region; `peak` = dense MFMA peak of
                the pipe the kernel runs on (MI355X_MICROARCH.md: bf16 2500, fp32 157.3 TFLOP/s).  For the emulated
                modes `mfma_pipe_util` is the EXECUTED-product rate / peak (6 bf16 products per algorithmic one).
  cpu_baseline  the CPU oracle (oracle/iefvad_oracle.py, torch CPU ops on the host cores) on a bounded sample of the
                same workload, rank 0 at N=1 only.  A reported baseline, not the target.
  xd_eval, shang_msad_eval   BASELINE configs 3 and 5 (XD-Violence-sized; ShanghaiTech + MSAD sized with K = 5), bf16 projections
                with fp32 state, packed evaluation loop: snippets/s, AUC / AP, and their distance from the fp32 CPU oracle on a
                bounded prefix of the list.
  train_step    SURVEY 8f-4: whole training steps (train-mode forward, loss, backward, AdamW) at the reference's UCF batch size.
  ucf_eval      BASELINE config 2 (UCF-Crime-sized synthetic set, 290 videos, ~69.5 k snippets) through harness.test:
                the reference's per-video call pattern and the cross-video batched pattern, snippets/s wall clock
                including H2D, |dAUC| against the CPU oracle on a bounded sample, and the oracle's own rate.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

T, D, H, L, K_STEPS = 256, 768, 8, 2, 10
GEMM_FLOPS_PER_SNIPPET = 4 * (2 * D * 3 * D) + 4 * (2 * D * D) + 4 * (2 * D * D) + K_STEPS * 2 * (2 * D * D)   # 47,185,920
TOTAL_FLOPS_PER_SNIPPET = 26_740_224 + K_STEPS * 2_359_296
PEAK_F32_MFMA_TFLOPS = 157.3                                    # MI355X_MICROARCH.md, chip-level table
PEAK_BF16_MFMA_TFLOPS = 2500.0                                  # dense bf16 / fp16 MFMA (same table)
PROFILE_ROUNDS = ("r05", "r04")                                # profiles/<round>_*_hbm_traffic.json this build's kernels were measured in, newest first
GEMM_KERNEL = {"f32": "iefvad_gemm_f32_t256_kernel",
               "bf16": "iefvad_inproj_chain_f32in_kernel / iefvad_inproj_chain_bf16_kernel (in_proj; the first layer rounds the fp32 "
                       "rows to bf16 itself) + iefvad_refine_chain_bf16_kernel (the 2K refinement projections + scorer, one launch) + "
                       "iefvad_heads_pchain_bf16_kernel (heads + fusion, persistent) + iefvad_outproj_ln_pchain_bf16_kernel (out_proj + LayerNorm, "
                       "persistent)",
               "bf16x6": "iefvad_gemm_split_n128_kernel", "fp16x3": "iefvad_gemm_split_f16_n128_kernel"}
PRODUCTS_PER_MAC = {"f32": 1.0, "bf16": 1.0, "bf16x6": 6.0, "fp16x3": 3.0}
DTYPE = {"f32": "f32", "bf16": "bf16",
         "bf16x6": "f32 emulated: exact 3-term bf16 split of both fp32 operands, 6 bf16 MFMA products, fp32 accumulate",
         "fp16x3": "near-f32 (22-bit products): 2-term fp16 split of both scaled fp32 operands, 3 fp16 MFMA products, "
                   "fp32 accumulate"}


def parse(argv=None):
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--chunks", type=int, default=8192,
                   help="chunks [256,768] per step: in total (strong scaling, the default) or per GPU (--scaling weak)")
    p.add_argument("--scaling", default="strong", choices=["strong", "weak"])
    p.add_argument("--micro-batch", type=int, default=0, help="chunks per internal pass (0 = library default)")
    p.add_argument("--outputs", default="scores", choices=["scores", "full"])
    p.add_argument("--compute", default="bf16x6", choices=["f32", "bf16", "bf16x6", "fp16x3"],
                   help="arithmetic of the dense projections: bf16x6 (default) = fp32-accurate, six bf16 MFMA products of the "
                        "exact 3-term split of each fp32 operand, fp32 accumulation (held to the f32 mode's parity gates); "
                        "f32 = fp32 MFMA; bf16 = bf16-rounded operands (reduced precision, BASELINE config 3)")
    p.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                   help="nccl (= RCCL over xGMI) for real multi-GPU runs; gloo only to rehearse N>1 on a one-GPU box")
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--no-extra-modes", action="store_true",
                   help="skip the side passes (f32_mfma_mode, bf16_mode, fp16x3_mode, weak_scaling)")
    p.add_argument("--no-ucf-eval", action="store_true", help="skip the dataset-shaped blocks (ucf_eval, xd_eval, shang_msad_eval)")
    p.add_argument("--cpu-seconds", type=float, default=10.0, help="budget of each CPU-oracle sample")
    p.add_argument("--plumbing-only", action="store_true",
                   help="launcher self-test for machines without a GPU (tests/test_bench_launcher_cpu.py): NO forward runs; "
                        "every rank fabricates the scores of its shard as a ramp of global snippet indices on the CPU and the "
                        "launch -> shard -> gather (gloo) -> max-over-ranks -> JSON path is exercised; `value` is null")
    p.add_argument("--plumbing-comm", default="none", choices=["none", "fail", "hang"],
                   help="--plumbing-only rehearsal of the gather-transport handshake (harness.ScoreComm): the library's communicator "
                        "create fails ('fail') or never returns ('hang') on the LAST rank; every rank must end up on the "
                        "torch.distributed transport, with the reason in the line's `gather` field, and the job must finish")
    p.add_argument("--plumbing-comm-timeout", type=float, default=5.0)
    return p.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------
# N > 1 from a plain shell: become the parent of a torch.distributed.run job (no GPU call has happened yet)
# ----------------------------------------------------------------------------------------------------------------
def launch_ranks(a) -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")       # dmabuf IPC: RCCL across processes needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", "4")
    proc = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:                                 # rank 0's JSON line (and nothing else) arrives here
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


def host_cpu_share():
    from iefvad_amd.harness import host_cpu_share as f
    return f()


def cpu_baseline(sd, seconds):
    """The oracle (CPU restatement of the reference forward) on the host cores: B=8 chunks per call, the
    best-case batched pattern of SURVEY.md 8d, repeated for ~`seconds`."""
    import numpy as np
    import torch
    from iefvad_amd import synth
    from oracle import iefvad_oracle as orc
    torch.set_num_threads(host_cpu_share())
    cores = torch.get_num_threads()
    model = orc.OracleMMFMIL(sd, orc.OracleConfig(num_layers=L, num_refinement_steps=K_STEPS, nu=8))
    img, ev = synth.make_inputs(1234, 8)
    img, ev = torch.from_numpy(img), torch.from_numpy(ev)
    model(img, ev)                                   # warm-up
    times = []
    t_end = time.perf_counter() + seconds
    while len(times) < 3 or (time.perf_counter() < t_end and len(times) < 200):
        t0 = time.perf_counter()
        model(img, ev)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": 8 * T / med, "unit": "snippets/s", "cores": cores, "kind": "port",
            "sample": f"oracle forward, B=8 chunks (2048 snippets) per call, median of {len(times)} calls, fp32"}


def make_model(sd, margs, dev, a, compute, outputs=None):
    import iefvad_amd
    model = iefvad_amd.MMFMIL(14, D, T, D, H, L, 8, 10, 10, "cuda", margs, outputs=outputs or a.outputs,
                              micro_batch=a.micro_batch, compute=compute)
    model.load_state_dict(sd)
    return model.to(dev).eval()


def roofline_block(compute, stage, steps, rows_per_step):
    """algorithmic GEMM FLOPs of the rows a launch processes / the kernel's average launch duration,
    against the dense peak of the matrix pipe the kernel issues on."""
    gemm_ms = (stage["qkv_gemm_ms"] + stage["out_gemm_ms"] + stage["head_gemm_ms"] + stage["refine_gemm_ms"]) / steps
    launches = stage["gemm_launches"] / steps
    flops = GEMM_FLOPS_PER_SNIPPET * rows_per_step                     # algorithmic, per step, this rank
    achieved = flops / (gemm_ms * 1e-3) / 1e12
    peak = PEAK_F32_MFMA_TFLOPS if compute == "f32" else PEAK_BF16_MFMA_TFLOPS
    r = {"bound": "mfma", "achieved": achieved, "peak": peak, "unit": "TFLOP/s", "frac": achieved / peak,
         "traffic": None, "kernel": GEMM_KERNEL[compute],
         "pipe": "fp32 MFMA (v_mfma_f32_32x32x2_f32)" if compute == "f32" else
                 ("fp16 MFMA" if compute == "fp16x3" else "bf16 MFMA (v_mfma_f32_16x16x32_bf16)"),
         "launches_per_step": launches, "avg_launch_ms": gemm_ms / launches,
         "algorithmic_flops_per_launch": flops / launches,
         "definition": "achieved = algorithmic GEMM FLOPs (SURVEY 8d, 47,185,920 per snippet) / sum of the kernel's "
                       "launch durations (hipEvents on the launch stream, inside the timed region)"}
    if compute == "bf16":
        pure_flops = (4 * (2 * D * 3 * D) + K_STEPS * 2 * (2 * D * D)) * rows_per_step      # in_proj + refinement launches only
        pure_ms = (stage["qkv_gemm_ms"] + stage["refine_gemm_ms"]) / steps
        r["achieved_pure_projection_launches"] = pure_flops / (pure_ms * 1e-3) / 1e12
        r["note"] = ("the 25 projections of a pass run as 6 launches of row-block kernels (a workgroup keeps 64 rows in LDS and every "
                     "wave streams its own weight columns): 2 x in_proj, 2 x iefvad_outproj_ln_pchain_bf16_kernel (out_proj + "
                     "residual + LayerNorm), iefvad_heads_pchain_bf16_kernel (both modalities' heads AND the precision-weighted "
                     "fusion) and ONE iefvad_refine_chain_bf16_kernel (the 2K refinement projections and the scorer, state on "
                     "chip); the fused launches are counted whole as GEMM time and the stand-alone LayerNorm / fusion / scorer "
                     "kernels do not run.  achieved_pure_projection_launches = the same quantity over the in_proj and "
                     "refinement-chain launches only")
    ppm = PRODUCTS_PER_MAC[compute]
    if ppm > 1:
        r["mfma_pipe_util"] = achieved * ppm / peak
        r["executed_products_per_algorithmic_mac"] = ppm
        r["achieved_vs_fp32_mfma_peak"] = achieved / PEAK_F32_MFMA_TFLOPS
        r["note"] = (f"fp32-accurate emulation: every algorithmic multiply-add is {int(ppm)} MFMA products, so frac cannot "
                     f"exceed {1 / ppm:.3f}; mfma_pipe_util counts the executed products")
    return r


def traffic_from_profiles(compute, rows_per_launch):
    """HBM-side bytes per GEMM launch as scalar keys of the roofline block.  NOT measured in this run: PMC counters need their
    own rocprofv3 passes, so `traffic` is read from the committed summary of such passes under profiles/ (tools/hbm_traffic.py
    writes it) -- and only from this round's file, at this launch size, whose recorded kernel list matches the kernels this build
    dispatches; otherwise it is null.  Provenance rides in sibling scalars (`traffic_source`, `traffic_head`, ...)."""
    none = {"traffic": None, "traffic_measured_in_this_run": False}
    name = {"f32": "gemm_hbm_traffic.json", "bf16x6": "gemm_split_hbm_traffic.json", "bf16": "gemm_bf16_hbm_traffic.json"}.get(compute)
    if not name:
        return none
    for rnd in PROFILE_ROUNDS:                           # newest first
        rel = os.path.join("profiles", f"{rnd}_{name}")
        path = os.path.join(ROOT, rel)
        if not os.path.exists(path):
            continue
        tj = json.load(open(path))
        if tj.get("rows_per_launch") != rows_per_launch:
            continue
        kernels = tj.get("kernels") or ([tj["kernel"]] if "kernel" in tj else [])
        if kernels and not all(k in GEMM_KERNEL[compute] for k in kernels):
            continue                                     # a PMC figure of another kernel set must not sit beside fresh timings
        return {"traffic": float(tj["traffic_bytes_per_launch"]), "traffic_unit": "bytes per launch (fabric FETCH_SIZE + WRITE_SIZE)",
                "traffic_source": rel, "traffic_head": tj.get("head"), "traffic_measured_in_this_run": False,
                "traffic_over_algorithmic": tj.get("traffic_over_algorithmic"), "traffic_kernels": ", ".join(kernels)}
    return none


def run_mode(model, img, ev, steps, warmup=1):
    """`steps` forwards of this rank's blocks (no gather) on the plain entry point for the rate, then ONE more through
    iefvad_forward_timed for the per-stage split (that entry synchronises the stream and brackets every launch with hipEvents:
    noise at 500 ms per step, not at the bf16 mode's 14 ms per micro-batch).  Returns (seconds, stage times of the extra step)."""
    import torch
    with torch.no_grad():
        for _ in range(warmup):
            model(img, ev, None, None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(img, ev, None, None, None)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        model(img, ev, None, None, None, timed=True)
        stage = dict(model.last_stage_times)
    return dt, stage


def extra_mode(sd, margs, dev, img, ev, a, compute, steps=3):
    """The same workload on this rank's GPU in another arithmetic mode, reported beside the headline."""
    model = make_model(sd, margs, dev, a, compute)
    dt, stage = run_mode(model, img, ev, steps)
    B = img.shape[0]
    value = B * T * steps / dt
    roof = roofline_block(compute, stage, 1, B * T)                  # stage times: one instrumented step outside the clock
    mb_eff = a.micro_batch if a.micro_batch > 0 else (256 if compute == "f32" else 1024)           # library defaults
    roof.update(traffic_from_profiles(compute, min(B, mb_eff) * T))
    out = {"compute": compute, "dtype": DTYPE[compute], "value": value, "unit": "snippets/s", "steps": steps,
           "ms_per_step": dt / steps * 1e3, "roofline": roof,
           "end_to_end_tflops": TOTAL_FLOPS_PER_SNIPPET * value / 1e12,
           "stage_ms_per_step": {k: v for k, v in stage.items() if k.endswith("_ms")},
           "stage_times_from": "one extra step through iefvad_forward_timed, outside the clock of `value`"}
    del model
    return out


def ucf_eval(sd, margs, dev, a):
    """BASELINE config 2: UCF-Crime-sized synthetic test set through `harness.test` (the counterpart of the reference's
    test(), test.py:46-212), fp32 arithmetic.  The loader is an in-memory list of DataLoader-shaped items (no file
    reads; the host-to-device copies of every video ARE inside the clock)."""
    import numpy as np
    import torch
    from iefvad_amd import harness, synth
    from oracle import iefvad_oracle as orc
    from sklearn.metrics import roc_auc_score
    seed, nvid, total_target = 1, 290, 69500
    lengths = synth.lognormal_lengths(seed, nvid, total_target)
    abnormal = [c for c in synth.UCF_CLASSES if c != "Normal"]
    classes = ["Normal" if i % 2 == 0 else abnormal[(i // 2) % 13] for i in range(nvid)]      # 150 / 140 as test.csv
    total = int(lengths.sum())
    gt = synth.make_gt(seed, total)
    items = []
    for i, n in enumerate(lengths):
        img, ev = synth.make_video(seed, i, int(n))
        ci, _ = harness.process_split(img, T)
        ce, _ = harness.process_split(ev, T)
        items.append((torch.from_numpy(ci).unsqueeze(0), torch.from_numpy(ce).unsqueeze(0), (classes[i],),
                      torch.tensor([int(n)])))
    args = argparse.Namespace(dataset="ucfcrime", visual_length=T)
    torch.set_num_threads(host_cpu_share())
    out = {"workload": f"BASELINE config 2: synthetic UCF-Crime-sized test set, {nvid} videos, {total} snippets, fp32 "
                       f"features held in host memory, K=10 nu=8 StudentT, harness.test end to end (H2D, forward, "
                       f"sigmoid, ordered scores, sklearn AUC/AP excluded from the clock)",
           "videos": nvid, "snippets": total}

    def timed_scores(model, batch_chunks, lanes):
        # warm-up: one full pass, so that every lane exists (library handle, repacked weights) and has captured its graphs
        harness.score_loader(model, items, T, dev, "ucfcrime", batch_chunks=batch_chunks, lanes=lanes)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        scores, cls, _, _ = harness.score_loader(model, items, T, dev, "ucfcrime", batch_chunks=batch_chunks, lanes=lanes)
        torch.cuda.synchronize()
        return scores, cls, time.perf_counter() - t0

    results = {}
    modes = [("f32", (("per_video", 0, 1), ("per_video_4lanes", 0, 4), ("batched", 64, 3)))]
    if a.compute != "f32":
        modes.append((a.compute, (("batched", 128, 3),)))
    for compute, patterns in modes:
        model = make_model(sd, margs, dev, a, compute, outputs="scores")
        for name, bc, lanes in patterns:
            scores, cls, dt = timed_scores(model, bc, lanes)
            res = harness.evaluate_scores(scores, cls, gt, "ucfcrime", verbose=False)
            key = f"{name}_{compute}"
            results[key] = scores
            out[key] = {"snippets_per_s": total / dt, "seconds": dt, "auc": res["roc"], "ap": res["ap"],
                        "ano_auc": res["ano_auc"],
                        "pattern": ("one forward per video (test.py:76-117)" + (f", consecutive videos on {lanes} HIP streams "
                                    "(same kernels, bit-identical scores)" if lanes > 1 else "")) if bc == 0 else
                                   f"valid rows of consecutive videos packed into iefvad_forward_videos calls of >= {bc} chunks "
                                   f"(chunker, NaN rule and [0:len] slicing on the device, tail on the valid rows), calls "
                                   f"round-robin on {lanes} HIP streams (staging + H2D of one overlaps the forward of another)"}
        if "per_video_4lanes_f32" in results and compute == "f32":
            out["per_video_4lanes_f32"]["bit_identical_to_per_video_f32"] = bool(all(
                np.array_equal(x, y) for x, y in zip(results["per_video_4lanes_f32"], results["per_video_f32"])))
        del model
    # CPU oracle in the reference's per-video pattern on a bounded prefix of the same list
    oracle = orc.OracleMMFMIL(sd, orc.OracleConfig(num_layers=L, num_refinement_steps=K_STEPS, nu=8))
    harness.score_loader(oracle, items[:1], T, "cpu", "ucfcrime")
    t0 = time.perf_counter()
    cpu_scores, nsub = [], 0
    for it in items:
        s, _, _, _ = harness.score_loader(oracle, [it], T, "cpu", "ucfcrime")
        cpu_scores += s
        nsub += len(s[0])
        if time.perf_counter() - t0 > a.cpu_seconds and len(cpu_scores) >= 8:
            break
    t_cpu = time.perf_counter() - t0
    nv = len(cpu_scores)
    cpu_cat = np.concatenate(cpu_scores)
    gt_sub = gt[:16 * nsub]
    auc_cpu = roc_auc_score(gt_sub, np.repeat(cpu_cat, 16))
    out["cpu_oracle_per_video"] = {"snippets_per_s": nsub / t_cpu, "seconds": t_cpu, "videos": nv, "snippets": nsub,
                                   "cores": torch.get_num_threads(), "kind": "port", "auc_on_sample": auc_cpu}
    for key, scores in results.items():
        gpu_cat = np.concatenate(scores[:nv])
        out[key]["max_abs_score_diff_vs_oracle_on_sample"] = float(np.abs(gpu_cat - cpu_cat).max())
        out[key]["abs_auc_diff_vs_oracle_on_sample"] = float(abs(roc_auc_score(gt_sub, np.repeat(gpu_cat, 16)) - auc_cpu))
        out[key]["x_cpu_oracle"] = out[key]["snippets_per_s"] / out["cpu_oracle_per_video"]["snippets_per_s"]
    return out


def dataset_eval(tag, parts, wseed, K, compute, dev, a, batch_chunks=128, lanes=2):
    """One of BASELINE's dataset-shaped configs (3: XD-Violence-sized, bf16; 5: the ShanghaiTech + MSAD test lists with their
    real ground truth, K = 5, bf16) through the evaluation loop.  `parts` = [(dataset, lengths, classes, gt, seed, normal_keys)]:
    one or more test lists scored in ONE packed pass over the same weights (synthetic features of the lists' sizes held in host
    memory as a DataLoader would deliver them: chunked and zero padded by process_split); the wall clock covers everything
    from those host tensors to the ordered score vectors (staging of the valid rows, H2D, `iefvad_forward_videos`, D2H).
    Metrics per list; scores / AUC / AP against the fp32 CPU oracle on a time-bounded, evenly spread sample of the videos."""
    import numpy as np
    import torch
    from iefvad_amd import harness, synth
    from oracle import iefvad_oracle as orc
    from sklearn.metrics import average_precision_score, roc_auc_score
    items, bounds = [], []
    for dataset, lengths, classes, gt, seed, normal_keys in parts:
        lo_i = len(items)
        for i, n in enumerate(lengths):
            img, ev = synth.make_video(seed, i, int(n))
            ci, _ = harness.process_split(img, T)
            ce, _ = harness.process_split(ev, T)
            items.append((torch.from_numpy(ci).unsqueeze(0), torch.from_numpy(ce).unsqueeze(0), (classes[i],), torch.tensor([int(n)])))
        bounds.append((lo_i, len(items)))
    nvid = len(items)
    total = int(sum(int(np.sum(p[1])) for p in parts))
    margs = argparse.Namespace(visual_layers=L, visual_head=H, num_refinement_steps=K, lambda_ref=0.5, noise_model="StudentT", nu=8)
    sd = synth.make_state_dict(wseed, D, L, K)
    model = make_model(sd, margs, dev, a, compute, outputs="scores")
    torch.set_num_threads(host_cpu_share())
    harness.score_loader(model, items, T, dev, "ucfcrime", batch_chunks=batch_chunks, lanes=lanes)          # warm-up: every lane
    torch.cuda.synchronize()
    dts = []
    for _ in range(7):                                 # a pass is 4 - 25 ms: seven of them, the median (single passes scatter by 2x on a shared host)
        t0 = time.perf_counter()
        scores, cls, _, _ = harness.score_loader(model, items, T, dev, "ucfcrime", batch_chunks=batch_chunks, lanes=lanes)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[len(dts) // 2]
    chunks = sum((int(n) // T + (1 if int(n) % T else 0)) if int(n) >= T else 1 for p in parts for n in p[1])
    out = {"workload": f"{tag}: {' + '.join(p[0] for p in parts)} test list(s), {nvid} videos, {total} snippets ({chunks} chunks = "
                       f"{chunks * T} chunk rows), K={K}, projections={compute}; valid rows of consecutive videos packed into "
                       f"iefvad_forward_videos calls of >= {batch_chunks} chunks, round-robin on {lanes} HIP streams",
           "videos": nvid, "snippets": total, "chunk_rows": chunks * T, "compute": compute, "snippets_per_s": total / dt,
           "seconds": dt, "seconds_all_passes": dts, "lists": {}}
    wire_scores = None
    if compute == "bf16":
        # the throughput mode's down-conversion on the wire (include/iefvad.h `wire_dtype`): the gather threads round the
        # fp32 rows to bf16 while staging, half the bytes cross PCIe.  Reported BESIDE the fp32-wire figure above, never as it.
        harness.score_loader(model, items, T, dev, "ucfcrime", batch_chunks=batch_chunks, lanes=lanes, wire_bf16=True)
        torch.cuda.synchronize()
        wdts = []
        for _ in range(7):
            t0 = time.perf_counter()
            wire_scores, _, _, _ = harness.score_loader(model, items, T, dev, "ucfcrime", batch_chunks=batch_chunks, lanes=lanes, wire_bf16=True)
            torch.cuda.synchronize()
            wdts.append(time.perf_counter() - t0)
        wdt = sorted(wdts)[len(wdts) // 2]
        out["bf16_wire"] = {"what": "same list, fp32 host rows rounded to bf16 by the staging threads (wire_dtype = BF16): 3,072 B per snippet "
                                    "over PCIe instead of 6,144",
                            "snippets_per_s": total / wdt, "seconds": wdt, "seconds_all_passes": wdts,
                            "max_abs_score_diff_vs_fp32_wire": float(max(np.abs(x - y).max() for x, y in zip(wire_scores, scores)))}
    for (dataset, lengths, classes, gt, seed, normal_keys), (lo_i, hi_i) in zip(parts, bounds):
        res = harness.evaluate_scores(scores[lo_i:hi_i], classes, gt, dataset, verbose=False, normal_keys=normal_keys)
        out["lists"][dataset] = {"videos": hi_i - lo_i, "snippets": int(np.sum(lengths)), "auc": res["roc"], "ap": res["ap"],
                                 "ano_auc": res["ano_auc"], "gt_positive_fraction": float(np.mean(gt))}
    del model
    # fp32 CPU oracle on an evenly spread sample of the videos, bounded in time
    oracle = orc.OracleMMFMIL(sd, orc.OracleConfig(num_layers=L, num_refinement_steps=K, nu=8))
    offs = np.concatenate([[0], np.cumsum([len(s_) for s_ in scores])])
    gt_all = np.concatenate([p[3] for p in parts])
    order = list(range(0, nvid, 7)) + [i for i in range(nvid) if i % 7]
    t0 = time.perf_counter()
    picked, cpu_scores = [], []
    for lo_i in range(0, len(order), 8):
        idx = order[lo_i:lo_i + 8]
        sc, _, _, _ = harness.score_loader(oracle, [items[i] for i in idx], T, "cpu", "ucfcrime", batch_chunks=8)
        picked += idx
        cpu_scores += sc
        if time.perf_counter() - t0 > a.cpu_seconds and len(picked) >= 32:
            break
    t_cpu = time.perf_counter() - t0
    nsub = int(sum(len(x) for x in cpu_scores))
    g_cat = np.concatenate([scores[i] for i in picked])
    c_cat = np.concatenate(cpu_scores)
    gt_sub = np.concatenate([gt_all[16 * offs[i]:16 * offs[i + 1]] for i in picked])
    out["vs_fp32_cpu_oracle_on_sample"] = {
        "videos": len(picked), "snippets": nsub, "oracle_snippets_per_s": nsub / t_cpu, "cores": torch.get_num_threads(),
        "max_abs_score_diff": float(np.abs(g_cat - c_cat).max()),
        "abs_auc_diff": abs(roc_auc_score(gt_sub, np.repeat(g_cat, 16)) - roc_auc_score(gt_sub, np.repeat(c_cat, 16))),
        "abs_ap_diff": abs(average_precision_score(gt_sub, np.repeat(g_cat, 16)) - average_precision_score(gt_sub, np.repeat(c_cat, 16)))}
    out["x_cpu_oracle"] = out["snippets_per_s"] / (nsub / t_cpu)
    if wire_scores is not None:
        w_cat = np.concatenate([wire_scores[i] for i in picked])
        out["bf16_wire"]["vs_fp32_cpu_oracle_on_sample"] = {
            "max_abs_score_diff": float(np.abs(w_cat - c_cat).max()),
            "abs_auc_diff": abs(roc_auc_score(gt_sub, np.repeat(w_cat, 16)) - roc_auc_score(gt_sub, np.repeat(c_cat, 16))),
            "abs_ap_diff": abs(average_precision_score(gt_sub, np.repeat(w_cat, 16)) - average_precision_score(gt_sub, np.repeat(c_cat, 16)))}
    return out


def train_step_block(dev, a, chunks=128, steps=4):
    """Both arithmetic modes of the training path: the headline's (bf16x6: forward projections and the backward's input-gradient
    products on the exact-split NT kernel, weight gradients on the split TN kernel, csrc/gemm_split_tn.h) first, fp32 MFMA throughout beside it."""
    out = train_step_one(dev, a, "bf16x6", chunks, steps)
    out["f32_mode"] = train_step_one(dev, a, "f32", chunks, steps)
    return out


def train_step_one(dev, a, compute, chunks=128, steps=4):
    """whole training steps of the reference's UCF configuration (train/ucf_train.py:43-106 with main.py's defaults:
    2 x batch_size 64 = 128 chunks of [256, 768], K = 10, StudentT nu = 8, attention dropout 0.1, AdamW lr 2e-5) on this GPU:
    train-mode forward, CLAS2 + regulariser + KL, backward of the loss head and of the model, AdamW -- all in libiefvad.
    Algorithmic FLOPs per step = 3 x the forward's 50.33 MFLOP per snippet."""
    import torch
    import iefvad_amd
    from iefvad_amd import losses, synth, trainer
    margs = argparse.Namespace(visual_layers=L, visual_head=H, num_refinement_steps=K_STEPS, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, D, T, D, H, L, 8, 10, 10, "cuda", margs, compute=compute)
    model.load_state_dict(synth.make_state_dict(0, D, L, K_STEPS))
    model = model.to(dev).train()
    opt = losses.AdamW(model.parameters(), lr=2e-5)
    gen = torch.Generator(device=dev).manual_seed(77)
    img = torch.randn(chunks, T, D, device=dev, generator=gen) * 0.45
    ev = torch.randn(chunks, T, D, device=dev, generator=gen) * 0.45
    labels = torch.zeros(chunks, 14, device=dev)
    labels[: chunks // 2, 0] = 1
    labels[chunks // 2:, 3] = 1
    lengths = torch.full((chunks,), T, dtype=torch.int64, device=dev)
    for _ in range(2):
        trainer.train_step(model, opt, img, ev, labels, lengths, "StudentT", 1.0, 1.0, want_terms=False)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s_ in range(steps):      # the loss terms are formed only on the step that is logged (ucf_train.py:108-128 prints every print_steps samples)
        terms = trainer.train_step(model, opt, img, ev, labels, lengths, "StudentT", 1.0, 1.0, want_terms=(s_ == steps - 1))
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    flops = 3 * TOTAL_FLOPS_PER_SNIPPET * chunks * T
    # the pipe the step's projection products issue on: bf16 MFMA (six products per algorithmic multiply-add) in bf16x6, fp32 MFMA in f32
    split = compute == "bf16x6"
    peak = PEAK_BF16_MFMA_TFLOPS if split else PEAK_F32_MFMA_TFLOPS
    out = {"workload": f"training step, {chunks} chunks x {T} x {D} (ucf_train.py: 2 x batch_size 64), K={K_STEPS}, L={L}, StudentT, attention "
                       f"dropout 0.1 (library mask generator), AdamW lr 2e-5, compute={compute}",
           "snippets_per_s": chunks * T / dt, "ms_per_step": dt * 1e3, "steps": steps, "algorithmic_tflop_per_step": flops / 1e12,
           "achieved_tflops": flops / dt / 1e12, "peak_tflops": peak, "frac": flops / dt / 1e12 / peak,
           "mfma_pipe_util": flops / dt / 1e12 * (PRODUCTS_PER_MAC[compute]) / peak,
           "peak_note": ("whole-step algorithmic FLOPs (3 x the forward's) over wall time against the dense bf16 MFMA peak; mfma_pipe_util counts the six "
                         "executed bf16 products per multiply-add of the projection products (of the attention products, S, Pd v, d Pd and d q run on "
                         "the same split arithmetic inside two fused launches per layer with P and its dropout mask as ONE stored tensor, d v and d k on the fp32 MFMA instruction, and the row "
                         "kernels on none, so this is an upper bound on the pipe's real occupancy)") if split else
                        "whole-step algorithmic FLOPs over wall time against the fp32 MFMA peak: every product on v_mfma_f32_32x32x2_f32",
           "loss_total": float(terms["total"]),
           "activation_buffer_gib": iefvad_amd.lib.load_library().iefvad_train_workspace_bytes(model._handle, chunks) / 2**30}
    return out


def metric_tail_block(dev, n=8192 * T, sample=131072):
    """the metric tail of test() (/root/reference/test.py:158-159: sklearn on np.repeat(scores, 16)) as the library entry
    iefvad_auc_ap on config 4's 2,097,152 snippet scores (33.5 M frames), device time incl. the result read-back; beside it sklearn on
    the host for a bounded sample of the same vectors (its cost grows n log n), and the agreement of the two on that sample."""
    import numpy as np
    import torch
    from iefvad_amd import harness, synth
    from sklearn.metrics import average_precision_score, roc_auc_score
    gen = torch.Generator(device=dev).manual_seed(5)
    scores = torch.sigmoid(torch.randn(n, device=dev, generator=gen) * 2)
    gt = torch.from_numpy(synth.make_gt(5, n)).to(torch.uint8).to(dev)
    harness.device_auc_ap(scores, gt)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        auc, ap = harness.device_auc_ap(scores, gt)
    dt = (time.perf_counter() - t0) / reps
    s_h, g_h = scores[:sample].cpu().numpy(), gt[: 16 * sample].cpu().numpy()
    t0 = time.perf_counter()
    a0, p0 = roc_auc_score(g_h, np.repeat(s_h, 16)), average_precision_score(g_h, np.repeat(s_h, 16))
    t_sk = time.perf_counter() - t0
    a1, p1 = harness.device_auc_ap(scores[:sample], gt[: 16 * sample])
    return {"snippets": n, "frames": 16 * n, "device_ms": dt * 1e3, "auc": auc, "ap": ap,
            "sklearn_sample": {"snippets": sample, "host_ms": t_sk * 1e3, "abs_auc_diff": abs(a1 - a0), "abs_ap_diff": abs(p1 - p0)},
            "kernel": "iefvad_auc_ap: radix sort of (score, positives) pairs + scan + tie-group reduction (csrc/metrics.h)"}


def xd_parts():
    from iefvad_amd import harness, synth
    lengths = synth.lognormal_lengths(2, 753, 145000)
    keys = harness.CLASS_KEYS["xd"]
    classes = [keys[i % len(keys)] for i in range(753)]
    return [("xd", lengths, classes, synth.make_gt(2, int(lengths.sum())), 2, ("normal",))]


def config5_parts():
    """The ShanghaiTech and MSAD test lists: real frame-level gt and label order (tests/golden/config5_gt.npz), synthetic
    features whose lengths sum to each gt exactly (8,723 + 9,009 = 17,732 snippets)."""
    from iefvad_amd import synth
    lists = synth.config5_lists(os.path.join(ROOT, "tests", "golden"))
    return [(d, lists[d][0], lists[d][1], lists[d][2], seed, nk) for seed, d, nk in ((51, "shang", ("normal",)), (52, "msad", ("Normal",)))]


def main():
    a = parse()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(a))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    # stdout carries exactly ONE line, the JSON record: native libraries (gloo, RCCL's NCCL WARN) and anything else that
    # writes to file descriptor 1 are sent to stderr from here on
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    import torch
    import torch.distributed as dist
    gpu = not a.plumbing_only
    backend = a.dist_backend if gpu else "gloo"
    if gpu:
        ndev = torch.cuda.device_count()
        dev_index = local_rank if backend == "nccl" else local_rank % max(ndev, 1)   # gloo rehearsal: ranks share GPUs
        torch.cuda.set_device(dev_index)
        dev = torch.device("cuda", dev_index)
    else:
        dev = torch.device("cpu")
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    def sync():
        if gpu:
            torch.cuda.synchronize()

    from iefvad_amd import harness, synth
    margs = argparse.Namespace(visual_layers=L, visual_head=H, num_refinement_steps=K_STEPS, lambda_ref=0.5,
                               noise_model="StudentT", nu=8)
    sd = synth.make_state_dict(0, D, L, K_STEPS) if gpu else None
    model = make_model(sd, margs, dev, a, a.compute) if gpu else None

    def shard(total_chunks, scaling):
        """(first chunk, chunk count) of this rank and the per-rank counts."""
        if scaling == "weak":
            return rank * total_chunks, total_chunks, [total_chunks] * world
        cuts = [r * total_chunks // world for r in range(world + 1)]        # [r*B/R, (r+1)*B/R)
        return cuts[rank], cuts[rank + 1] - cuts[rank], [cuts[r + 1] - cuts[r] for r in range(world)]

    def make_blocks(nchunks):
        if not gpu:
            return None, None
        gen = torch.Generator(device=dev)
        gen.manual_seed(1234 + rank)
        return (torch.randn(nchunks, T, D, device=dev, generator=gen) * 0.45,
                torch.randn(nchunks, T, D, device=dev, generator=gen) * 0.45)

    # how the ranks' score vectors are gathered is decided once, by all ranks together (harness.ScoreGatherer): the library's
    # RCCL gather, or -- when a rank cannot create the communicator -- torch.distributed's all-gather; the line says which
    gatherer = None
    if world > 1 and backend == "nccl":
        gatherer = harness.ScoreGatherer(dev)
    elif world > 1 and not gpu and a.plumbing_comm != "none":
        def fake_create(ident, nranks, r):
            if r == nranks - 1:
                if a.plumbing_comm == "hang":
                    time.sleep(3600)
                raise RuntimeError("injected failure on the last rank")
            return 0xC0FFEE
        hooks = {"version": lambda: 22205, "unique_id": lambda: b"\0" * 128, "create": fake_create, "destroy": lambda h: None,
                 "nranks": lambda h: world}
        gatherer = harness.ScoreGatherer("cpu", create_timeout=a.plumbing_comm_timeout, _hooks=hooks)
    elif world > 1:
        gatherer = harness.ScoreGatherer("cpu", prefer_library=False)
        gatherer.label = "torch.distributed gloo (scores staged through the host)"
    state = {"gather": gatherer.label if gatherer else None}

    def timed_run(img, ev, first, counts, steps, warmup):
        """Contract timing: `warmup` untimed steps, barrier + sync, exactly `steps` steps, sync + barrier, MAX over
        ranks.  Returns (max seconds, per-rank seconds, stage sums, last gathered scores)."""
        stage = {}
        snips = [c * T for c in counts]

        def step():
            if gpu:
                with torch.no_grad():
                    out = model(img, ev, None, None, None, timed=True)
                for k, v in model.last_stage_times.items():
                    stage[k] = stage.get(k, 0.0) + v
                scores = out["logits"].reshape(-1)
            else:       # --plumbing-only: the global snippet indices of this rank's shard
                scores = torch.arange(first * T, (first + counts[rank]) * T, dtype=torch.float32)
            if world > 1:
                scores = gatherer(scores if backend == "nccl" or not gpu else scores.cpu(), snips)
            return scores

        for _ in range(warmup):
            step()
        stage.clear()
        if world > 1:
            dist.barrier()
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            scores = step()
        sync()
        mine = time.perf_counter() - t0
        if world > 1:
            dist.barrier()
        dt = time.perf_counter() - t0
        per_rank = [mine]
        if world > 1:
            tdev = dev if backend == "nccl" else "cpu"
            tt = torch.tensor([dt], device=tdev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
            allr = torch.empty(world, device=tdev, dtype=torch.float64)
            dist.all_gather_into_tensor(allr, torch.tensor([mine], device=tdev, dtype=torch.float64))
            per_rank = allr.tolist()
        # (IEFVAD_TIMING_PROBE: tools/*_probe.sh run deliberately wrong-result probe builds for their timings only)
        assert scores.numel() == sum(snips) and (bool(torch.isfinite(scores).all()) or bool(os.environ.get("IEFVAD_TIMING_PROBE")))
        return dt, per_rank, stage, scores

    first, B, counts = shard(a.chunks, a.scaling)
    img, ev = make_blocks(B)
    dt, per_rank, stage, scores = timed_run(img, ev, first, counts, a.steps, a.warmup)
    total_chunks = sum(counts)
    gathered = int(scores.numel())
    in_order = bool((scores == torch.arange(total_chunks * T, dtype=torch.float32)).all()) if not gpu else None

    weak = None
    if gpu and world > 1 and a.scaling == "strong" and not a.no_extra_modes:
        del img, ev, scores
        wimg, wev = make_blocks(a.chunks)
        wsteps = max(2, min(a.steps, 3))
        wdt, wper, _, _ = timed_run(wimg, wev, rank * a.chunks, [a.chunks] * world, wsteps, 1)
        weak = {"value": world * a.chunks * T * wsteps / wdt, "unit": "snippets/s", "steps": wsteps,
                "ms_per_step": wdt / wsteps * 1e3, "chunks_per_gpu": a.chunks,
                "per_rank_ms_per_step": [t / wsteps * 1e3 for t in wper]}
        del wimg, wev
        img, ev = make_blocks(B)

    if rank == 0:
        value = total_chunks * T * a.steps / dt
        line = {
            "metric": "snippets/sec at [B,T=256,d=768]", "value": value if gpu else None, "unit": "snippets/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": a.scaling, "vs_baseline": None, "dtype": DTYPE[a.compute],
            "data": "synthetic",
            "config": {"workload": f"BASELINE config 4: synthetic [B={total_chunks},T=256,d=768] fp32 image+event blocks "
                                   f"resident in HBM ({'B in total, rank r holds chunks [r*B/N,(r+1)*B/N)' if a.scaling == 'strong' else 'B per GPU'}), "
                                   f"K=10 nu=8 StudentT, seeded random weights, outputs={a.outputs}, projections={a.compute}",
                       "chunks_total": total_chunks, "chunks_per_gpu": counts, "snippets_per_step": total_chunks * T,
                       "parallelism": (f"video-sharded x{world}, one score all-gather per step" if world > 1 else "single GPU")},
            "per_rank_ms_per_step": [t / a.steps * 1e3 for t in per_rank],
        }
        if gpu:
            roof = roofline_block(a.compute, stage, a.steps, B * T)
            mb_eff = a.micro_batch if a.micro_batch > 0 else (256 if a.compute == "f32" else 1024)     # library defaults
            roof.update(traffic_from_profiles(a.compute, min(B, mb_eff) * T))
            line["roofline"] = roof
            line["stage_ms_per_step"] = {k: v / a.steps for k, v in stage.items() if k.endswith("_ms")}
            line["end_to_end_tflops_per_gpu"] = TOTAL_FLOPS_PER_SNIPPET * value / world / 1e12
        else:
            line["plumbing_only"] = "no forward ran: fabricated scores, launcher / shard / gather / timing self-test"
            line["gathered_in_order"] = in_order
        if world > 1:
            comm = gatherer.comm if gatherer is not None else None
            line["rccl_ranks"] = comm.nranks if comm is not None else (dist.get_world_size() if backend == "nccl" else 0)
            line["dist_backend"] = backend
            line["gather"] = state["gather"]
            line["gathered_scores"] = gathered
            if weak is not None:
                line["weak_scaling"] = weak
        if gpu and world == 1 and not a.no_extra_modes:     # N=1 only: rank 0 must not linger at N>1
            for key, mode in (("f32_mfma_mode", "f32"), ("bf16_mode", "bf16"), ("fp16x3_mode", "fp16x3"), ("bf16x6_mode", "bf16x6")):
                if mode != a.compute:
                    line[key] = extra_mode(sd, margs, dev, img, ev, a, mode)
        img = ev = None
        if gpu and world == 1 and not a.no_ucf_eval:
            model = None
            torch.cuda.empty_cache()
            line["ucf_eval"] = ucf_eval(sd, margs, dev, a)
            # BASELINE configs 3 and 5 (bf16 projections, fp32 state), bounded: driver-side records of what tests/test_gpu_bf16.py gates
            line["xd_eval"] = dataset_eval("BASELINE config 3 (synthetic XD-Violence-sized set)", xd_parts(), 17, 10, "bf16", dev, a,
                                           batch_chunks=128, lanes=2)
            line["shang_msad_eval"] = dataset_eval("BASELINE config 5 (real gt and label order, synthetic features)", config5_parts(),
                                                   19, 5, "bf16", dev, a, batch_chunks=128, lanes=2)
            line["train_step"] = train_step_block(dev, a)
            line["metric_tail"] = metric_tail_block(dev)
        if gpu and world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, a.cpu_seconds)
        print(json.dumps(line), file=json_out, flush=True)
    if gatherer is not None:
        gatherer.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
