#!/usr/bin/env python3
"""bench.py -- snippets/sec of the IEF-VAD fusion-inference forward on MI355X.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

A "step" is one pass of the hot path (MMFMIL.forward through libiefvad.so, scores-only outputs) over
one batch of synthetic [B, T=256, d=768] fp32 image + event feature blocks that are already resident
in HBM, followed -- when N > 1 -- by the single RCCL all-gather of per-snippet scores.  Default batch:
BASELINE.json config 4's B = 8192 chunks (2,097,152 snippets) PER GPU; every rank holds its own
blocks (videos shard embarrassingly), so scaling is weak and `value` is the whole-job
aggregate.  Weights are seeded synthetic tensors of the reference architecture (K=10, nu=8, StudentT).

Arithmetic (--compute): the default, bf16x6, is fp32-ACCURATE: every fp32 operand of a dense projection is split
exactly into three bf16 terms and the product is accumulated in fp32 from six bf16 MFMA products (truncation <= 2^-23 of a product in
the worst case, ~2^-27 typically -- below the fp32 accumulation rounding that follows; csrc/gemm_split.h).  It is held to the same parity gates as the fp32
MFMA mode (tests/test_gpu_bf16x6.py) and its error against an fp64 evaluation is not larger.  MI355X multiplies bf16
16x faster than fp32, so this beats the fp32 MFMA instruction; the same workload on that instruction
(--compute f32) is timed in the same run and reported as `f32_mfma_mode`.

The JSON line also carries
  roofline      the dense-projection GEMM kernel (~85 % of device time): the MFMA FLOPs the kernel executes in a
                step (bf16x6: six bf16 multiply-adds per algorithmic fp32 multiply-add; the algorithmic rate is
                `algorithmic_fp32_tflops`) / the sum of that kernel's launch durations in the step, timed with
                hipEvents on the launch stream inside the timed region, against the dense MFMA peak of the
                operand type from MI355X_MICROARCH.md (bf16 2500, fp32 157.3 TFLOP/s);
  cpu_baseline  the CPU oracle (oracle/iefvad_oracle.py, torch CPU ops on all host cores) timed on a
                bounded sample of the same workload, rank 0 at N=1 only.  A reported baseline, not the target.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

T, D, H, L, K_STEPS = 256, 768, 8, 2, 10
GEMM_FLOPS_PER_SNIPPET = 4 * (2 * D * 3 * D) + 4 * (2 * D * D) + 4 * (2 * D * D) + K_STEPS * 2 * (2 * D * D)
TOTAL_FLOPS_PER_SNIPPET = 26_740_224 + K_STEPS * 2_359_296
PEAK_F32_MFMA_TFLOPS = 157.3                                    # MI355X_MICROARCH.md, chip-level table
PEAK_BF16_MFMA_TFLOPS = 2500.0                                  # dense bf16 MFMA (same table)


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=5)
    p.add_argument("--warmup", type=int, default=2)
    p.add_argument("--chunks", type=int, default=8192, help="chunks [256,768] per GPU per step")
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
                   help="skip the short fp32-MFMA pass that a default (bf16x6) run appends as `f32_mfma_mode`")
    p.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline sample")
    return p.parse_args()


def host_cpu_share():
    from iefvad_amd.harness import host_cpu_share as f
    return f()


def cpu_baseline(sd, seconds):
    """The oracle (CPU restatement of the reference forward) on the host cores: B=8 chunks per call, the
    best-case batched pattern of SURVEY.md 8d, repeated for ~`seconds`."""
    import numpy as np
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


def extra_mode(sd, margs, dev, img, ev, a, compute, steps=3):
    """The same workload on this rank's GPU in another compute mode (no gather), reported beside the headline: "f32" = the
    projections on the fp32 matrix-core instruction (v_mfma_f32_32x32x2_f32); "fp16x3" = the opt-in two-term fp16 split."""
    import iefvad_amd
    model = iefvad_amd.MMFMIL(14, D, T, D, H, L, 8, 10, 10, "cuda", margs, outputs=a.outputs,
                              micro_batch=a.micro_batch, compute=compute)
    model.load_state_dict(sd)
    model = model.to(dev).eval()
    stage = {}
    with torch.no_grad():
        model(img, ev, None, None, None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            model(img, ev, None, None, None, timed=True)
            for k, v in model.last_stage_times.items():
                stage[k] = stage.get(k, 0.0) + v
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    B = img.shape[0]
    gemm_ms = (stage["qkv_gemm_ms"] + stage["out_gemm_ms"] + stage["head_gemm_ms"] + stage["refine_gemm_ms"]) / steps
    alg = GEMM_FLOPS_PER_SNIPPET * B * T / (gemm_ms * 1e-3) / 1e12
    if compute == "fp16x3":
        roof = {"bound": "mfma", "kernel": "iefvad_gemm_split_f16_n128_kernel", "achieved": 3.0 * alg,
                "peak": PEAK_BF16_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": 3.0 * alg / PEAK_BF16_MFMA_TFLOPS,
                "algorithmic_fp32_tflops": alg,
                "note": "opt-in near-fp32 arithmetic: 2-term fp16 split of both scaled fp32 operands, three fp16 MFMA products "
                        "(products to ~2^-20.4 worst case, 2^-23 typical), fp32 accumulation; meets the f32 parity gates (tests/test_gpu_fp16x3.py), error vs "
                        "fp64 in profiles/r01_mode_accuracy.json"}
    else:
        roof = {"bound": "mfma", "kernel": "iefvad_gemm_f32_t256_kernel", "achieved": alg,
                "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s", "frac": alg / PEAK_F32_MFMA_TFLOPS}
    return {"compute": compute, "value": B * T * steps / dt, "unit": "snippets/s per GPU", "steps": steps,
            "ms_per_step": dt / steps * 1e3, "roofline": roof,
            "stage_ms_per_step": {k: v / steps for k, v in stage.items() if k.endswith("_ms")}}


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    import torch.distributed as dist
    ndev = torch.cuda.device_count()
    dev_index = local_rank if a.dist_backend == "nccl" else local_rank % max(ndev, 1)   # rehearsal: ranks share GPUs
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if world > 1:
        if a.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)     # nccl == RCCL on ROCm
        else:
            dist.init_process_group("gloo")

    import iefvad_amd
    from iefvad_amd import synth
    from iefvad_amd.harness import gather_scores
    margs = argparse.Namespace(visual_layers=L, visual_head=H, num_refinement_steps=K_STEPS, lambda_ref=0.5,
                               noise_model="StudentT", nu=8)
    sd = synth.make_state_dict(0, D, L, K_STEPS)
    model = iefvad_amd.MMFMIL(14, D, T, D, H, L, 8, 10, 10, "cuda", margs, outputs=a.outputs,
                              micro_batch=a.micro_batch, compute=a.compute)
    model.load_state_dict(sd)
    model = model.to(dev).eval()

    B = a.chunks
    gen = torch.Generator(device=dev)
    gen.manual_seed(1234 + rank)
    img = torch.randn(B, T, D, device=dev, generator=gen) * 0.45
    ev = torch.randn(B, T, D, device=dev, generator=gen) * 0.45

    stage = {}

    def step():
        with torch.no_grad():
            out = model(img, ev, None, None, None, timed=True)
        for k, v in model.last_stage_times.items():
            stage[k] = stage.get(k, 0.0) + v
        scores = out["logits"].reshape(-1)
        if world > 1:
            scores = gather_scores(scores if a.dist_backend == "nccl" else scores.cpu())
        return scores

    for _ in range(a.warmup):
        step()
    stage.clear()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        scores = step()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=dev if a.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    assert scores.numel() == world * B * T and bool(torch.isfinite(scores).all())

    if rank == 0:
        snippets = world * B * T * a.steps
        value = snippets / dt
        gemm_ms = (stage["qkv_gemm_ms"] + stage["out_gemm_ms"] + stage["head_gemm_ms"] + stage["refine_gemm_ms"]) / a.steps
        launches = stage["gemm_launches"] / a.steps
        gemm_flops = GEMM_FLOPS_PER_SNIPPET * B * T                     # per step, this rank
        achieved = gemm_flops / (gemm_ms * 1e-3) / 1e12
        traffic = None      # HBM-side bytes per GEMM launch from the committed PMC passes (same rows per launch)
        tpath = os.path.join(ROOT, "profiles", {"f32": "r01_gemm_hbm_traffic.json", "bf16x6": "r01_gemm_split_hbm_traffic.json"}
                             .get(a.compute, "none"))
        peak = PEAK_F32_MFMA_TFLOPS if a.compute == "f32" else PEAK_BF16_MFMA_TFLOPS
        # bf16x6: each algorithmic (fp32) multiply-add is executed as six bf16 MFMA multiply-adds
        executed = achieved * {"bf16x6": 6.0, "fp16x3": 3.0}.get(a.compute, 1.0)
        kernel = {"f32": "iefvad_gemm_f32_t256_kernel", "bf16": "iefvad_gemm_bf16_kernel",
                  "bf16x6": "iefvad_gemm_split_n128_kernel", "fp16x3": "iefvad_gemm_split_f16_n128_kernel"}[a.compute]
        dtype = {"f32": "f32", "bf16": "bf16",
                 "bf16x6": "f32 emulated: exact 3-term bf16 split of both fp32 operands, 6 bf16 MFMA products, fp32 accumulate",
                 "fp16x3": "near-f32 (22-bit products): 2-term fp16 split of both scaled fp32 operands, 3 fp16 MFMA products, "
                           "fp32 accumulate"}[a.compute]
        mb_eff = a.micro_batch if a.micro_batch > 0 else (256 if a.compute == "f32" else 1024)   # library defaults
        if os.path.exists(tpath):
            tj = json.load(open(tpath))
            if tj.get("rows_per_launch") == min(B, mb_eff) * T:      # the PMC passes ran at this launch size
                traffic = tj["traffic_bytes_per_launch"]
        line = {
            "metric": "snippets/sec at [B,T=256,d=768]", "value": value, "unit": "snippets/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt / a.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": dtype, "data": "synthetic",
            "config": {"workload": f"synthetic [B={B},T=256,d=768] fp32 image+event blocks per GPU resident in HBM "
                                   f"(BASELINE config 4 batch), K=10 nu=8 StudentT, seeded random weights, "
                                   f"outputs={a.outputs}, projections={a.compute}",
                       "chunks_per_gpu": B, "snippets_per_step": world * B * T,
                       "parallelism": f"video-sharded x{world}, score all-gather" if world > 1 else "single GPU"},
            "roofline": {"bound": "mfma", "achieved": executed, "peak": peak, "unit": "TFLOP/s",
                         "frac": executed / peak, "traffic": traffic,
                         "kernel": kernel,
                         "algorithmic_fp32_tflops": achieved,
                         "algorithmic_vs_fp32_mfma_peak": achieved / PEAK_F32_MFMA_TFLOPS,
                         "launches_per_step": launches,
                         "avg_launch_ms": gemm_ms / launches,
                         "flops_per_launch": gemm_flops / launches,
                         "executed_mfma_flops_per_launch": gemm_flops / launches * {"bf16x6": 6.0, "fp16x3": 3.0}.get(a.compute, 1.0),
                         "note": ("achieved / peak are the bf16 MFMA FLOPs the kernel executes (six bf16 multiply-adds per "
                                  "algorithmic fp32 multiply-add of SURVEY 8d) against the dense bf16 MFMA peak; the algorithmic "
                                  "rate is algorithmic_fp32_tflops = flops_per_launch / avg_launch_ms"
                                  if a.compute == "bf16x6" else
                                  "achieved = algorithmic GEMM FLOPs (SURVEY 8d) / kernel time")},
            "stage_ms_per_step": {k: v / a.steps for k, v in stage.items() if k.endswith("_ms")},
            "end_to_end_tflops": TOTAL_FLOPS_PER_SNIPPET * value / world / 1e12,
        }
        if world == 1 and a.compute == "bf16x6" and not a.no_extra_modes:   # N=1 only: rank 0 must not linger at N>1
            line["f32_mfma_mode"] = extra_mode(sd, margs, dev, img, ev, a, "f32")
            line["fp16x3_mode"] = extra_mode(sd, margs, dev, img, ev, a, "fp16x3")
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(sd, a.cpu_seconds)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
