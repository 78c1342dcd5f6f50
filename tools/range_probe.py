#!/usr/bin/env python3
"""Range / non-finite probe of the arithmetic modes at a split-sized batch (B = 48 chunks), against the CPU oracle.
Prints one JSON object: (1) input scale x3000 -- per mode, max |error| vs the fp64 oracle per output and the number
of rows beyond the fp32 gate; (2) non-finite inputs (inf, NaN, |x| near FLT_MAX) -- per mode and case, whether the
NaN pattern of every output equals the fp32 oracle's, and the error on the finite part."""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import iefvad_amd  # noqa: E402
from iefvad_amd import harness, synth  # noqa: E402
from oracle import iefvad_oracle as orc  # noqa: E402

B = 48
KEYS = list(iefvad_amd.OUTPUT_KEYS)


def model(sd, compute, K=10):
    args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=K, lambda_ref=0.5, noise_model="StudentT", nu=8)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args, compute=compute)
    m.load_state_dict(sd)
    return m.to("cuda:0").eval()


def run(m, img, ev):
    with torch.no_grad():
        out = m(torch.from_numpy(img).cuda(), torch.from_numpy(ev).cuda(), None, None, None)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def main():
    torch.set_num_threads(harness.host_cpu_share())
    res = {}
    sd = synth.make_state_dict(5)
    cfg = orc.OracleConfig(num_layers=2, num_refinement_steps=10, nu=8)
    for scale in (100.0, 3000.0):
        img, ev = synth.make_inputs(13, B)
        img, ev = (img * scale).astype(np.float32), (ev * scale).astype(np.float32)
        r64 = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), cfg, dtype=torch.float64)
        r32 = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), cfg)
        block = {"oracle_f32": {k: float((r32[k].double() - r64[k]).abs().max()) for k in ("fused", "image_mu", "logits")}}
        bad = ((r32["fused"].double() - r64["fused"]).abs().amax(-1) > 4e-5)
        block["oracle_f32"]["rows_beyond_gate"] = int(bad.sum())
        for mode in ("f32", "bf16x6", "fp16x3"):
            got = run(model(sd, mode), img, ev)
            e = {k: float(np.abs(got[k] - r64[k].numpy()).max()) for k in ("fused", "image_mu", "logits")}
            rows = np.abs(got["fused"] - r64["fused"].numpy()).max(-1) > 4e-5
            e["rows_beyond_gate"] = int(rows.sum())
            e["chunks_with_bad_rows"] = int(rows.any(-1).sum())
            rows_mu = np.abs(got["image_mu"] - r64["image_mu"].numpy()).max(-1) > 4e-5
            e["image_mu_rows_beyond_gate"] = int(rows_mu.sum())
            e["finite"] = bool(all(np.isfinite(got[k]).all() for k in KEYS))
            block[mode] = e
        res[f"scale_{int(scale)}"] = block
    # non-finite inputs
    sd2 = synth.make_state_dict(81, 768, 2, 2)
    cfg2 = orc.OracleConfig(num_refinement_steps=2)
    cases = {"inf_img": (5, 17, 5, "img", np.inf), "nan_ev": (9, 200, 700, "ev", np.nan),
             "big_img": (20, 3, 11, "img", 3.3e38), "neginf_ev": (30, 255, 767, "ev", -np.inf),
             "big_ev_3e37": (40, 100, 100, "ev", 3e37)}
    for name, (c, r, d, which, val) in cases.items():
        img, ev = synth.make_inputs(82, B)
        (img if which == "img" else ev)[c, r, d] = val
        ref = orc.forward(sd2, torch.from_numpy(img), torch.from_numpy(ev), cfg2)
        block = {"oracle_nan_chunks": {k: [int(i) for i in np.nonzero(np.isnan(ref[k].numpy()).reshape(B, -1).any(1))[0]] for k in KEYS},
                 "oracle_inf_count": {k: int(np.isinf(ref[k].numpy()).sum()) for k in KEYS}}
        for mode in ("f32", "bf16x6", "fp16x3"):
            got = run(model(sd2, mode, K=2), img, ev)
            e = {}
            for k in KEYS:
                rr = ref[k].numpy()
                same = bool(np.array_equal(np.isnan(got[k]), np.isnan(rr)))
                fin = np.isfinite(rr) & np.isfinite(got[k])
                e[k] = {"nan_pattern_equal": same, "nan_got": int(np.isnan(got[k]).sum()), "nan_ref": int(np.isnan(rr).sum()),
                        "inf_got": int(np.isinf(got[k]).sum()),
                        "max_err_finite": float(np.abs(got[k][fin] - rr[fin]).max()) if fin.any() else None,
                        "nan_chunks_got": [int(i) for i in np.nonzero(np.isnan(got[k]).reshape(B, -1).any(1))[0]][:8]}
            block[mode] = e
        res[name] = block
    print(json.dumps(res))


if __name__ == "__main__":
    main()
