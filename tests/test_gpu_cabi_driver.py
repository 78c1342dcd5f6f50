"""The C ABI used WITHOUT Python or torch on the calling side: tests/cabi/abi_driver.cpp includes only
include/iefvad.h and the HIP runtime, loads seeded weights + inputs from a flat fp32 blob, runs the forward and
writes the logits, which must equal the ctypes/torch path's bit for bit."""
import argparse
import os
import subprocess

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("B,compute", [(2, "f32"), (48, "bf16x6"), (48, "fp16x3"), (64, "bf16")])
def test_torch_free_c_driver_matches_python_path(tmp_path, B, compute):
    """B = 2 runs the fp32 kernels; B = 48 is large enough for the split kernels of the bf16x6 / fp16x3 arithmetic; B = 64 in the
    bf16 mode for its two fused kernels (heads + fusion, out_proj + LayerNorm)."""
    L, K = 2, 3
    exe = tmp_path / "abi_driver"
    libdir = os.path.dirname(iefvad_amd.lib.LIB_PATH)
    # plain g++: the driver is host-only C++ (HIP runtime API + the C header), no device code, no torch
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include",
                           "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cabi", "abi_driver.cpp"),
                           "-L", libdir, "-liefvad", "-L", "/opt/rocm/lib", "-lamdhip64",
                           f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib", "-o", str(exe)])
    sd = synth.make_state_dict(91, 768, L, K)
    img, ev = synth.make_inputs(92, B)
    blob = tmp_path / "blob.bin"
    with open(blob, "wb") as f:
        for key, _, _ in synth.state_dict_keys(L, K):
            f.write(sd[key].numpy().astype(np.float32).tobytes())
        f.write(np.zeros(3, np.float32).tobytes())      # keep the feature blocks 16-byte aligned
        f.write(img.tobytes())
        f.write(ev.tobytes())
    out = tmp_path / "logits.bin"
    r = subprocess.run([str(exe), str(blob), str(B), str(L), str(K), str(out), str(iefvad_amd.lib.COMPUTE_CODES[compute])],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "abi_driver OK" in r.stdout
    got = np.fromfile(out, dtype=np.float32).reshape(B, 256)
    # the metric tail the driver printed (iefvad_auc_ap on its logits) against sklearn on the same logits and the same frame rule
    from sklearn.metrics import average_precision_score, roc_auc_score
    j = np.arange(B * 256 * 16, dtype=np.uint64)
    gt = ((((j * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)) >> np.uint64(29)) == 0).astype(np.float64)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("metrics AUC")][0].split()
    y = np.repeat(got.reshape(-1), 16)
    assert abs(float(line[2]) - roc_auc_score(gt, y)) < 1e-12 and abs(float(line[4]) - average_precision_score(gt, y)) < 1e-12
    args = argparse.Namespace(visual_layers=L, visual_head=8, num_refinement_steps=K, lambda_ref=0.5,
                              noise_model="StudentT", nu=8)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, L, 8, 10, 10, "cuda", args, outputs="scores", compute=compute)
    m.load_state_dict(sd)
    m = m.to("cuda:0").eval()
    with torch.no_grad():
        ref = m(torch.from_numpy(img).cuda(), torch.from_numpy(ev).cuda(), None, None, None)["logits"].cpu().numpy()
    assert np.array_equal(got, ref.reshape(B, 256))
