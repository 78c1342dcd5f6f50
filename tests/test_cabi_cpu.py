"""C-ABI checks that need no GPU: the shared library loads, exports every symbol include/iefvad.h
declares, and the ctypes structures have the layout a C compiler gives the header."""
import ctypes as C
import os
import re
import subprocess

import pytest

import iefvad_amd
from iefvad_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "iefvad.h")


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(L.LIB_PATH):
        L.build_library()
    return L.load_library()


def test_exports_every_declared_symbol(lib):
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    declared = sorted(set(re.findall(r"\b(iefvad_[a-z0-9_]+)\s*\(", text)))
    assert declared == sorted(L.SYMBOLS)
    for s in declared:
        assert hasattr(lib, s), s
    assert lib.iefvad_abi_version() == L.ABI_VERSION


def test_struct_layout_matches_c_compiler(tmp_path):
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "iefvad.h"\nint main(){printf("%zu %zu %zu %zu %zu %zu %zu\\n",'
                   'sizeof(iefvad_config),sizeof(iefvad_weights),sizeof(iefvad_outputs),sizeof(iefvad_stage_times),'
                   'offsetof(iefvad_weights,ref_w1),offsetof(iefvad_weights,cls_w),offsetof(iefvad_config,lambda_ref));return 0;}\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    want = [C.sizeof(L.Config), C.sizeof(L.Weights), C.sizeof(L.Outputs), C.sizeof(L.StageTimes),
            L.Weights.ref_w1.offset, L.Weights.cls_w.offset, L.Config.lambda_ref.offset]
    assert got == want


def test_create_rejects_bad_config_without_touching_the_gpu(lib):
    h = C.c_void_p()
    base = dict(abi_version=L.ABI_VERSION, embed_dim=768, seq_len=256, num_heads=8, num_layers=2, num_steps=10,
                noise_model=1, compute=0, lambda_ref=0.5, nu=8.0, epsilon=1e-8, micro_batch=0)
    for bad, frag in [(dict(embed_dim=512), "D=768"), (dict(num_layers=0), "num_layers"), (dict(num_steps=65), "num_steps"),
                      (dict(noise_model=7), "Unsupported noise_model"), (dict(abi_version=9), "abi_version"),
                      (dict(compute=5), "compute mode")]:
        cfg = L.Config(**dict(base, **bad))
        assert lib.iefvad_create(C.byref(cfg), C.byref(h)) != 0
        assert frag in L.last_error(), (bad, L.last_error())
        assert not h.value
    assert lib.iefvad_workspace_bytes(None, 4) == 0


def test_gather_entry_points_reject_bad_arguments_without_a_gpu(lib):
    """iefvad_comm_* / iefvad_gather_scores validate before they touch RCCL or the device."""
    h = C.c_void_p()
    ident = C.create_string_buffer(L.COMM_ID_BYTES)
    assert lib.iefvad_comm_create(ident, 2, 2, C.byref(h)) != 0 and "rank 2 of 2" in L.last_error()
    assert lib.iefvad_comm_create(None, 1, 0, C.byref(h)) != 0 and "null" in L.last_error()
    assert lib.iefvad_comm_unique_id(None) != 0
    assert lib.iefvad_comm_nranks(None) == 0
    assert lib.iefvad_gather_scores(None, None, 0, None, None, None) != 0 and "null" in L.last_error()
    lib.iefvad_comm_destroy(None)


def test_shim_refuses_cpu_tensors():
    import argparse
    import torch
    args = argparse.Namespace(visual_layers=1, visual_head=8, num_refinement_steps=0, lambda_ref=0.5,
                              noise_model="StudentT", nu=8)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 1, 8, 10, 10, "cuda", args).eval()
    x = torch.zeros(1, 256, 768)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(x, x, None, None, None)
