"""Parity of the HIP path (through the C ABI of libiefvad.so) against the golden vectors captured from
the reference and against the CPU oracle.  Needs a real MI355X: run with `-m gpu`."""
import argparse
import ctypes as C

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import synth
from oracle import iefvad_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def make_model(L, K, lam, noise, nu, sd, **kw):
    args = argparse.Namespace(visual_layers=L, visual_head=8, num_refinement_steps=K, lambda_ref=lam,
                              noise_model=noise, nu=nu)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, L, 8, 10, 10, "cuda", args, **kw)
    m.load_state_dict(sd)
    return m.to("cuda:0").eval()


def run(model, img, ev):
    with torch.no_grad():
        out = model(torch.from_numpy(img).cuda(), torch.from_numpy(ev).cuda(), None, None, None)
    torch.cuda.synchronize()
    return {k: v.cpu().numpy() for k, v in out.items()}


def test_native_library_is_loaded():
    lib = iefvad_amd.lib.load_library()
    assert lib.iefvad_abi_version() == iefvad_amd.lib.ABI_VERSION
    with open("/proc/self/maps") as f:
        assert "libiefvad.so" in f.read()


def test_gemm_kernel_matches_fp64():
    """C = A W^T + b on the fp32 MFMA kernel vs an fp64 host product; asymmetric operands catch a
    transposed accumulator map.  Tolerance: fp32 k-ordered fma chain, K=768 -> ~1e-7 * sum|a*b|."""
    lib = iefvad_amd.lib.load_library()
    rng = np.random.default_rng(0)
    for (M, N, K) in [(256, 768, 768), (512, 2304, 768), (128, 128, 32), (8192, 768, 768)]:   # small-tile and 128x128 kernels
        A = rng.standard_normal((M, K)).astype(np.float32)
        W = rng.standard_normal((N, K)).astype(np.float32)
        b = rng.standard_normal(N).astype(np.float32)
        dA, dW, db = (torch.from_numpy(x).cuda() for x in (A, W, b))
        dC = torch.empty(M, N, device="cuda")
        rc = lib.iefvad_gemm_bias(dA.data_ptr(), dW.data_ptr(), db.data_ptr(), dC.data_ptr(), M, N, K, 0,
                                  C.c_void_p(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, iefvad_amd.lib.last_error()
        torch.cuda.synchronize()
        ref = A.astype(np.float64) @ W.astype(np.float64).T + b
        err = np.abs(dC.cpu().numpy() - ref).max()
        assert err < 2e-4 * np.sqrt(K / 768.0) + 1e-5, (M, N, K, err)


def test_four_gemm_tilings_are_bit_identical():
    """The library picks the 128x256 ring kernel when its grid fills the chip, the 128x128 kernel for mid-size
    grids, a 64x64-tile kernel for small M and a 32x32-tile kernel on 16x16x4 MFMAs for the per-video pattern
    (M = 256 .. ~1500); all sum k in the same order, so the same rows give the same bits whatever M they arrive in
    (this is what makes micro-batching and cross-video packing exact)."""
    lib = iefvad_amd.lib.load_library()
    rng = np.random.default_rng(1)
    M, N, K = 16384, 768, 768
    A = torch.from_numpy(rng.standard_normal((M, K)).astype(np.float32)).cuda()
    W = torch.from_numpy(rng.standard_normal((N, K)).astype(np.float32)).cuda()
    b = torch.from_numpy(rng.standard_normal(N).astype(np.float32)).cuda()
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def gemm(rows):
        out = torch.empty(rows.shape[0], N, device="cuda")
        assert lib.iefvad_gemm_bias(rows.data_ptr(), W.data_ptr(), b.data_ptr(), out.data_ptr(), rows.shape[0], N, K, 0, st) == 0
        return out

    t256 = gemm(A)                             # 128 x 3 = 384 blocks of 128x256  -> ring kernel
    t128 = gemm(A[:8192].contiguous())         # 64 x 6 = 384 blocks of 128x128   -> double-buffer kernel
    t64 = gemm(A[2048:4096].contiguous())      # 32 x 12 = 384 blocks of 64x64    -> small-tile kernel
    t32 = gemm(A[1024:1280].contiguous())      # 4 x 12 = 48 blocks of 64x64 < 320 -> 32x32 tiles, 16x16x4 MFMA
    t32b = gemm(A[4096:5376].contiguous())     # M = 1280: 240 blocks of 64x64     -> 32x32 tiles
    torch.cuda.synchronize()
    assert torch.equal(t256[:8192], t128)
    assert torch.equal(t256[2048:4096], t64)
    assert torch.equal(t256[1024:1280], t32)
    assert torch.equal(t256[4096:5376], t32b)
    ref = A[:256].double().cpu() @ W.double().cpu().t() + b.double().cpu()
    assert (t256[:256].double().cpu() - ref).abs().max().item() < 2e-4


def test_per_video_sized_forwards_equal_rows_of_a_large_batch():
    """B = 1, 2, 3, 5 forwards (32x32-tile GEMMs, every epilogue: qkv scale, residual, heads split, relu, refine)
    against the same chunks inside a B = 40 batch (64x64 / 128x128 / ring kernels): all eight outputs bit-identical."""
    sd = synth.make_state_dict(91, 768, 2, 3)
    img, ev = synth.make_inputs(92, 40)
    model = make_model(2, 3, 0.5, "StudentT", 8, sd)
    big = run(model, img, ev)
    for lo, n in ((0, 1), (7, 2), (20, 3), (33, 5)):
        part = run(model, img[lo:lo + n], ev[lo:lo + n])
        for k in iefvad_amd.OUTPUT_KEYS:
            assert np.array_equal(part[k], big[k][lo:lo + n]), (k, lo, n)


@pytest.mark.parametrize("name", H.golden_cases())
def test_forward_matches_reference_golden(name):
    g, cfg, sd, img, ev = H.load_case(name)
    model = make_model(cfg["L"], cfg["K"], cfg["lam"], cfg["noise"], cfg["nu"], sd)
    out = run(model, img, ev)
    assert list(out.keys()) == list(iefvad_amd.OUTPUT_KEYS)
    assert out["logits"].shape == (cfg["B"], 256, 1) and out["fused"].shape == (cfg["B"], 256, 768)
    errs = H.compare_outputs(out, g)
    print(name, errs)


def test_scores_only_mode_matches_full_and_golden():
    g, cfg, sd, img, ev = H.load_case("base_k10_student8")
    full = run(make_model(cfg["L"], cfg["K"], cfg["lam"], cfg["noise"], cfg["nu"], sd), img, ev)
    lite = run(make_model(cfg["L"], cfg["K"], cfg["lam"], cfg["noise"], cfg["nu"], sd, outputs="scores"), img, ev)
    assert set(lite.keys()) == {"logits", "w_i_mean", "w_e_mean"}
    assert np.array_equal(lite["logits"], full["logits"])          # same kernels, same order -> bit equal
    assert np.abs(lite["w_i_mean"] - g["w_i_mean"]).max() < 2e-6
    assert np.abs(lite["w_e_mean"] - g["w_e_mean"]).max() < 2e-6


def test_micro_batching_is_bit_identical():
    """Chunks are independent batch rows: B=5 in one pass == the same chunks in passes of 2."""
    sd = synth.make_state_dict(31, 768, 2, 3)
    img, ev = synth.make_inputs(32, 5)
    a = run(make_model(2, 3, 0.5, "StudentT", 8, sd), img, ev)
    b = run(make_model(2, 3, 0.5, "StudentT", 8, sd, micro_batch=2), img, ev)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    c = run(make_model(2, 3, 0.5, "StudentT", 8, sd), img[3:4], ev[3:4])
    assert np.array_equal(a["logits"][3:4], c["logits"])


def test_forward_vs_oracle_larger_batch():
    """B=16 (4096 snippets) against the CPU oracle on the same seeded inputs."""
    sd = synth.make_state_dict(41, 768, 2, 10)
    img, ev = synth.make_inputs(42, 16)
    out = run(make_model(2, 10, 0.5, "StudentT", 8, sd), img, ev)
    ref = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), orc.OracleConfig())
    for k in iefvad_amd.OUTPUT_KEYS:
        d = np.abs(out[k] - ref[k].numpy()).max()
        assert d <= (H.TOL_LOGIT if k == "logits" else H.TOL_BIG), (k, d)
    ds = np.abs(H.sigmoid(out["logits"]) - H.sigmoid(ref["logits"].numpy())).max()
    assert ds <= H.TOL_SIGMOID


def test_bf16_and_fp64_inputs_follow_to_float():
    sd = synth.make_state_dict(51, 768, 1, 1)
    img, ev = synth.make_inputs(52, 1)
    model = make_model(1, 1, 0.5, "StudentT", 8, sd)
    cfg = orc.OracleConfig(num_layers=1, num_refinement_steps=1)
    for dt in (torch.bfloat16, torch.float64):
        ti, te = torch.from_numpy(img).to(dt), torch.from_numpy(ev).to(dt)
        with torch.no_grad():
            out = model(ti.cuda(), te.cuda(), None, None, None)
        ref = orc.forward(sd, ti, te, cfg)
        assert (out["logits"].cpu() - ref["logits"]).abs().max().item() < H.TOL_LOGIT


def test_error_paths():
    sd = synth.make_state_dict(61, 768, 1, 0)
    model = make_model(1, 0, 0.5, "StudentT", 8, sd)
    x = torch.zeros(1, 256, 768)
    with pytest.raises(RuntimeError):           # CPU tensors: no fallback
        model(x, x, None, None, None)
    with pytest.raises(ValueError):             # wrong T
        model(torch.zeros(1, 128, 768).cuda(), torch.zeros(1, 128, 768).cuda(), None, None, None)
    model.temporal.noise_model = "Laplace"
    with pytest.raises(ValueError):             # imf_vad.py:137-138
        model(x.cuda(), x.cuda(), None, None, None)
    model.temporal.noise_model = "StudentT"
    model.train()                               # train mode is the differentiable path (tests/test_gpu_train.py)
    assert model(x.cuda(), x.cuda(), None, None, None)["logits"].requires_grad


def test_literal_overflow_semantics():
    """logvar < -88.7 on both modalities makes exp(-logvar) overflow: the reference's literal formula
    gives inf/inf = NaN (imf_vad.py:135-142).  Parity mode keeps it."""
    sd = synth.make_state_dict(71, 768, 1, 0)
    for m in ("image", "event"):
        sd[f"temporal.{m}_logvar.weight"].zero_()
        sd[f"temporal.{m}_logvar.bias"].fill_(-100.0)
    img, ev = synth.make_inputs(72, 1)
    out = run(make_model(1, 0, 0.5, "StudentT", 8, sd), img, ev)
    ref = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), orc.OracleConfig(num_layers=1, num_refinement_steps=0))
    assert np.isnan(ref["w_i"].numpy()).all() and np.isnan(out["w_i"]).all()
    assert np.isnan(out["logits"]).all()


@pytest.mark.parametrize("seed", range(6))
def test_randomized_configs_vs_oracle(seed):
    """Random (L, K, noise model, nu, lambda, B, input dtype, input scale) draws against the CPU oracle."""
    rng = np.random.default_rng(1000 + seed)
    L = int(rng.integers(1, 4))
    K = int(rng.integers(0, 5))
    noise = ["StudentT", "Gaussian"][int(rng.integers(0, 2))]
    nu = int(rng.integers(2, 12))
    lam = float(rng.choice([0.1, 0.25, 0.5, 0.9]))
    B = int(rng.integers(1, 4))
    scale = float(rng.choice([0.05, 0.45, 2.0]))
    dt = [torch.float32, torch.float16][int(rng.integers(0, 2))]
    sd = synth.make_state_dict(2000 + seed, 768, L, K)
    img, ev = synth.make_inputs(3000 + seed, B, scale=scale)
    ti, te = torch.from_numpy(img).to(dt), torch.from_numpy(ev).to(dt)
    model = make_model(L, K, lam, noise, nu, sd)
    with torch.no_grad():
        out = model(ti.cuda(), te.cuda(), None, None, None)
    ref = orc.forward(sd, ti, te, orc.OracleConfig(num_layers=L, num_refinement_steps=K, lambda_ref=lam,
                                                   noise_model=noise, nu=nu))
    for k in iefvad_amd.OUTPUT_KEYS:
        d = (out[k].cpu() - ref[k]).abs().max().item()
        assert d <= (H.TOL_LOGIT if k == "logits" else H.TOL_BIG), (k, d, dict(L=L, K=K, noise=noise, nu=nu, lam=lam, B=B))
    ds = (torch.sigmoid(out["logits"].cpu()) - torch.sigmoid(ref["logits"])).abs().max().item()
    assert ds <= H.TOL_SIGMOID


def test_non_finite_inputs_propagate_like_the_reference():
    """An inf in one snippet of one modality: attention mixes it into every row of that chunk, so the
    reference's outputs are NaN for the whole chunk of that modality and the fusion makes the chunk's scores
    NaN; other chunks are untouched.  The HIP path must produce the same NaN pattern and the same finite values."""
    sd = synth.make_state_dict(81, 768, 2, 2)
    img, ev = synth.make_inputs(82, 3)
    img[1, 17, 5] = np.inf
    model = make_model(2, 2, 0.5, "StudentT", 8, sd)
    out = run(model, img, ev)
    ref = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), orc.OracleConfig(num_refinement_steps=2))
    for k in iefvad_amd.OUTPUT_KEYS:
        r = ref[k].numpy()
        assert np.array_equal(np.isnan(out[k]), np.isnan(r)), k
        fin = ~np.isnan(r)
        assert np.abs(out[k][fin] - r[fin]).max() <= (H.TOL_LOGIT if k == "logits" else H.TOL_BIG), k
    assert np.isnan(out["logits"][1]).all() and np.isfinite(out["logits"][[0, 2]]).all()
    assert np.isfinite(out["event_mu"]).all()          # the event branch never saw the inf


def test_graph_replay_is_bit_identical_to_direct_launches():
    """Calls with B <= graph_chunks replay a cached hipGraph (inputs and outputs staged through library-owned buffers);
    graph_chunks = -1 launches the same kernels one by one.  Same bits, for every output, every input dtype, repeated
    calls with fresh tensors, interleaved batch sizes, and after a weight update."""
    sd = synth.make_state_dict(93, 768, 2, 3)
    img, ev = synth.make_inputs(94, 6)
    direct = make_model(2, 3, 0.5, "StudentT", 8, sd, graph_chunks=-1)
    graphed = make_model(2, 3, 0.5, "StudentT", 8, sd, graph_chunks=4)
    lite = make_model(2, 3, 0.5, "StudentT", 8, sd, graph_chunks=4, outputs="scores")
    lite_direct = make_model(2, 3, 0.5, "StudentT", 8, sd, graph_chunks=-1, outputs="scores")
    for rep in range(2):
        for lo, n in ((0, 1), (1, 3), (0, 1), (2, 4), (0, 6)):          # B = 6 > graph_chunks: direct path inside `graphed`
            a = run(direct, img[lo:lo + n], ev[lo:lo + n])
            b = run(graphed, img[lo:lo + n].copy(), ev[lo:lo + n].copy())
            c = run(lite, img[lo:lo + n], ev[lo:lo + n])
            for k in iefvad_amd.OUTPUT_KEYS:
                assert np.array_equal(a[k], b[k]), (k, lo, n, rep)
            assert np.array_equal(a["logits"], c["logits"])
            # the graphed scores-mode row means come back through the library's small staging buffer: compare VALUES (the
            # full-dict path has no means to compare bits with) and, below, bits against an ungraphed scores-mode model
            assert np.abs(a["w_i"].mean(-1) - c["w_i_mean"]).max() < 1e-6 and np.abs(a["w_e"].mean(-1) - c["w_e_mean"]).max() < 1e-6
            d = run(lite_direct, img[lo:lo + n], ev[lo:lo + n])
            for k in ("logits", "w_i_mean", "w_e_mean"):
                assert np.array_equal(c[k], d[k]), (k, lo, n, rep)
    h16 = run(graphed, img[:2].astype(np.float16), ev[:2].astype(np.float16))
    d16 = run(direct, img[:2].astype(np.float16), ev[:2].astype(np.float16))
    for k in iefvad_amd.OUTPUT_KEYS:
        assert np.array_equal(h16[k], d16[k]), k
    with torch.no_grad():
        for m in (direct, graphed):
            m.temporal.classifier.bias.add_(0.25)
    a, b = run(direct, img[:1], ev[:1]), run(graphed, img[:1], ev[:1])
    assert np.array_equal(a["logits"], b["logits"])
