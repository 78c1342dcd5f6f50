"""the metric tail of the evaluation loop as a library entry (`iefvad_auc_ap`, csrc/metrics.h) against the calls the
reference makes (/root/reference/test.py:158-159: sklearn's roc_auc_score / average_precision_score on np.repeat(scores, 16)).
Everything goes through the C ABI (ctypes); sklearn on the host is the checker.  `-m gpu`."""
import ctypes as C

import numpy as np
import pytest
import torch
from sklearn.metrics import average_precision_score, roc_auc_score

from iefvad_amd import harness, lib as L, synth

pytestmark = pytest.mark.gpu


def auc_ap(scores: np.ndarray, gt: np.ndarray, repeat: int = 16):
    lib = L.load_library()
    s = torch.from_numpy(np.ascontiguousarray(scores, dtype=np.float32)).cuda()
    g = torch.from_numpy((np.asarray(gt) != 0).astype(np.uint8)).cuda()
    n = s.numel()
    ws = torch.empty(lib.iefvad_auc_ap_workspace_bytes(n) + 256, dtype=torch.uint8, device="cuda")
    off = (-ws.data_ptr()) % 256
    out = torch.full((2,), -7.0, dtype=torch.float64, device="cuda")
    rc = lib.iefvad_auc_ap(C.c_void_p(s.data_ptr()), C.c_void_p(g.data_ptr()), n, repeat, C.c_void_p(out.data_ptr()),
                           C.c_void_p(out.data_ptr() + 8), C.c_void_p(ws.data_ptr() + off), ws.numel() - off,
                           C.c_void_p(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, L.last_error()
    return tuple(out.cpu().tolist())


def sk(scores, gt, repeat=16):
    y = np.repeat(np.asarray(scores, dtype=np.float32), repeat)
    return roc_auc_score(gt, y), average_precision_score(gt, y)


@pytest.mark.parametrize("n,quant", [(1, None), (2, None), (63, 4), (64, None), (500, None), (2000, 64), (4095, None), (4096, 16), (4097, None),
                                     (12289, 3), (69510, None), (145000, 1000)])
def test_auc_ap_equals_sklearn_including_ties_and_tile_edges(n, quant):
    """Sizes around the 4096-pair tile of the sort and the scans; `quant` rounds the scores to a few levels (heavy ties: thresholds are
    the DISTINCT values, as sklearn's _binary_clf_curve); the gt depends on the score so the curves are not trivial."""
    rng = np.random.default_rng(n)
    s = rng.random(n).astype(np.float32)
    if quant:
        s = (np.round(s * quant) / quant).astype(np.float32)
    gt = (rng.random(16 * n) < 0.15 + 0.5 * np.repeat(s, 16)).astype(np.float64)
    if gt.min() == gt.max():
        gt[0], gt[-1] = 0.0, 1.0
    auc, ap = auc_ap(s, gt)
    a0, p0 = sk(s, gt)
    assert abs(auc - a0) < 1e-12 and abs(ap - p0) < 1e-12, (auc - a0, ap - p0)


def test_auc_ap_on_signed_tiny_and_saturated_scores():
    """Keys must order like the floats: negative values, +-0 (one threshold), denormals, exact 0 / 1 from a saturated sigmoid (large tie
    groups), +-inf."""
    rng = np.random.default_rng(0)
    n = 30000
    s = rng.standard_normal(n).astype(np.float32)
    s[::5] = 1.0
    s[1::5] = 0.0
    s[2::50] = -0.0
    s[3::70] = 1e-42
    s[4::90] = -1e-42
    gt = (rng.random(16 * n) < 0.3).astype(np.float64)
    auc, ap = auc_ap(s, gt)
    a0, p0 = sk(s, gt)
    assert abs(auc - a0) < 1e-12 and abs(ap - p0) < 1e-12
    # +-inf (sklearn refuses them): they must rank above / below every finite score, i.e. like +-3e38 stand-ins
    s[7::1000] = np.inf
    s[9::1000] = -np.inf
    auc, ap = auc_ap(s, gt)
    a0, p0 = sk(np.where(np.isinf(s), np.sign(s) * np.float32(3e38), s), gt)
    assert abs(auc - a0) < 1e-12 and abs(ap - p0) < 1e-12


def test_auc_ap_all_scores_equal_and_degenerate_labels():
    n = 10000
    gt = (np.random.default_rng(1).random(16 * n) < 0.2).astype(np.float64)
    auc, ap = auc_ap(np.full(n, 0.5, np.float32), gt)             # one threshold: AUC 1/2, AP = prevalence
    assert auc == 0.5 and abs(ap - gt.mean()) < 1e-15
    s = np.random.default_rng(2).random(n).astype(np.float32)
    auc, ap = auc_ap(s, np.zeros(16 * n))                          # no positive frame: sklearn raises for AUC, returns 0 for AP
    assert np.isnan(auc) and ap == 0.0
    auc, ap = auc_ap(s, np.ones(16 * n))
    assert np.isnan(auc) and abs(ap - 1.0) < 1e-15
    s[77] = np.nan                                                 # sklearn refuses NaN scores; here both results are NaN
    auc, ap = auc_ap(s, gt)
    assert np.isnan(auc) and np.isnan(ap)


def test_auc_ap_other_repeat_factors_and_determinism():
    rng = np.random.default_rng(5)
    for repeat in (1, 3, 16, 20):
        n = 5000
        s = (np.round(rng.random(n) * 200) / 200).astype(np.float32)
        gt = (rng.random(repeat * n) < 0.25).astype(np.float64)
        a, p = auc_ap(s, gt, repeat)
        y = np.repeat(s, repeat)
        assert abs(a - roc_auc_score(gt, y)) < 1e-12 and abs(p - average_precision_score(gt, y)) < 1e-12
        assert (a, p) == auc_ap(s, gt, repeat)                     # integer AUC numerator, fixed-order AP sum: same bits every run


def test_auc_ap_at_config4_size_and_dataset_shapes():
    """2,097,152 snippets (BASELINE config 4: 33.5 M frames) and the snippet totals of configs 2 / 3 / 5, scores as sigmoid outputs
    in fp32 (natural ties), against sklearn on the materialised x16 repeat."""
    for n, seed in ((2097152, 4), (69500, 2), (145000, 3), (17732, 5)):
        rng = np.random.default_rng(seed)
        logit = rng.standard_normal(n).astype(np.float32) * 3
        s = (1.0 / (1.0 + np.exp(-logit))).astype(np.float32)
        gt = synth.make_gt(seed, n)
        auc, ap = auc_ap(s, gt)
        a0, p0 = sk(s, gt)
        assert abs(auc - a0) < 1e-12 and abs(ap - p0) < 1e-12, n


def test_auc_ap_argument_errors_and_harness_wrapper():
    lib = L.load_library()
    s = torch.rand(100, device="cuda")
    g = torch.zeros(1600, dtype=torch.uint8, device="cuda")
    out = torch.zeros(2, dtype=torch.float64, device="cuda")
    ws = torch.empty(lib.iefvad_auc_ap_workspace_bytes(100) + 256, dtype=torch.uint8, device="cuda")
    off = (-ws.data_ptr()) % 256
    p = lambda t, o=0: C.c_void_p(t.data_ptr() + o)
    assert lib.iefvad_auc_ap_workspace_bytes(0) == 0
    assert lib.iefvad_auc_ap(p(s), p(g), 100, 16, p(out), p(out, 8), p(ws, off), 64, None) != 0 and "workspace" in L.last_error()
    assert lib.iefvad_auc_ap(p(s), p(g), 100, 16, p(out), p(out, 8), p(ws, off + 8), ws.numel() - off - 8, None) != 0      # misaligned
    assert lib.iefvad_auc_ap(p(s), p(g), 0, 16, p(out), p(out, 8), p(ws, off), ws.numel() - off, None) != 0
    assert lib.iefvad_auc_ap(p(s), p(g), 1 << 29, 16, p(out), p(out, 8), p(ws, off), ws.numel() - off, None) != 0 and "32-bit" in L.last_error()
    assert lib.iefvad_auc_ap(p(s), None, 100, 16, p(out), p(out, 8), p(ws, off), ws.numel() - off, None) != 0
    assert lib.iefvad_auc_ap(p(s), p(g), 100, 16, None, None, p(ws, off), ws.numel() - off, None) != 0
    # one result only
    g[::3] = 1
    assert lib.iefvad_auc_ap(p(s), p(g), 100, 16, None, p(out, 8), p(ws, off), ws.numel() - off, None) == 0
    torch.cuda.synchronize()
    assert out[0].item() == 0.0 and 0.0 < out[1].item() < 1.0
    # the harness wrapper (float64 0/1 gt on the host, as np.load(args.gt_path) delivers it)
    rng = np.random.default_rng(9)
    n = 69510
    sc = rng.random(n).astype(np.float32)
    sc[::7] = sc[1::7][: len(sc[::7])]
    gt = (rng.random(16 * n) < 0.2).astype(np.float64)
    auc, ap = harness.device_auc_ap(torch.from_numpy(sc).cuda(), torch.from_numpy(gt))
    a0, p0 = sk(sc, gt)
    assert abs(auc - a0) < 1e-12 and abs(ap - p0) < 1e-12
