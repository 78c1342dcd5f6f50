"""Host-side callers (chunker, dataset, evaluation loop, sharding) on the CPU.  The model behind the
loop is the CPU oracle -- the harness code under test is the product's."""
import argparse
import os

import numpy as np
import pytest
import torch

from iefvad_amd import harness, synth
from oracle import iefvad_oracle as orc
from tests import helpers as H


def test_process_split_matches_reference_rule():
    rng = np.random.default_rng(0)
    for n in (1, 37, 255, 256, 257, 512, 700):
        f = rng.standard_normal((n, 8)).astype(np.float16 if n == 37 else np.float32)
        a, la = harness.process_split(f, 256)
        b, lb = orc.process_split(f, 256)
        assert la == lb == n and a.dtype == f.dtype and a.shape == b.shape and np.array_equal(a, b)
    assert harness.process_split(np.ones((256, 4), np.float32), 256)[0].shape == (2, 256, 4)   # all-zero extra chunk
    assert harness.process_split(np.ones((256, 4), np.float32), 256)[0][1].max() == 0


@pytest.fixture(scope="module")
def config1(tmp_path_factory, golden_dir):
    """The synthetic config-1 .npy set written to disk exactly as make_golden.py wrote it."""
    return H.write_config1_set(tmp_path_factory.mktemp("cfg1"), golden_dir)


def test_loader_items_have_reference_shapes(config1):
    g, args, gt, sd = config1
    loader = harness.get_test_loader(args)
    shapes = {}
    for item, n in zip(loader, g["lengths"]):
        assert int(item[3]) == int(n)
        shapes[int(n)] = tuple(item[0].shape)
    assert shapes[100] == (1, 256, 768) and shapes[256] == (1, 2, 256, 768) and shapes[300] == (1, 2, 256, 768)
    assert shapes[512] == (1, 3, 256, 768) and shapes[700] == (1, 3, 256, 768) and shapes[37] == (1, 256, 768)


def test_evaluation_loop_reproduces_reference_test(config1, capsys):
    """harness.test() driven by the oracle model vs the capture of the reference's own test() run."""
    g, args, gt, sd = config1
    model = orc.OracleMMFMIL(sd, orc.OracleConfig())
    roc, ap = harness.test(args, model, harness.get_test_loader(args), 256, None, gt, "cpu")
    res = harness.test.last_result
    scores = np.concatenate(res["scores"])
    assert scores.shape == g["scores"].shape
    assert np.abs(scores - g["scores"]).max() < 2e-6
    assert abs(roc - float(g["roc"])) < 1e-4 and abs(ap - float(g["ap"])) < 1e-4      # AUC to 4 d.p.
    assert abs(res["ano_auc"] - float(g["ano_auc"])) < 1e-4
    out = capsys.readouterr().out
    assert "AUC1:" in out and "Ano-AUC:" in out and out.count("ROC:") == 14
    # printed lines agree with what the reference printed
    ref_lines = [l for l in str(g["stdout"]).splitlines() if l.strip()]
    assert out.splitlines()[0] == ref_lines[0]


def test_cross_video_batching_and_empty_chunk_skip_keep_scores(config1):
    g, args, gt, sd = config1
    model = orc.OracleMMFMIL(sd, orc.OracleConfig())
    s1, c1, _, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cpu", "ucfcrime")
    s2, c2, _, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cpu", "ucfcrime", batch_chunks=8)
    assert c1 == c2 == [str(c) for c in g["classes"]]
    for a, b in zip(s1, s2):
        assert a.shape == b.shape and np.abs(a - b).max() < 1e-6


def test_empty_class_raises_like_the_reference(config1):
    g, args, gt, sd = config1
    scores = [np.full(int(n), 0.5, np.float32) for n in g["lengths"][:3]]
    with pytest.raises(ValueError):       # np.concatenate([]) -- test.py:166-167
        harness.evaluate_scores(scores, [str(c) for c in g["classes"][:3]], gt[:16 * int(g["lengths"][:3].sum())],
                                "ucfcrime", verbose=False)


def test_partition_by_snippets_is_contiguous_and_balanced():
    lengths = synth.lognormal_lengths(1, 290, 69500)
    for world in (1, 2, 3, 8):
        parts = harness.partition_by_snippets(lengths, world)
        assert parts[0][0] == 0 and parts[-1][1] == len(lengths)
        assert all(parts[r][1] == parts[r + 1][0] for r in range(world - 1))
        loads = [int(lengths[a:b].sum()) for a, b in parts]
        assert sum(loads) == int(lengths.sum())
        assert max(loads) <= lengths.sum() / world + lengths.max()


def test_device_auc_ap_has_no_host_fallback():
    """The metric tail is a library entry (iefvad_auc_ap, csrc/metrics.h; parity vs sklearn: tests/test_gpu_metrics.py).  On host
    tensors it refuses instead of computing something else; the host route is evaluate_scores (sklearn, as the reference)."""
    with pytest.raises(RuntimeError, match="HIP device only"):
        harness.device_auc_ap(torch.rand(64), torch.zeros(64 * 16))


def test_perturbation_sweep_reproduces_reference_run_test(golden_dir):
    """harness.run_perturbation_test driven by the oracle vs the capture of test2.run_test (same torch RNG
    stream -> same perturbed time steps)."""
    g = np.load(os.path.join(golden_dir, "sweep_test2.npz"))
    lengths, seed = [int(v) for v in g["lengths"]], int(g["seed"])
    gt = synth.make_gt(seed, sum(lengths))
    model = orc.OracleMMFMIL(synth.make_state_dict(int(g["wseed"])), orc.OracleConfig())

    def loader():
        for i, n in enumerate(lengths):
            img, ev = synth.make_video(seed, i, n)
            ci, _ = harness.process_split(img, 256)
            ce, _ = harness.process_split(ev, 256)
            yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([n])

    args = argparse.Namespace(visual_length=256)
    torch.manual_seed(0)
    cache = {}
    for tag, kw in (("img02", dict(sigma_img=0.2, sigma_ev=0)), ("ev03", dict(sigma_img=0, sigma_ev=0.3))):
        r = harness.run_perturbation_test(args, model, loader(), gt, "cpu", clean_cache=cache, **kw)
        assert np.allclose([float(x) for x in r[:10]], g[tag + "_scalars"], rtol=0, atol=2e-6), tag
        assert np.abs(r[10].numpy() - g[tag + "_w_img_change"]).max() < 2e-6
        assert np.abs(r[11].numpy() - g[tag + "_w_ev_change"]).max() < 2e-6
    assert cache["sweep"].clean_passes == 1      # the clean pass ran once, not once per level
    assert r.auc == r[4] and r.w_ev_change is r[11]


def test_streaming_file_pipeline_matches_per_video_loop(config1):
    """FeatureFilePipeline + evaluate_files (threads, cross-video packing, empty-chunk drop, device metric tail)
    vs the reference-pattern loop on the same .npy files."""
    g, args, gt, sd = config1
    model = orc.OracleMMFMIL(sd, orc.OracleConfig())
    res = harness.evaluate_files(args, model, gt, "cpu", batch_chunks=6, workers=3)
    scores = np.concatenate(res["scores"])
    assert res["snippets"] == int(g["lengths"].sum()) and scores.shape == g["scores"].shape
    assert np.abs(scores - g["scores"]).max() < 2e-6
    assert abs(res["roc"] - float(g["roc"])) < 1e-4 and abs(res["ap"] - float(g["ap"])) < 1e-4
    assert res["classes"] == [str(c) for c in g["classes"]]
    # chunk accounting: a 256-snippet video occupies ONE chunk here (the reference's all-zero second chunk is dropped)
    pipe = harness.FeatureFilePipeline([p for p in open(args.test_list).read().split()[1:] for p in [p.split(",")[0]]],
                                       ["x"] * 16, 256, "event_thr_10", "cpu", batch_chunks=1000)
    (img, ev, meta), = list(pipe.batches())
    by_len = {n: nch for _, n, nch in meta}
    assert by_len[256] == 1 and by_len[257] == 2 and by_len[512] == 2 and by_len[37] == 1 and by_len[1500] == 6
    assert img.shape[0] == sum(nch for _, _, nch in meta)


def test_mixed_dtype_between_modalities_is_widened_not_narrowed():
    """fp16 image features with fp32 event features: the reference casts each modality with `.to(torch.float)`
    (imf_vad.py:41-42), so the event features must reach the model at full precision (not cast to the image dtype)."""
    seen = []

    def model(img, ev, *_):
        seen.append((img.dtype, ev.dtype, ev.clone()))
        return {"logits": torch.zeros(img.shape[0], img.shape[1], 1), "w_i": torch.zeros_like(img, dtype=torch.float32),
                "w_e": torch.zeros_like(img, dtype=torch.float32)}

    img, ev = synth.make_video(3, 0, 40)
    ci, _ = harness.process_split(img.astype(np.float16), 256)
    ce, _ = harness.process_split(ev, 256)
    item = (torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([40]))
    for bc in (0, 4):
        seen.clear()
        harness.score_loader(model, [item], 256, "cpu", "ucfcrime", batch_chunks=bc)
        assert seen[0][0] == seen[0][1] == torch.float32
        assert torch.equal(seen[0][2][0, :40], torch.from_numpy(ev))          # event features bit-exact, not fp16-rounded


def test_pipeline_rejects_event_file_of_another_length(tmp_path):
    d = tmp_path / "rgb"
    d.mkdir()
    (tmp_path / "event_thr_10").mkdir()
    p = str(d / "a__5.npy")
    np.save(p, np.zeros((40, 768), np.float32))
    np.save(p.replace("rgb", "event_thr_10"), np.zeros((41, 768), np.float32))
    pipe = harness.FeatureFilePipeline([p], ["Normal"], 256, "event_thr_10", "cpu")
    with pytest.raises(ValueError, match="do not match"):
        list(pipe.batches())


def test_has_nan_matches_isnan_any():
    """The evaluation loop's NaN presence check (conditional nan_to_num, test.py:90-95) is a sum first and an exact scan
    only when the sum is NaN; +inf and -inf together (NaN sum, no NaN element) must not count as a NaN."""
    x = torch.randn(2, 256, 768)
    assert not harness._has_nan(x)
    x[1, 3, 4] = float("nan")
    assert harness._has_nan(x)
    y = torch.randn(2, 256, 768)
    y[0, 0, 0], y[0, 0, 1] = float("inf"), float("-inf")
    assert not harness._has_nan(y) and not torch.isnan(y).any()
    h = (torch.randn(2, 256, 768) * 60000).half()            # fp16 values whose fp16 sum would overflow
    assert not harness._has_nan(h)
    h[0, 0, 0] = float("nan")
    assert harness._has_nan(h)
    assert not harness._has_nan(torch.zeros(3, dtype=torch.int32))


# ------------------------------------------------------------------------------------------------
# the three test() entries against the reference's three call sites
# ------------------------------------------------------------------------------------------------
def test_three_test_entries_bind_the_reference_call_sites_verbatim():
    """The reference calls its three `test()` functions with three different argument lists.  Each call is bound here,
    argument for argument as the reference writes it, against the entry that replaces it; `label_map`, `vis`, `attn`
    must land where the reference's own signature puts them."""
    import inspect
    A, M, L, T, P, G, D = (object() for _ in range(7))
    LM = {"A": "normal"}
    # /root/reference/test.py:380-390:  test(args, model, test_loader, args.visual_length, None, gt, device, attn=False, vis=True)
    b = inspect.signature(harness.test).bind(A, M, L, T, None, G, D, attn=False, vis=True)
    assert b.arguments["device"] is D and b.arguments["attn"] is False and b.arguments["vis"] is True
    # /root/reference/train/ucf_train.py:130-139:  test(args, model, test_loader, args.visual_length, prompt_text, gt, device, vis=...)
    b = inspect.signature(harness.ucf_test).bind(A, M, L, T, P, G, D, vis=True)
    assert b.arguments["prompt_text"] is P and b.arguments["device"] is D and b.arguments["vis"] is True
    assert "attn" not in b.arguments                                    # default False, ucf_test.py:24
    # /root/reference/train/xd_train.py:102-112:  test(args, model, test_loader, args.visual_length, prompt_text, gt, device, label_map, vis=...)
    b = inspect.signature(harness.xd_test).bind(A, M, L, T, P, G, D, LM, vis=False)
    assert b.arguments["label_map"] is LM and b.arguments["vis"] is False and "attn" not in b.arguments
    # positional ORDER of the tails, as the reference declares them (ucf_test.py:24-25: attn, vis; xd_test.py:23-25: label_map, vis, attn)
    names = lambda f: [p.name for p in inspect.signature(f).parameters.values()
                       if p.kind is inspect.Parameter.POSITIONAL_OR_KEYWORD]
    head = ["args", "model", "test_loader", "maxlen", "prompt_text", "gt", "device"]
    assert names(harness.ucf_test) == head + ["attn", "vis"]
    assert names(harness.xd_test) == head + ["label_map", "vis", "attn"]
    assert names(harness.test)[:9] == head + ["attn", "vis"]
    # xd_test.py:68 indexes label_map for every video: a missing map is an error at the call, not a silent skip
    with pytest.raises(TypeError):
        inspect.signature(harness.xd_test).bind(A, M, L, T, P, G, D)


@pytest.fixture(scope="module")
def xdset(tmp_path_factory, golden_dir):
    return H.write_xd_set(tmp_path_factory.mktemp("xd"), golden_dir)


def test_xd_test_entry_reproduces_the_reference_capture(xdset, capsys):
    """`harness.xd_test`, called as xd_train.py:102-112 calls it, against the reference's own run on the XD-shaped set
    (root test.test with dataset 'xd' -- the same loop; tests/golden/make_golden.py::gen_harness_xd_case)."""
    from sklearn.metrics import roc_auc_score
    g, args, gt, sd, label_map = xdset
    model = orc.OracleMMFMIL(sd, orc.OracleConfig(num_refinement_steps=int(g["K"])))
    logged = []
    ret = harness.xd_test(args, model, harness.get_test_loader(args), 256, ["p"], gt, "cpu", label_map, vis=False, log=logged.append)
    assert isinstance(ret, tuple) and len(ret) == 2                    # `AUC, AP = test(...)`, xd_train.py:102
    roc, ap = ret
    res = harness.xd_test.last_result
    assert res["classes"] == [str(c) for c in g["classes"]]            # remapped by the first label field, xd_test.py:68
    scores = np.concatenate(res["scores"])
    assert np.abs(scores - g["scores"]).max() < 2e-6
    assert abs(roc - float(g["roc"])) < 1e-4 and abs(ap - float(g["ap"])) < 1e-4
    # Ano-AUC of xd_test.py:328-346: every class but 'normal', straight from the CAPTURED scores with sklearn
    offs = np.concatenate([[0], np.cumsum(g["lengths"])])
    keep = [i for i, c in enumerate(g["classes"]) if str(c) != "normal"]
    want = roc_auc_score(np.concatenate([gt[16 * offs[i]:16 * offs[i + 1]] for i in keep]),
                         np.repeat(np.concatenate([g["scores"][offs[i]:offs[i + 1]] for i in keep]), 16))
    assert abs(res["ano_auc"] - want) < 1e-4
    out = capsys.readouterr().out.splitlines()
    ref_lines = [l for l in str(g["stdout"]).splitlines() if l.strip()]
    assert out[0] == ref_lines[0]                                       # "AUC1: ..  AP1: .."
    assert [l for l in out if "ROC:" in l] == [l for l in ref_lines if "ROC:" in l]      # no "Total Samples" in xd_test.py:168
    assert set(logged[0]) == {"test/AP1", "test/ROC1", "test/Ano-AUC"} and len(logged) == 1 + 7
    # attn=True as a keyword (xd_test.py:25): the 4-tuple with the list the reference never fills
    r4 = harness.xd_test(args, model, harness.get_test_loader(args), 256, None, gt, "cpu", label_map, False, True)
    assert len(r4) == 4 and r4[2] == [] and r4[3] == res["classes"]


def test_ucf_test_entry_filter_and_total_samples(config1, capsys):
    """`harness.ucf_test` as ucf_train.py:130-139 calls it: same scores / AUC / AP as the capture; the per-class lines end in
    "Total Samples: n" (ucf_test.py:173-174); Ano-AUC leaves out 'Normal' AND 'normal' (ucf_test.py:340)."""
    g, args, gt, sd = config1
    model = orc.OracleMMFMIL(sd, orc.OracleConfig())
    auc, ap = harness.ucf_test(args, model, harness.get_test_loader(args), 256, None, gt, "cpu", vis=False)
    assert abs(auc - float(g["roc"])) < 1e-4 and abs(ap - float(g["ap"])) < 1e-4
    assert abs(harness.ucf_test.last_result["ano_auc"] - float(g["ano_auc"])) < 1e-4   # no 'normal' key in the UCF set: same value
    out = capsys.readouterr().out
    cls_lines = [l for l in out.splitlines() if "ROC:" in l]
    assert len(cls_lines) == 14 and all("\tTotal Samples: " in l for l in cls_lines)
    # the three filters on one synthetic class table (test.py:336 / ucf_test.py:340 / xd_test.py:334)
    one, zero = np.ones(16), np.zeros(16)
    cw_gt = {"Normal": [np.concatenate([one, zero])], "normal": [np.concatenate([zero, one])], "x": [np.concatenate([one, zero])]}
    cw_pr = {"Normal": [np.array([0.9, 0.2])], "normal": [np.array([0.8, 0.3])], "x": [np.array([0.6, 0.5])]}
    assert harness.compute_ano_auc(cw_gt, cw_pr, normal_keys=("Normal", "normal")) == 1.0      # 'x' alone: perfectly ranked
    a_root = harness.compute_ano_auc(cw_gt, cw_pr, normal_keys=("Normal",))       # 'normal' (inverted) + 'x'
    a_xd = harness.compute_ano_auc(cw_gt, cw_pr, normal_keys=("normal",))         # 'Normal' + 'x'
    assert a_xd == 1.0 and a_root < 1.0
