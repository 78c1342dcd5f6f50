"""The reference's OWN end-to-end capture driven through the HIP path: `tests/golden/harness_config1.npz` holds what
the reference's test() (/root/reference/test.py:46-212) computed and printed on the config-1 synthetic .npy set (16
videos over the 14 UCF class keys, lengths around the 256-chunk edge, one NaN element, one fp16 file).  Here
`harness.test(args, iefvad_amd.MMFMIL(...), loader, ...)` on cuda:0 must reproduce it: per-snippet scores to 2e-6,
ROC / AP / Ano-AUC to 1e-4 (4 d.p.), the printed summary lines.  Plus the streaming .npy loader (SURVEY.md 8f-2:
pinned staging, side-stream H2D, record_stream) against the per-video loop on the same files.  `-m gpu`."""
import argparse

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import harness, synth
from tests import helpers as H

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def config1(tmp_path_factory, golden_dir):
    return H.write_config1_set(tmp_path_factory.mktemp("cfg1gpu"), golden_dir)


def gpu_model(sd, **kw):
    args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5,
                              noise_model="StudentT", nu=8)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args, **kw)
    m.load_state_dict(sd)
    return m


@pytest.mark.parametrize("batch_chunks", [0, 8])
@pytest.mark.parametrize("outputs", ["full", "scores"])
def test_reference_test_capture_through_the_hip_path(config1, capsys, batch_chunks, outputs):
    g, args, gt, sd = config1
    model = gpu_model(sd, outputs=outputs)                  # harness.test moves it to the device itself (test.py:57-58)
    roc, ap = harness.test(args, model, harness.get_test_loader(args), 256, None, gt, "cuda:0", batch_chunks=batch_chunks)
    res = harness.test.last_result
    scores = np.concatenate(res["scores"])
    assert scores.shape == g["scores"].shape
    assert np.abs(scores - g["scores"]).max() <= H.TOL_SIGMOID
    assert abs(roc - float(g["roc"])) < 1e-4 and abs(ap - float(g["ap"])) < 1e-4
    assert abs(res["ano_auc"] - float(g["ano_auc"])) < 1e-4
    assert res["classes"] == [str(c) for c in g["classes"]]
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    ref_lines = [ln for ln in str(g["stdout"]).splitlines() if ln.strip()]
    assert out[0] == ref_lines[0] and out[1] == ref_lines[1]                  # "AUC1: .. AP1: ..", "Ano-AUC: .."
    assert sum("ROC:" in ln for ln in out) == 14
    # every per-class line the reference printed, number for number (2 decimals)
    ours = {ln.split(" ROC:")[0]: ln for ln in out if " ROC:" in ln}
    theirs = {ln.split(" ROC:")[0]: ln for ln in ref_lines if " ROC:" in ln}
    assert ours == theirs


def test_per_video_and_packed_scores_are_bit_identical_in_f32(config1):
    g, args, gt, sd = config1
    model = gpu_model(sd, outputs="scores").to("cuda:0").eval()
    s1, _, wi1, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime")
    s2, _, wi2, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime", batch_chunks=8)
    for a, b in zip(s1 + wi1, s2 + wi2):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("outputs", ["scores", "full"])
def test_per_video_loop_on_several_streams_is_bit_identical(config1, outputs):
    """score_loader(lanes=3): consecutive videos go to three HIP streams, each with its own lane of the model (same
    Parameters, own library handle / workspace / pinned staging).  Scores, weight means and their order must equal the
    one-stream loop bit for bit; run twice, so that the cached lanes (and their hipGraphs) are reused; then through
    harness.test against the reference capture; then after load_state_dict of other weights (every lane must notice)."""
    g, args, gt, sd = config1
    model = gpu_model(sd, outputs=outputs).to("cuda:0").eval()
    s1, c1, wi1, we1 = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime")
    for _ in range(2):
        s3, c3, wi3, we3 = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime", lanes=3)
        assert c1 == c3 and len(s1) == len(s3)
        for a, b in zip(s1 + wi1 + we1, s3 + wi3 + we3):
            assert np.array_equal(a, b, equal_nan=True)
    assert len(model.lanes(3)) == 3 and model.lanes(3)[1] is model.lanes(3)[1]
    roc, ap = harness.test(args, model, harness.get_test_loader(args), 256, None, gt, "cuda:0", lanes=3)
    assert abs(roc - float(g["roc"])) < 1e-4 and abs(ap - float(g["ap"])) < 1e-4
    sd2 = synth.make_state_dict(77, 768, 2, 10)
    model.load_state_dict(sd2)
    t3, _, _, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime", lanes=3)
    t1, _, _, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime")
    for a, b in zip(t1, t3):
        assert np.array_equal(a, b, equal_nan=True)
    assert not np.array_equal(np.concatenate(t1), np.concatenate(s1))


@pytest.mark.parametrize("compute", ["f32", "bf16x6"])
def test_streaming_file_pipeline_on_the_gpu(config1, compute):
    """harness.evaluate_files on cuda:0 (header pre-scan, read() into pinned staging, H2D on a side stream,
    wait_event / record_stream, cross-video packing, all-zero chunk drop, device AUC/AP) vs the per-video loop of
    harness.score_loader on the same files, and vs the reference capture."""
    g, args, gt, sd = config1
    model = gpu_model(sd, outputs="scores", compute=compute).to("cuda:0").eval()
    res = harness.evaluate_files(args, model, gt, "cuda:0", batch_chunks=6, workers=3)
    s_loop, classes, _, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime")
    a, b = np.concatenate(res["scores"]), np.concatenate(s_loop)
    assert res["snippets"] == int(g["lengths"].sum()) and a.shape == b.shape == g["scores"].shape
    assert res["classes"] == classes == [str(c) for c in g["classes"]]
    if compute == "f32":
        assert np.array_equal(a, b)                     # bit-reproducible across batch compositions in this mode
    assert np.abs(a - b).max() <= H.TOL_SIGMOID
    assert np.abs(a - g["scores"]).max() <= H.TOL_SIGMOID
    assert abs(res["roc"] - float(g["roc"])) < 1e-4 and abs(res["ap"] - float(g["ap"])) < 1e-4
    # a second pass over the same pipeline object state (fresh staging buffers) gives the same bits
    res2 = harness.evaluate_files(args, model, gt, "cuda:0", batch_chunks=64, workers=2)
    if compute == "f32":
        assert np.array_equal(np.concatenate(res2["scores"]), a)


def test_streaming_pipeline_mixed_dtypes_nan_and_chunk_multiples(tmp_path):
    """Files the config-1 capture does not hold: fp16 image with fp32 event features (widened, not narrowed), NaN and
    +-inf in an fp16 file (conditional nan_to_num with the SOURCE dtype's limits, test.py:90-95), lengths that are
    multiples of 256 (the reference's all-zero extra chunk is dropped; its rows are sliced away by test.py:121)."""
    from oracle import iefvad_oracle as orc
    sd = synth.make_state_dict(3, 768, 1, 1)
    args_m = argparse.Namespace(visual_layers=1, visual_head=8, num_refinement_steps=1, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 1, 8, 10, 10, "cuda", args_m, outputs="scores")
    model.load_state_dict(sd)
    model = model.to("cuda:0").eval()
    lengths = [256, 40, 512, 300, 1]
    rows = []
    for i, n in enumerate(lengths):
        img, ev = synth.make_video(21, i, n)
        if i == 1:
            img = img.astype(np.float16)                       # mixed: fp16 image, fp32 event
        if i == 3:
            img, ev = img.astype(np.float16), ev.astype(np.float16)
            img[7, 3], img[8, 4], ev[9, 5] = np.nan, np.inf, -np.inf
        d = tmp_path / "rgb" / "Normal"
        d.mkdir(parents=True, exist_ok=True)
        (tmp_path / "event_thr_10" / "Normal").mkdir(parents=True, exist_ok=True)
        p = str(d / f"m{i}__5.npy")
        np.save(p, img)
        np.save(p.replace("rgb", "event_thr_10"), ev)
        rows.append(p)
    csv = tmp_path / "t.csv"
    csv.write_text("path,label\n" + "".join(f"{p},Normal\n" for p in rows))
    args = argparse.Namespace(dataset="ucfcrime", visual_length=256, test_list=str(csv))
    gt = synth.make_gt(21, sum(lengths))
    res = harness.evaluate_files(args, model, gt, "cuda:0", batch_chunks=3, workers=2)
    s_loop, _, _, _ = harness.score_loader(model, harness.get_test_loader(args), 256, "cuda:0", "ucfcrime")
    oracle = orc.OracleMMFMIL(sd, orc.OracleConfig(num_layers=1, num_refinement_steps=1))
    s_cpu, _, _, _ = harness.score_loader(oracle, harness.get_test_loader(args), 256, "cpu", "ucfcrime")
    for i, (a, b, c) in enumerate(zip(res["scores"], s_loop, s_cpu)):
        assert a.shape == b.shape == c.shape == (lengths[i],)
        if i != 3:
            assert np.isfinite(a).all()
        # video 3: the ev file holds -inf without a NaN, so the reference leaves it (conditional nan_to_num) and the
        # chunk's scores are NaN in all three
        assert np.array_equal(np.isnan(a), np.isnan(c)) and np.array_equal(np.isnan(b), np.isnan(c)), i
        fin = ~np.isnan(c)
        if fin.any():
            assert np.abs(a[fin] - c[fin]).max() <= H.TOL_SIGMOID and np.abs(b[fin] - c[fin]).max() <= H.TOL_SIGMOID, i


def test_refresh_weights_after_a_data_write():
    """Writes through `param.data` bypass torch's version counter; `refresh_weights()` makes the library re-read."""
    sd = synth.make_state_dict(9, 768, 1, 0)
    args_m = argparse.Namespace(visual_layers=1, visual_head=8, num_refinement_steps=0, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 1, 8, 10, 10, "cuda", args_m, outputs="scores")
    model.load_state_dict(sd)
    model = model.to("cuda:0").eval()
    img, ev = (torch.from_numpy(x).cuda() for x in synth.make_inputs(5, 1))
    with torch.no_grad():
        a = model(img, ev)["logits"].clone()
        model.temporal.classifier.bias.data.add_(1.0)
        model.refresh_weights()
        b = model(img, ev)["logits"].clone()
        model.temporal.classifier.bias.add_(1.0)               # a versioned in-place op is noticed by itself
        c = model(img, ev)["logits"].clone()
    assert torch.allclose(b, a + 1.0, atol=1e-6) and torch.allclose(c, a + 2.0, atol=1e-6)


@pytest.mark.parametrize("batch_chunks", [0, 8])
def test_xd_test_entry_reproduces_the_reference_capture_through_the_hip_path(tmp_path, golden_dir, capsys, batch_chunks):
    """`harness.xd_test` called positionally as /root/reference/train/xd_train.py:102-112 calls `test` (label_map 8th), HIP model:
    the reference's own run on the XD-shaped set (label codes remapped by their first field, seven class keys)."""
    g, args, gt, sd, label_map = H.write_xd_set(tmp_path, golden_dir)
    a = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=int(g["K"]), lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(7, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", a, outputs="scores")
    model.load_state_dict(sd)
    ret = harness.xd_test(args, model, harness.get_test_loader(args), 256, None, gt, "cuda:0", label_map, vis=False,
                          batch_chunks=batch_chunks)
    assert len(ret) == 2
    res = harness.xd_test.last_result
    assert res["classes"] == [str(c) for c in g["classes"]]
    assert np.abs(np.concatenate(res["scores"]) - g["scores"]).max() <= H.TOL_SIGMOID
    assert abs(ret[0] - float(g["roc"])) < 1e-4 and abs(ret[1] - float(g["ap"])) < 1e-4
    out = [ln for ln in capsys.readouterr().out.splitlines() if ln.strip()]
    ref_lines = [ln for ln in str(g["stdout"]).splitlines() if ln.strip()]
    assert out[0] == ref_lines[0]
    assert [ln for ln in out if " ROC:" in ln] == [ln for ln in ref_lines if " ROC:" in ln]


def test_default_entry_takes_the_packed_route_with_the_per_video_bits(config1, capsys):
    """`harness.test / ucf_test / xd_test` without `batch_chunks`: in compute "f32" (and "bf16") the packed route -- the list walked inside
    the library -- gives the bits of one forward per video, so it is the default; "bf16x6" keeps the per-video pattern.  Checked here: the
    default call's scores equal the explicit per-video call's bit for bit, and the capture's numbers."""
    g, args, gt, sd = config1
    for compute in ("f32", "bf16"):
        model = gpu_model(sd, outputs="scores", compute=compute)
        harness.ucf_test(args, model, harness.get_test_loader(args), 256, None, gt, "cuda:0")
        dflt = harness.ucf_test.last_result["scores"]
        harness.ucf_test(args, model, harness.get_test_loader(args), 256, None, gt, "cuda:0", batch_chunks=0)
        per_video = harness.ucf_test.last_result["scores"]
        for a, b in zip(dflt, per_video):
            assert np.array_equal(a, b), compute
        if compute == "f32":
            assert np.abs(np.concatenate(dflt) - g["scores"]).max() <= H.TOL_SIGMOID
    capsys.readouterr()
