"""BASELINE config 2 shape on the GPU: a UCF-Crime-sized synthetic test set (290 videos, ~69.5 k snippets,
heavy-tailed lengths) scored through the product harness + HIP path, against the CPU oracle run in the
reference's per-video pattern.  Gate: scores within the fp32 tolerance, AUC / AP / Ano-AUC equal to 4 d.p.
Plus size-independent properties at a larger batch."""
import argparse

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import harness, synth
from oracle import iefvad_oracle as orc
from tests import helpers as H

pytestmark = pytest.mark.gpu


def ucf_shaped_set(seed=1, n_videos=290, total=69500):
    lengths = synth.lognormal_lengths(seed, n_videos, total)
    abnormal = [c for c in synth.UCF_CLASSES if c != 'Normal']
    classes = ['Normal'] * 150 + [abnormal[i % 13] for i in range(n_videos - 150)]   # 150 normal + 140 abnormal (test.csv)
    rng = np.random.default_rng([seed, 11])
    order = rng.permutation(n_videos)
    classes = [classes[i] for i in order]
    return lengths, classes


def items(seed, lengths, classes):
    for i, (n, c) in enumerate(zip(lengths, classes)):
        img, ev = synth.make_video(seed, i, int(n))
        ci, _ = harness.process_split(img, 256)
        ce, _ = harness.process_split(ev, 256)
        yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), (c,), torch.tensor([int(n)])


def gpu_model(sd, **kw):
    args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=10, lambda_ref=0.5,
                              noise_model="StudentT", nu=8)
    m = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args, **kw)
    m.load_state_dict(sd)
    return m.to("cuda:0").eval()


def test_ucf_shaped_set_auc_parity():
    seed = 1
    lengths, classes = ucf_shaped_set(seed)
    total = int(lengths.sum())
    assert 60000 < total < 80000 and len(lengths) == 290
    gt = synth.make_gt(seed, total)
    sd = synth.make_state_dict(7)
    model = gpu_model(sd, outputs="scores")
    s_gpu, c_gpu, _, _ = harness.score_loader(model, items(seed, lengths, classes), 256, "cuda:0", "ucfcrime",
                                              batch_chunks=256)
    torch.set_num_threads(16)
    oracle = orc.OracleMMFMIL(sd, orc.OracleConfig())
    s_cpu, c_cpu, _, _ = harness.score_loader(oracle, items(seed, lengths, classes), 256, "cpu", "ucfcrime")
    assert c_gpu == c_cpu == classes
    a, b = np.concatenate(s_gpu), np.concatenate(s_cpu)
    assert a.shape == b.shape == (total,)
    assert np.abs(a - b).max() <= H.TOL_SIGMOID
    r_gpu = harness.evaluate_scores(s_gpu, classes, gt, "ucfcrime", verbose=False)
    r_cpu = harness.evaluate_scores(s_cpu, classes, gt, "ucfcrime", verbose=False)
    for k in ("roc", "ap", "ano_auc"):
        assert abs(r_gpu[k] - r_cpu[k]) < 1e-4, (k, r_gpu[k], r_cpu[k])     # AUC equal to 4 d.p.
    print("config-2 shape: snippets", total, "max|dscore|", float(np.abs(a - b).max()), "AUC", r_gpu["roc"], r_cpu["roc"])


def test_chunk_permutation_equivariance_at_scale():
    """Chunks are independent batch rows (attention never crosses a chunk, imf_vad.py:115): permuting the
    B=512 chunks of a call permutes the outputs, bit for bit, whatever micro-batch they land in."""
    sd = synth.make_state_dict(8)
    model = gpu_model(sd, outputs="scores", micro_batch=96)
    g = torch.Generator(device="cuda:0")
    g.manual_seed(5)
    img = torch.randn(512, 256, 768, device="cuda:0", generator=g) * 0.45
    ev = torch.randn(512, 256, 768, device="cuda:0", generator=g) * 0.45
    perm = torch.randperm(512, device="cuda:0", generator=g)
    with torch.no_grad():
        a = model(img, ev, None, None, None)
        b = model(img[perm].contiguous(), ev[perm].contiguous(), None, None, None)
    for k in a:
        assert torch.equal(a[k][perm], b[k]), k
    assert bool(torch.isfinite(a["logits"]).all())
    # zero padding after the valid rows changes the valid rows' scores (unmasked attention, Appendix C-1) ...
    img2, ev2 = img[:2].clone(), ev[:2].clone()
    img2[:, 100:], ev2[:, 100:] = 0, 0
    with torch.no_grad():
        c = model(img2, ev2, None, None, None)
    assert not torch.equal(c["logits"][:, :100], a["logits"][:2, :100])
    # ... and an all-zero chunk yields identical rows (every row sees the same keys)
    z = torch.zeros(1, 256, 768, device="cuda:0")
    with torch.no_grad():
        d = model(z, z, None, None, None)
    assert float((d["logits"] - d["logits"][0, 0]).abs().max()) < 1e-6


def test_perturbation_sweep_on_gpu_matches_reference_capture(golden_dir):
    """The robustness sweep (test2.py:35-123) through the HIP path vs the reference's own run_test capture."""
    import os
    g = np.load(os.path.join(golden_dir, "sweep_test2.npz"))
    lengths, seed = [int(v) for v in g["lengths"]], int(g["seed"])
    gt = synth.make_gt(seed, sum(lengths))

    def loader():
        for i, n in enumerate(lengths):
            img, ev = synth.make_video(seed, i, n)
            ci, _ = harness.process_split(img, 256)
            ce, _ = harness.process_split(ev, 256)
            yield torch.tensor(ci).unsqueeze(0), torch.tensor(ce).unsqueeze(0), ("Normal",), torch.tensor([n])

    args = argparse.Namespace(visual_length=256)
    for outputs in ("weights", "full"):      # "weights": logits + w_i / w_e + their row means from the kernels; "full": the drop-in dict
        model = gpu_model(synth.make_state_dict(int(g["wseed"])), outputs=outputs)
        torch.manual_seed(0)
        cache = {}
        for tag, kw in (("img02", dict(sigma_img=0.2, sigma_ev=0)), ("ev03", dict(sigma_img=0, sigma_ev=0.3))):
            r = harness.run_perturbation_test(args, model, loader(), gt, "cuda:0", clean_cache=cache, **kw)
            assert np.allclose([float(x) for x in r[:10]], g[tag + "_scalars"], rtol=0, atol=2e-6), (outputs, tag)
            assert np.abs(r[10].numpy() - g[tag + "_w_img_change"]).max() < 2e-6
            assert np.abs(r[11].numpy() - g[tag + "_w_ev_change"]).max() < 2e-6
        sweep = cache["sweep"]
        assert sweep.clean_passes == 1 and all(t[0].is_cuda for t in sweep.packed)      # uploaded once, clean pass once


@pytest.mark.parametrize("compute", ["f32", "bf16", "bf16x6"])
@pytest.mark.parametrize("dtype", [torch.float32, torch.float16])
def test_row_scale_in_the_input_load_equals_scaling_the_tensor(compute, dtype):
    """`iefvad_forward_scaled` (test2.py:71-77's `x[:, idx] = x[:, idx] * 0.01` folded into the library's input load): the forward on
    (x, row_scale) must equal, bit for bit, the forward on the tensor torch scaled itself -- fp32 rows multiply in fp32, fp16
    rows are rounded to fp16 after the product as torch stores them; one modality scaled, the other untouched (NULL vector);
    B = 9 chunks so the bf16x6 handle runs its split kernels and the bf16 one spans a partial tile."""
    model = gpu_model(synth.make_state_dict(5), outputs="scores", compute=compute)
    model.to("cuda:0").eval()
    B = 9
    img, ev = synth.make_inputs(77, B)
    img, ev = torch.from_numpy(img).to(dtype).cuda(), torch.from_numpy(ev).to(dtype).cuda()
    gen = torch.Generator().manual_seed(1)
    sc = torch.ones(B, 256)
    for b in range(B):
        sc[b, torch.randperm(256, generator=gen)[:77]] = 0.01
    sc = sc.reshape(-1).cuda()
    with torch.no_grad():
        got = model(img, ev, None, None, None, row_scale=(sc, None))
        scaled = img.clone()
        rows = sc.reshape(B, 256) != 1
        scaled[rows] = scaled[rows] * 0.01
        want = model(scaled, ev, None, None, None)
        clean = model(img, ev, None, None, None)
        got_e = model(img, ev, None, None, None, row_scale=(None, sc))
        scaled_e = ev.clone()
        scaled_e[rows] = scaled_e[rows] * 0.01
        want_e = model(img, scaled_e, None, None, None)
    for k in want:
        assert torch.equal(got[k], want[k]), k
        assert torch.equal(got_e[k], want_e[k]), k
    assert not torch.equal(got["logits"], clean["logits"])
    # the scale vector is indexed by the row's position in the CALL: three internal passes of four, four and one chunk see their own slices
    small = gpu_model(synth.make_state_dict(5), outputs="scores", compute=compute, micro_batch=4)
    with torch.no_grad():
        got_mb = small(img, ev, None, None, None, row_scale=(sc, sc))
        both = scaled.clone()
        want_mb = small(both, scaled_e, None, None, None)
    for k in want_mb:
        assert torch.equal(got_mb[k], want_mb[k]), k
    with pytest.raises(ValueError, match="row_scale"):
        model(img, ev, None, None, None, row_scale=(sc[:-1], None))


def test_full_config4_batch_properties():
    """BASELINE config 4 at its full single-GPU size (B = 8192 chunks = 2,097,152 snippets, inputs resident in
    HBM), checked through size-independent properties: every score finite; chunks picked from the big call
    are bit-identical to the same chunks forwarded alone (micro-batching, tile selection and position in the
    batch do not matter); duplicated chunks give duplicated scores."""
    sd = synth.make_state_dict(9)
    model = gpu_model(sd, outputs="scores")
    g = torch.Generator(device="cuda:0")
    g.manual_seed(1234)
    B = 8192
    img = torch.randn(B, 256, 768, device="cuda:0", generator=g) * 0.45
    ev = torch.randn(B, 256, 768, device="cuda:0", generator=g) * 0.45
    img[4097], ev[4097] = img[11], ev[11]                     # a duplicated chunk far away in another micro-batch
    with torch.no_grad():
        out = model(img, ev, None, None, None)
    lg = out["logits"]
    assert lg.shape == (B, 256, 1) and bool(torch.isfinite(lg).all())
    assert bool(torch.isfinite(out["w_i_mean"]).all()) and bool(torch.isfinite(out["w_e_mean"]).all())
    assert torch.equal(lg[11], lg[4097])
    s = torch.sigmoid(lg)
    assert 0.0 < float(s.min()) and float(s.max()) < 1.0
    assert float((out["w_i_mean"] + out["w_e_mean"] - 1.0).abs().max()) < 1e-5      # n_i + n_e = 1 up to eps
    pick = [0, 255, 256, 4097, 8191]
    with torch.no_grad():
        sub = model(img[pick].contiguous(), ev[pick].contiguous(), None, None, None)
    assert torch.equal(sub["logits"], lg[pick])
    del img, ev, out, sub
    torch.cuda.empty_cache()
