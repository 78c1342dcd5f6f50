"""The CPU oracle (oracle/iefvad_oracle.py) against the golden vectors captured from the
reference model itself (tests/golden/make_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import iefvad_oracle as orc
from tests import helpers as H


@pytest.mark.parametrize("name", H.golden_cases())
def test_oracle_matches_reference_outputs(name):
    g, cfg, sd, img, ev = H.load_case(name)
    out = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), H.oracle_cfg(cfg))
    out = {k: v.numpy() for k, v in out.items()}
    errs = H.compare_outputs(out, g)
    # per-row means that the harness derives (test.py:131-136)
    assert np.abs(out["w_i"].mean(-1) - g["w_i_mean"]).max() < 2e-6
    assert np.abs(out["w_e"].mean(-1) - g["w_e_mean"]).max() < 2e-6
    print(name, errs)


@pytest.mark.parametrize("name", H.golden_cases(big=True))
def test_oracle_matches_reference_outputs_at_a_split_sized_batch(name):
    """B = 48 (large enough for the split kernels of the bf16x6 / fp16x3 arithmetics): all chunks' logits and weight
    means, the 768-d outputs of three chunks, against the reference's own outputs."""
    g, cfg, sd, img, ev = H.load_case(name)
    torch.set_num_threads(8)
    out = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), H.oracle_cfg(cfg))
    out = {k: v.numpy() for k, v in out.items()}
    H.compare_outputs(out, g)
    assert np.abs(out["w_i"].mean(-1) - g["w_i_mean"]).max() < 2e-6
    assert np.abs(out["w_e"].mean(-1) - g["w_e_mean"]).max() < 2e-6


def test_oracle_fp64_noise_floor():
    """fp32 vs fp64 evaluation of the same restatement: the noise floor the fp32 gates sit on."""
    g, cfg, sd, img, ev = H.load_case("base_k10_student8")
    o32 = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), H.oracle_cfg(cfg), torch.float32)
    o64 = orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), H.oracle_cfg(cfg), torch.float64)
    for k in o32:
        d = (o32[k].double() - o64[k]).abs().max().item()
        assert d < 1e-5, (k, d)


def test_oracle_rejects_unknown_noise_model():
    g, cfg, sd, img, ev = H.load_case("l1_k2")
    c = H.oracle_cfg(cfg)
    c.noise_model = "Laplace"
    with pytest.raises(ValueError):   # reference: imf_vad.py:137-138
        orc.forward(sd, torch.from_numpy(img), torch.from_numpy(ev), c)


def test_process_split_shapes():
    """Chunk rule of data/tools.py:100-114 (shapes probed on the reference)."""
    for n, shape in [(100, (256, 768)), (37, (256, 768)), (256, (2, 256, 768)), (300, (2, 256, 768)),
                     (512, (3, 256, 768)), (700, (3, 256, 768))]:
        f = np.ones((n, 768), np.float32)
        out, ln = orc.process_split(f, 256)
        assert out.shape == shape and ln == n
        assert out.reshape(-1, 768)[:n].min() == 1 and out.reshape(-1, 768)[n:].max(initial=0) == 0


def test_oracle_harness_scores_match_reference_test_loop(golden_dir):
    """Per-video sigmoid scores of config 1 vs the scores captured inside the reference's test()."""
    import os
    from iefvad_amd import synth
    g = np.load(os.path.join(golden_dir, "harness_config1.npz"))
    lengths = [int(v) for v in g["lengths"]]
    seed = int(g["seed"])
    sd = synth.make_state_dict(int(g["wseed"]))
    model = orc.OracleMMFMIL(sd, orc.OracleConfig())
    # the 4 shortest and 2 multi-chunk videos keep the CPU suite fast; the full set runs in test_harness
    pick = [0, 1, 3, 5, 13]
    offs = np.concatenate([[0], np.cumsum(lengths)])
    vids = []
    for i in pick:
        img, ev = synth.make_video(seed, i, lengths[i])
        if i == 2:
            img[5, 7] = np.nan
        if i == 5:
            img, ev = img.astype(np.float16), ev.astype(np.float16)
        vids.append((img, ev))
    scores = orc.score_videos(model, vids)
    for i, s in zip(pick, scores):
        ref = g["scores"][offs[i]:offs[i + 1]]
        assert s.shape == ref.shape
        assert np.abs(s - ref).max() < 2e-6


def test_oracle_loss_terms_match_the_reference_fixture(golden_dir):
    """f-4: the oracle's restatement of CLAS2 / regulariser / KL against tests/golden/loss_terms.npz, which make_golden.py
    produced with the reference's own CLAS2 (train/loss.py:18-30) and the torch calls of train/ucf_train.py:75-98."""
    import os
    from iefvad_amd import synth
    g = np.load(os.path.join(golden_dir, "loss_terms.npz"))
    for seed in (1, 2):
        o, labels, lengths = synth.make_loss_inputs(seed)
        want = g[f"seed{seed}"]
        for j, (noise, nu) in enumerate((("Gaussian", 8), ("StudentT", 8), ("StudentT", 5))):
            t = orc.loss_terms(o["logits"], o["image_mu"], o["event_mu"], o["image_logvar"], o["event_logvar"], labels, lengths,
                               noise_model=noise, nu=nu)
            got = [t["classification"], t["cos"], t["norm"], t["kl_image"], t["kl_event"]]
            ref = [want[0], want[1], want[2], want[3 + 2 * j], want[4 + 2 * j]]
            assert np.allclose(got, ref, rtol=0, atol=2e-6), (seed, noise, nu, got, ref)


def test_loss_gradient_fixture_is_consistent_with_the_loss_fixture(golden_dir):
    """loss_grads.npz (autograd through the reference's CLAS2 + the trainers' torch calls) was generated from the same inputs as
    loss_terms.npz: its totals must be the sums of that fixture's terms, and its tie-invariant sums must balance."""
    import os
    terms = np.load(os.path.join(golden_dir, "loss_terms.npz"))
    grads = np.load(os.path.join(golden_dir, "loss_grads.npz"))
    for seed, j in ((1, 1), (2, 0)):                       # seed 1: StudentT nu = 8 (1, 1); seed 2: Gaussian (0.01, 0.01)
        t = terms[f"seed{seed}"]
        nu, lam_reg, lam_kl, student = grads[f"seed{seed}_cfg"]
        want = t[0] + lam_reg * (t[1] + t[2]) + lam_kl * (t[3 + 2 * j] + t[4 + 2 * j])
        assert abs(float(grads[f"seed{seed}_total"]) - want) <= 1e-6 * max(1.0, abs(want))      # fp32 terms there, fp64 here
        assert np.isfinite(grads[f"seed{seed}_logits_video_sums"]).all()
        assert np.abs(grads[f"seed{seed}_image_mu_zero_row"]).max() > 1e12          # the clamp branch of F.normalize


# ---- the VadCLIP-residue classes, oracle restatement vs the reference's own outputs ---------------------------
def _vadclip_setup(golden_dir):
    """The fixture's modules rebuilt from the product's mirrors (same seed -> same init -> same perturbation): their state_dicts feed
    the oracle.  Returns (fixture, inputs, state_dicts)."""
    import os
    import torch
    from iefvad_amd import layers as PL, module as PM, synth
    g = np.load(os.path.join(golden_dir, "vadclip_modules.npz"))
    mods = {}
    for key, seed, pseed, make in (("sim", 101, 1, lambda: PL.SimilarityAdj(768, 384)),
                                   ("gc1", 102, 2, lambda: PL.GraphConvolution(768, 384, residual=True)),
                                   ("gc2", 103, 3, lambda: PL.GraphConvolution(384, 384, bias=True, residual=True)),
                                   ("gat", 104, 4, lambda: PL.GraphAttentionLayer(768, 128, dropout=0.0, alpha=0.2, concat=True)),
                                   ("blk", 105, 5, lambda: PM.ResidualAttentionBlock(768, 8, None))):
        torch.manual_seed(seed)
        m = synth.perturb_module(make(), pseed).eval()
        chk = np.array([[float(p.detach().double().sum()), float(p.detach().double().abs().sum())] for _, p in m.named_parameters()])
        assert chk.shape == g[key + "_checks"].shape and np.allclose(chk, g[key + "_checks"], rtol=1e-12, atol=1e-9), key   # same registration order, same initialisers
        mods[key] = m
    x = torch.from_numpy(synth.smooth_features(int(g["x_seed"]), 2))
    xs = torch.from_numpy(synth.smooth_features(4, 2)).permute(1, 0, 2).contiguous() * 0.3
    return g, x, xs, mods


def test_vadclip_module_oracles_match_the_reference_outputs(golden_dir):
    """oracle/vadclip_oracle.py against what the reference's own classes computed (tests/golden/make_golden.py::gen_vadclip_cases):
    SimilarityAdj with and without seq_len, GraphConvolution with the Conv1d and the identity residual, GraphAttentionLayer incl. rows
    without any edge, ResidualAttentionBlock plain and with attn_mask + key padding mask.  fp32 gates."""
    import torch
    from oracle import vadclip_oracle as vo
    g, x, xs, mods = _vadclip_setup(golden_dir)
    rows = [int(r) for r in g["rows"]]
    sd = {k: {n: p.detach() for n, p in m.named_parameters()} for k, m in mods.items()}
    assert float(g["sim_threshold_margin"]) > 2e-5
    adj_full = vo.similarity_adj(sd["sim"], x, None)
    adj_len = vo.similarity_adj(sd["sim"], x, [200, 256])
    assert np.abs(adj_full[:, rows].numpy() - g["adj_full"]).max() <= 2e-6
    assert np.abs(adj_len[:, rows].numpy() - g["adj_len"]).max() <= 2e-6
    assert float(adj_len[0, 200:].abs().max()) == 0 and float(adj_len[0, :, 200:].abs().max()) == 0
    y1 = vo.graph_convolution(sd["gc1"], x, adj_len)
    assert np.abs(y1[:, rows].numpy() - g["gc1_out"]).max() <= 2e-5
    x1 = vo.quick_gelu(y1)
    assert np.abs(x1[:, rows].numpy() - g["gc1_gelu"]).max() <= 2e-5
    y2 = vo.graph_convolution(sd["gc2"], x1, adj_len)
    assert np.abs(y2[:, rows].numpy() - g["gc2_out"]).max() <= 2e-5
    gen = torch.Generator().manual_seed(9)
    gadj = (torch.rand(256, 256, generator=gen) < 0.1).float() * torch.rand(256, 256, generator=gen)
    gadj[[5, 77, 255]] = 0
    got = vo.graph_attention(sd["gat"], x[0], gadj, 0.2, True)
    assert np.abs(got[rows + [5, 77]].numpy() - g["gat_out"]).max() <= 2e-5
    yb = vo.residual_attention_block(sd["blk"], xs, 8)
    assert np.abs(yb[rows].numpy() - g["blk_out"]).max() <= 2e-5
    mask = torch.full((256, 256), float("-inf"))
    for c in range(4):
        mask[c * 64:(c + 1) * 64, c * 64:(c + 1) * 64] = 0
    pad = torch.zeros(2, 256, dtype=torch.bool)
    pad[1, 216:] = True
    ym = vo.residual_attention_block(sd["blk"], xs, 8, mask, pad)
    assert np.abs(ym[rows].numpy() - g["blk_masked_out"]).max() <= 2e-5
    # DistanceAdj: parity unpinned (layers.py:176,178 need 'cuda'); the restated formula against scipy's pdist, as the source computes it
    from scipy.spatial.distance import pdist, squareform
    d = squareform(pdist(np.arange(256).reshape(-1, 1), metric="cityblock").astype(np.float32))
    want = np.exp(-d / np.exp(np.float32(1.0)))
    assert np.abs(vo.distance_adj(2, 256).numpy() - want[None]).max() <= 1e-6
