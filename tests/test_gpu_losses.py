"""f-4 loss head (`iefvad_loss_forward` / `iefvad_loss_backward`, csrc/loss.h; `iefvad_amd.losses`) against the reference-generated
fixture tests/golden/loss_terms.npz (the reference's CLAS2, train/loss.py:18-30, and the trainers' torch calls,
train/ucf_train.py:75-98) and against the fp64 oracle on other inputs.  Tolerance 1e-6 on every term."""
import os

import numpy as np
import pytest
import torch

import iefvad_amd
from iefvad_amd import losses, synth
from oracle import iefvad_oracle as orc

pytestmark = pytest.mark.gpu
TOL = 1e-6


def to_dev(o):
    return {k: torch.from_numpy(v).cuda() for k, v in o.items()}


def test_loss_terms_match_the_reference_fixture(golden_dir):
    g = np.load(os.path.join(golden_dir, "loss_terms.npz"))
    for seed in (1, 2):
        o, labels, lengths = synth.make_loss_inputs(seed)
        want = g[f"seed{seed}"]
        d = to_dev(o)
        cls = losses.CLAS2(d["logits"], torch.from_numpy(labels).cuda(), torch.from_numpy(lengths).cuda(), "cuda")
        assert cls.shape == () and abs(float(cls) - want[0]) <= TOL
        for j, (noise, nu) in enumerate((("Gaussian", 8), ("StudentT", 8), ("StudentT", 5))):
            t = losses.training_losses(d, torch.from_numpy(labels), lengths, noise_model=noise, nu=nu)
            got = {k: float(v) for k, v in t.items()}
            ref = dict(classification=want[0], cos=want[1], norm=want[2], kl_image=want[3 + 2 * j], kl_event=want[4 + 2 * j])
            for k, v in ref.items():
                assert abs(got[k] - v) <= TOL * max(1.0, abs(v)), (seed, noise, nu, k, got[k], v)
            assert abs(got["reg"] - (want[1] + want[2])) <= 2e-6 and abs(got["kl"] - (ref["kl_image"] + ref["kl_event"])) <= 2e-6
            assert abs(got["total"] - (got["classification"] + got["reg"] + got["kl"])) <= 2e-6


def test_loss_terms_on_model_outputs_match_the_oracle_and_are_deterministic():
    """The terms on what the HIP forward itself returns (full dict, eval mode), xd-style weights (0.01, 0.01), against the
    fp64 oracle evaluated on the same tensors; two calls give the same bits (fixed-order reductions, no atomics)."""
    import argparse
    sd = synth.make_state_dict(23, 768, 2, 2)
    args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=2, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args)
    model.load_state_dict(sd)
    model = model.to("cuda:0").eval()
    img, ev = synth.make_inputs(24, 5)
    with torch.no_grad():
        out = model(torch.from_numpy(img).cuda(), torch.from_numpy(ev).cuda(), None, None, None)
    labels = torch.zeros(5, 14)
    labels[torch.arange(5), torch.tensor([0, 4, 0, 9, 2])] = 1
    lengths = torch.tensor([256, 120, 31, 256, 77])
    a = losses.training_losses(out, labels, lengths, "StudentT", model.temporal.nu, 0.01, 0.01)
    b = losses.training_losses(out, labels, lengths, "StudentT", model.temporal.nu, 0.01, 0.01)
    ref = orc.loss_terms(*(out[k].cpu() for k in ("logits", "image_mu", "event_mu", "image_logvar", "event_logvar")), labels, lengths,
                         "StudentT", 8, 0.01, 0.01)
    for k in losses.TERMS:
        assert torch.equal(a[k], b[k]), k
        assert abs(float(a[k]) - ref[k]) <= TOL * max(1.0, abs(ref[k])), (k, float(a[k]), ref[k])


def test_loss_forward_rejects_bad_arguments():
    o, labels, lengths = synth.make_loss_inputs(1)
    d = to_dev(o)
    with pytest.raises(ValueError, match="Unsupported noise_model"):
        losses.training_losses(d, torch.from_numpy(labels), lengths, noise_model="Laplace")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        losses.CLAS2(torch.from_numpy(o["logits"]), torch.from_numpy(labels), lengths)
    bad = dict(d, logits=d["logits"][:, :128])
    with pytest.raises((RuntimeError, ValueError)):
        losses.training_losses(bad, torch.from_numpy(labels), lengths)


KEYS5 = ("logits", "image_mu", "event_mu", "image_logvar", "event_logvar")


def test_loss_gradients_match_autograd_through_the_reference_loss(golden_dir):
    """`losses.training_loss(...).backward()` (iefvad_loss_backward) against tests/golden/loss_grads.npz: fp64 autograd through
    the reference's own CLAS2 and the trainers' torch calls.  Sampled entries of all five gradients, the per-video sums of
    d logits (tie-invariant: videos 0 and 2 hold exactly tied scores, where torch.topk's choice among equals decides who gets
    the gradient) and the whole row of the all-zero image_mu row, which runs through the clamp branches of F.normalize /
    F.cosine_similarity (a 1e15-scale value in autograd too)."""
    g = np.load(os.path.join(golden_dir, "loss_grads.npz"))
    for seed in (1, 2):
        tag = f"seed{seed}"
        nu, lam_reg, lam_kl, student = g[tag + "_cfg"]
        o, labels, lengths = synth.make_loss_inputs(seed)
        d = {k: torch.from_numpy(v).cuda().requires_grad_(True) for k, v in o.items()}
        total = losses.training_loss(d, torch.from_numpy(labels), lengths, "StudentT" if student else "Gaussian", float(nu),
                                     float(lam_reg), float(lam_kl))
        assert abs(float(total.detach()) - float(g[tag + "_total"])) <= 2e-6 * max(1.0, abs(float(g[tag + "_total"])))
        total.backward()
        for k in KEYS5:
            got = d[k].grad.detach().cpu().numpy().astype(np.float64).reshape(-1)
            idx, want = g[f"{tag}_{k}_idx"], g[f"{tag}_{k}_val"]
            scale = float(g[f"{tag}_{k}_abssum"]) / got.size if k != "image_mu" else np.median(np.abs(want))
            err = np.abs(got[idx] - want)
            assert (err <= 2e-4 * np.abs(want) + 2e-4 * scale).all(), (seed, k, err.max(), scale)
        B, T = d["logits"].shape[0], d["logits"].shape[1]
        sums = d["logits"].grad.detach().cpu().numpy().astype(np.float64).reshape(B, T).sum(1)
        assert np.abs(sums - g[tag + "_logits_video_sums"]).max() <= 1e-6
        assert float(d["logits"].grad.reshape(B, T)[3, int(lengths[3]):].abs().sum()) == 0.0 or int(lengths[3]) == T
        for k in ("image_mu", "event_mu"):
            got = d[k].grad.detach().cpu().numpy().astype(np.float64)[1, 5]
            want = g[f"{tag}_{k}_zero_row"]
            assert np.abs(got - want).max() <= 1e-3 * max(np.abs(want).max(), 1e-12), (seed, k)


def test_loss_gradient_is_the_directional_derivative_of_the_oracle_total():
    """Independent of the fixture: on the HIP model's own outputs, <grad, direction> equals the central difference of the fp64
    oracle's total along a random direction, per tensor; `grad_out` scales the result; unused inputs get no gradient."""
    import argparse
    sd = synth.make_state_dict(29, 768, 2, 2)
    args = argparse.Namespace(visual_layers=2, visual_head=8, num_refinement_steps=2, lambda_ref=0.5, noise_model="StudentT", nu=8)
    model = iefvad_amd.MMFMIL(14, 768, 256, 768, 8, 2, 8, 10, 10, "cuda", args)
    model.load_state_dict(sd)
    model = model.to("cuda:0").eval()
    img, ev = synth.make_inputs(30, 3)
    with torch.no_grad():
        out = model(torch.from_numpy(img).cuda(), torch.from_numpy(ev).cuda(), None, None, None)
    labels = torch.zeros(3, 14)
    labels[torch.arange(3), torch.tensor([0, 6, 1])] = 1
    lengths = torch.tensor([256, 100, 45])
    x = {k: out[k].detach().clone().requires_grad_(True) for k in KEYS5}
    (3.0 * losses.training_loss(x, labels, lengths, "StudentT", 8, 0.5, 0.25)).backward()
    base = {k: out[k].detach().cpu().double() for k in KEYS5}
    gen = torch.Generator().manual_seed(3)
    for k in KEYS5:
        dirn = torch.randn(base[k].shape, generator=gen, dtype=torch.float64)
        if k == "logits":
            dirn[1, 100:] = 0
            dirn[2, 45:] = 0
        eps = 1e-7          # fp64 oracle: small enough that no row crosses the kink of | |mu_i| - |mu_e| | (the heads' norms lie close together)
        tot = []
        for sgn in (1.0, -1.0):
            t = dict(base)
            t[k] = base[k] + sgn * eps * dirn
            tot.append(orc.loss_terms(*(t[q] for q in KEYS5), labels, lengths, "StudentT", 8, 0.5, 0.25)["total"])
        fd = 3.0 * (tot[0] - tot[1]) / (2 * eps)
        an = float((x[k].grad.detach().cpu().double() * dirn).sum())
        assert abs(an - fd) <= 2e-3 * max(abs(fd), 1e-6), (k, an, fd)
    y = {k: out[k].detach().clone() for k in KEYS5}
    y["logits"].requires_grad_(True)
    losses.training_loss(y, labels, lengths).backward()
    assert y["logits"].grad is not None and y["image_mu"].grad is None


def test_adamw_step_matches_torch_optim_adamw():
    """`losses.AdamW` (iefvad_adamw_step) against torch.optim.AdamW on the CPU, constructed as the trainers construct it
    (`AdamW(params, lr=...)`: betas (0.9, 0.999), eps 1e-8, weight_decay 0.01), six steps on two tensors with fresh gradients
    each step; a parameter without a gradient is left alone."""
    g = torch.Generator().manual_seed(5)
    shapes = [(768, 768), (2304,), (5,)]
    ref = [torch.randn(s, generator=g).requires_grad_(True) for s in shapes]
    dev = [r.detach().clone().cuda().requires_grad_(True) for r in ref]
    o_ref = torch.optim.AdamW(ref, lr=3e-4)
    o_dev = losses.AdamW(dev, lr=3e-4)
    for step in range(6):
        for i, (r, d) in enumerate(zip(ref, dev)):
            if i == 2 and step % 2:
                r.grad, d.grad = None, None
                continue
            gr = torch.randn(r.shape, generator=g) * (10.0 ** (step - 3))
            r.grad, d.grad = gr.clone(), gr.clone().cuda()
        o_ref.step()
        o_dev.step()
        for r, d in zip(ref, dev):
            assert float((r.detach() - d.detach().cpu()).abs().max()) <= 2e-6 * float(r.detach().abs().max()), step
    with pytest.raises(ValueError, match="no CPU fallback"):
        losses.AdamW([torch.zeros(4)])
