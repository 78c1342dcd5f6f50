"""Seeded synthetic weights, feature blocks and video sets.

No trained checkpoint or feature file ships with the reference (Google-Drive links,
/root/reference/README.md:63-64,99-100), so every config of BASELINE.json is exercised on
synthetic data of the reference's shapes.  numpy `Generator` streams are stable across
numpy versions, so the same (seed, index) pairs regenerate identical tensors in the
build container and on the GPU box; that is what lets `tests/golden/*.npz` hold only the
reference's *outputs*.
"""
from __future__ import annotations

import math
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

MODALITIES = ("image", "event")

# class keys of the reference's per-class tables (/root/reference/test.py:20-43)
UCF_CLASSES = ['Abuse', 'Arrest', 'Arson', 'Assault', 'Burglary', 'Explosion', 'Fighting', 'RoadAccidents',
               'Robbery', 'Shooting', 'Shoplifting', 'Stealing', 'Vandalism', 'Normal']


def state_dict_keys(num_layers: int = 2, num_steps: int = 10) -> List[Tuple[str, Tuple[str, ...], str]]:
    """(key, shape template, kind) for every tensor of the reference's `state_dict`, in its
    registration order (/root/reference/model/imf_vad.py:69-107)."""
    out = []
    for m in MODALITIES:
        for l in range(num_layers):
            p = f"temporal.{m}_attn_layers.{l}."
            out += [(p + "in_proj_weight", ("3D", "D"), "mat"), (p + "in_proj_bias", ("3D",), "bias"),
                    (p + "out_proj.weight", ("D", "D"), "mat"), (p + "out_proj.bias", ("D",), "bias")]
        for l in range(num_layers):
            p = f"temporal.{m}_norms.{l}."
            out += [(p + "weight", ("D",), "gamma"), (p + "bias", ("D",), "bias")]
    for m in MODALITIES:
        out += [(f"temporal.whiten_{m}.weight", ("D",), "gamma"), (f"temporal.whiten_{m}.bias", ("D",), "bias")]
    for head in ("image_mu", "event_mu", "image_logvar", "event_logvar"):
        out += [(f"temporal.{head}.weight", ("D", "D"), "mat"), (f"temporal.{head}.bias", ("D",), "bias")]
    for k in range(num_steps):
        for j in (0, 2):
            p = f"temporal.refinement_blocks.{k}.{j}."
            out += [(p + "weight", ("D", "D"), "mat"), (p + "bias", ("D",), "bias")]
    out += [("temporal.classifier.weight", ("1", "D"), "mat"), ("temporal.classifier.bias", ("1",), "bias")]
    return out


def make_state_dict(seed: int, D: int = 768, num_layers: int = 2, num_steps: int = 10,
                    mat_scale: float = 1.0) -> Dict[str, torch.Tensor]:
    """Seeded weights under the reference's key names and shapes.  Biases and LayerNorm affine
    terms are deliberately non-trivial (torch's default init has zero attention biases and unit
    LayerNorm weights, which would hide indexing mistakes)."""
    sd = {}
    a = mat_scale / math.sqrt(D)
    for idx, (key, shape_t, kind) in enumerate(state_dict_keys(num_layers, num_steps)):
        shape = tuple({"D": D, "3D": 3 * D, "1": 1}[s] for s in shape_t)
        rng = np.random.default_rng([seed, idx])
        if kind == "mat":
            w = rng.uniform(-a, a, size=shape)
        elif kind == "gamma":
            w = 1.0 + rng.uniform(-0.1, 0.1, size=shape)
        else:
            w = rng.uniform(-0.05, 0.05, size=shape)
        sd[key] = torch.from_numpy(w.astype(np.float32))
    return sd


def make_inputs(seed: int, B: int, T: int = 256, D: int = 768, scale: float = 0.45,
                dtype=np.float32) -> Tuple[np.ndarray, np.ndarray]:
    """Seeded N(0, scale^2) image / event feature blocks [B,T,D] (scale ~ CLIP ViT-L/14 per-dim)."""
    rng = np.random.default_rng([seed, 1000003])
    img = (rng.standard_normal((B, T, D)) * scale).astype(dtype)
    ev = (rng.standard_normal((B, T, D)) * scale).astype(dtype)
    return img, ev


def make_video(seed: int, index: int, length: int, D: int = 768, scale: float = 0.45,
               dtype=np.float32) -> Tuple[np.ndarray, np.ndarray]:
    """One synthetic video: image and event feature files of `length` snippets, [length, D]."""
    rng = np.random.default_rng([seed, 7, index])
    img = (rng.standard_normal((length, D)) * scale).astype(dtype)
    ev = (rng.standard_normal((length, D)) * scale).astype(dtype)
    return img, ev


# config 1: at least one video per UCF class key, lengths around the 256 chunk edge
CONFIG1_LENGTHS = [37, 100, 255, 256, 257, 300, 512, 700, 64, 129, 16, 511, 260, 1, 1500, 90]
CONFIG1_CLASSES = UCF_CLASSES + ['Normal', 'Arson']


def make_gt(seed: int, total_snippets: int, p: float = 0.2) -> np.ndarray:
    """Frame-level Bernoulli(p) ground truth, 16 frames per snippet, float64 like the
    reference's gt.npy (/root/reference/test.py:379,129)."""
    rng = np.random.default_rng([seed, 99])
    return (rng.random(16 * total_snippets) < p).astype(np.float64)


def lognormal_lengths(seed: int, n_videos: int, total: int, lo: int = 16, hi: int = 8000) -> np.ndarray:
    """Heavy-tailed video lengths rescaled so that they sum to ~`total` snippets (configs 2, 3, 5)."""
    rng = np.random.default_rng([seed, 5])
    x = np.exp(rng.normal(0.0, 1.0, n_videos))
    x = np.clip(x / x.sum() * total, lo, hi)
    x = np.clip(np.round(x / x.sum() * total), lo, hi).astype(np.int64)
    return x


def exact_lengths(seed: int, n_videos: int, total: int, lo: int = 4, hi: int = 400) -> np.ndarray:
    """`lognormal_lengths` nudged so that the lengths sum to EXACTLY `total` (the longest / shortest videos absorb
    the rounding remainder one snippet at a time, staying inside [lo, hi])."""
    x = lognormal_lengths(seed, n_videos, total, lo, hi)
    diff = int(total - x.sum())
    order = np.argsort(-x, kind="stable")
    i = 0
    while diff != 0:
        j = order[i % n_videos]
        step = 1 if diff > 0 else -1
        if lo <= x[j] + step <= hi:
            x[j] += step
            diff -= step
        i += 1
    assert int(x.sum()) == total
    return x


def config5_lists(golden_dir: str):
    """BASELINE config 5 (ShanghaiTech + MSAD test lists, K = 5): the REAL frame-level ground truth and label order of
    /root/reference/list/{shang,msad}/rgb/vitl/{gt.npy,test.csv} (fixture tests/golden/config5_gt.npz, written by
    make_golden.py) with synthetic video lengths that sum to each gt exactly (8,723 + 9,009 = 17,732 snippets; the
    feature files, hence the true lengths, are not in the container).
    Returns {dataset: (lengths, labels, gt float64)} in list order."""
    import os
    g = np.load(os.path.join(golden_dir, "config5_gt.npz"))
    out = {}
    for seed, d in ((51, "shang"), (52, "msad")):
        frames = int(g[f"{d}_frames"])
        gt = np.unpackbits(g[f"{d}_bits"])[:frames].astype(np.float64)
        labels = [str(x) for x in g[f"{d}_labels"]]
        out[d] = (exact_lengths(seed, len(labels), frames // 16), labels, gt)
    return out


LOSS_LENGTHS = [256, 37, 100, 16, 255, 1, 200, 64]


def make_loss_inputs(seed: int, B: int = 8, T: int = 256, D: int = 768):
    """Seeded stand-ins for a training batch's model outputs (f-4 loss head fixtures): logits [B,T,1], image_mu / event_mu /
    image_logvar / event_logvar [B,T,D], one-hot labels [B,14] (column 0 = normal) and lengths.  One image_mu row is all
    zero (the eps branches of F.normalize / F.cosine_similarity), a few logits tie exactly (top-k among equal scores)."""
    rng = np.random.default_rng([seed, 31])
    logits = (rng.standard_normal((B, T, 1)) * 2.0).astype(np.float32)
    logits[0, 10:14, 0] = logits[0, 9, 0]                       # ties
    logits[2, :, 0] = np.round(logits[2, :, 0] * 2) / 2         # many ties
    out = {"logits": logits}
    for k, sc, sh in (("image_mu", 0.5, 0.0), ("event_mu", 0.4, 0.05), ("image_logvar", 0.3, -0.5), ("event_logvar", 0.4, -0.2)):
        out[k] = (rng.standard_normal((B, T, D)) * sc + sh).astype(np.float32)
    out["image_mu"][1, 5] = 0.0
    labels = np.zeros((B, 14), np.float32)
    cls = [0, 3, 0, 7, 13, 0, 1, 2][:B]
    labels[np.arange(B), cls] = 1.0
    lengths = np.array(LOSS_LENGTHS[:B], np.int64)
    return out, labels, lengths


TRAIN_LENGTHS = [256, 100, 37, 256, 180, 16, 255, 64]


def make_train_batch(seed: int, B: int, T: int = 256, D: int = 768):
    """A seeded training batch in the shape the reference's trainers build (/root/reference/train/ucf_train.py:44-58): the first
    half normal, the second half abnormal videos, each resampled / zero padded to T snippets by the train loader
    (data/tools.py:89-97: `pad` leaves zeros behind a short video's `length` rows), one-hot labels [B, 14] with column 0 =
    normal, lengths [B].  Returns (img, ev, labels, lengths) as numpy arrays."""
    img, ev = make_inputs(seed, B, T, D)
    lengths = np.array([TRAIN_LENGTHS[i % len(TRAIN_LENGTHS)] for i in range(B)], np.int64)
    for i, n in enumerate(lengths):
        img[i, n:] = 0
        ev[i, n:] = 0
    labels = np.zeros((B, 14), np.float32)
    for i in range(B):
        labels[i, 0 if i < (B + 1) // 2 else 1 + (3 * i) % 13] = 1.0
    return img, ev, labels, lengths


def make_dropout_mask(seed: int, L: int, B: int, p: float = 0.1, H: int = 8, T: int = 256) -> np.ndarray:
    """Seeded keep mask of the attention dropout, uint8 [2 modalities, L, B, H, T, T] (1 = keep, probability 1 - p): what a test
    injects into both the reference (through torch.nn.functional.dropout, tests/golden/make_golden.py) and
    `iefvad_amd.MMFMIL.dropout_mask`."""
    rng = np.random.default_rng([seed, 4242])
    return (rng.random((2, L, B, H, T, T), dtype=np.float32) >= p).astype(np.uint8)


# ---- seeded inputs and parameters for the module-level fixtures of layers.py / module.py --------------------
def perturb_module(module, seed: int, scale: float = 0.03):
    """Add seeded noise to every parameter in registration order.  The reference's default initialisers leave attention / Linear
    biases at zero and LayerNorm at (1, 0), which would hide a swapped bias or affine term; the fixture generator and the tests call
    this on the reference class and on its mirror after constructing both under the same torch.manual_seed."""
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for _, p in module.named_parameters():
            p.add_(torch.randn(p.shape, generator=g, dtype=torch.float32).to(p.device) * scale)
    return module


def smooth_features(seed: int, B: int, T: int = 256, D: int = 768) -> np.ndarray:
    """[B, T, D] features with temporal structure (a slow random walk plus noise): neighbouring snippets have cosine similarity well
    above SimilarityAdj's 0.7 threshold, distant ones below it, as consecutive CLIP embeddings of a video do."""
    rng = np.random.default_rng([seed, 31])
    walk = np.cumsum(rng.standard_normal((B, T, D)) * 0.25, axis=1)
    x = walk + rng.standard_normal((B, T, D)) * 0.35 + rng.standard_normal((B, 1, D)) * 0.5
    return x.astype(np.float32)
