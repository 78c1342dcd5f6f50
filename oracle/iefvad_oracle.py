"""CPU oracle for the IEF-VAD fusion-inference hot path.  TEST INFRASTRUCTURE ONLY.

This file is the *checker*, never the product: only `tests/`, `__graft_entry__.smoke()` and
the `cpu_baseline` leg of `bench.py` may import it.  The product path (`iefvad_amd`) never
imports anything under `oracle/` and fails loudly when its HIP library is missing.

It restates, with explicit tensor algebra on the CPU (no `nn.MultiheadAttention`,
no `nn.LayerNorm`, no `nn.Linear`), the algorithm of the reference's
`MMFMIL.forward` -> `MultiModal_Fusion_Attn_Iter.forward`
(/root/reference/model/imf_vad.py:40-44 and :109-161), plus the chunking rule of
`process_split` (/root/reference/data/tools.py:100-114) and the per-video scoring loop of
`test()` (/root/reference/test.py:46-163).

Parity pin: the reference has no tests and no golden vectors, so
the pin is the output of the reference model itself, imported in the build container by
`tests/golden/make_golden.py` and committed as `tests/golden/*.npz`.
`tests/test_oracle_golden.py` checks this restatement against those vectors.

Arithmetic lives in torch CPU tensor ops (matmul / exp / mean); `dtype` selects fp32
(the reference's arithmetic) or fp64 (to measure the fp32 noise floor).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

MODALITIES = ("image", "event")


# ----------------------------------------------------------------------------------------
# the restated forward
# ----------------------------------------------------------------------------------------
def _layer_norm(x: torch.Tensor, g: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    # nn.LayerNorm(D): biased variance over the last dim, eps inside the sqrt
    # (/root/reference/model/imf_vad.py:73,80,83-84)
    mu = x.mean(dim=-1, keepdim=True)
    xc = x - mu
    var = (xc * xc).mean(dim=-1, keepdim=True)
    return xc / torch.sqrt(var + eps) * g + b


def _self_attention(x: torch.Tensor, w_in, b_in, w_out, b_out, H: int) -> torch.Tensor:
    # nn.MultiheadAttention(D, H, batch_first=True)(x, x, x) in eval mode: packed in-proj in
    # (q,k,v) order, scale 1/sqrt(D/H), softmax over ALL T keys (no mask), out-proj
    # (/root/reference/model/imf_vad.py:69-72,115,121)
    B, T, D = x.shape
    dh = D // H
    qkv = x @ w_in.t() + b_in
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(B, T, H, dh).transpose(1, 2)
    k = k.reshape(B, T, H, dh).transpose(1, 2)
    v = v.reshape(B, T, H, dh).transpose(1, 2)
    s = (q @ k.transpose(-1, -2)) / math.sqrt(dh)
    s = s - s.max(dim=-1, keepdim=True).values
    p = torch.exp(s)
    p = p / p.sum(dim=-1, keepdim=True)
    a = (p @ v).transpose(1, 2).reshape(B, T, D)
    return a @ w_out.t() + b_out


class OracleConfig:
    def __init__(self, num_layers=2, num_heads=8, num_refinement_steps=10, lambda_ref=0.5,
                 noise_model="StudentT", nu=8, epsilon=1e-8):
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.num_refinement_steps = num_refinement_steps
        self.lambda_ref = lambda_ref
        self.noise_model = noise_model
        self.nu = nu
        self.epsilon = epsilon  # never forwarded by the reference's MMFMIL -> always 1e-8 (imf_vad.py:30-38,58)


def forward(sd: Dict[str, torch.Tensor], image_features, event_features, cfg: OracleConfig,
            dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """Restatement of MultiModal_Fusion_Attn_Iter.forward (/root/reference/model/imf_vad.py:109-161)
    after the `.to(torch.float)` casts of MMFMIL.forward (:41-42)."""
    if cfg.noise_model not in ("Gaussian", "StudentT"):
        # /root/reference/model/imf_vad.py:137-138
        raise ValueError("Unsupported noise_model. Choose 'Gaussian' or 'StudentT'.")
    W = {k: v.to(dtype) for k, v in sd.items()}
    # the reference casts the inputs to fp32 first (imf_vad.py:41-42); fp64 mode then widens
    x_in = {"image": torch.as_tensor(image_features).to(torch.float32).to(dtype),
            "event": torch.as_tensor(event_features).to(torch.float32).to(dtype)}
    enc = {}
    for m in MODALITIES:
        x = x_in[m]
        for l in range(cfg.num_layers):                                        # :113-116 / :119-122
            p = f"temporal.{m}_attn_layers.{l}."
            a = _self_attention(x, W[p + "in_proj_weight"], W[p + "in_proj_bias"],
                                W[p + "out_proj.weight"], W[p + "out_proj.bias"], cfg.num_heads)
            x = _layer_norm(x + a, W[f"temporal.{m}_norms.{l}.weight"], W[f"temporal.{m}_norms.{l}.bias"])
        enc[m] = _layer_norm(x, W[f"temporal.whiten_{m}.weight"], W[f"temporal.whiten_{m}.bias"])  # :117,:123
    mu, lv = {}, {}
    for m in MODALITIES:                                                        # :125-128
        mu[m] = enc[m] @ W[f"temporal.{m}_mu.weight"].t() + W[f"temporal.{m}_mu.bias"]
        lv[m] = enc[m] @ W[f"temporal.{m}_logvar.weight"].t() + W[f"temporal.{m}_logvar.bias"]
    factor = (cfg.nu + 1) / cfg.nu if cfg.noise_model == "StudentT" else 1.0    # :130-136
    w_i = factor * torch.exp(-lv["image"])
    w_e = factor * torch.exp(-lv["event"])
    denom = w_i + w_e + cfg.epsilon                                             # :140-142
    n_i = w_i / denom
    n_e = w_e / denom
    z = n_i * mu["image"] + n_e * mu["event"]                                   # :144
    for k in range(cfg.num_refinement_steps):                                   # :146-149
        p = f"temporal.refinement_blocks.{k}."
        h = torch.relu(z @ W[p + "0.weight"].t() + W[p + "0.bias"])
        r = h @ W[p + "2.weight"].t() + W[p + "2.bias"]
        z = z - cfg.lambda_ref * r
    logits = z @ W["temporal.classifier.weight"].t() + W["temporal.classifier.bias"]  # :150
    return {"fused": z, "logits": logits, "image_mu": mu["image"], "event_mu": mu["event"],
            "image_logvar": lv["image"], "event_logvar": lv["event"], "w_i": n_i, "w_e": n_e}  # :152-161


class OracleMMFMIL:
    """Callable with the reference model's call signature
    (`model(img, ev, padding_mask, text, lengths)`, /root/reference/test.py:111-117) that runs
    the restated forward on the CPU.  Used by tests as the stand-in for the reference model
    and by bench.py's cpu_baseline leg.  padding_mask/text/lengths are ignored exactly as
    the reference ignores them (imf_vad.py:40-44)."""

    def __init__(self, sd: Dict[str, torch.Tensor], cfg: OracleConfig, dtype=torch.float32):
        self.sd, self.cfg, self.dtype = sd, cfg, dtype

    def to(self, *_a, **_k):
        return self

    def eval(self):
        return self

    def __call__(self, img_visual, ev_visual, padding_mask=None, text=None, lengths=None, return_attn=False):
        with torch.no_grad():
            return forward(self.sd, img_visual.cpu(), ev_visual.cpu(), self.cfg, self.dtype)


# ----------------------------------------------------------------------------------------
# chunker + per-video scoring loop (the callers either side of the path)
# ----------------------------------------------------------------------------------------
def process_split(feat: np.ndarray, length: int = 256) -> Tuple[np.ndarray, int]:
    """/root/reference/data/tools.py:100-114.  len < length -> [length, D] zero padded;
    else [len//length + 1, length, D] with the last chunk zero padded (ALL zero when
    len % length == 0)."""
    n = feat.shape[0]
    if n < length:
        out = np.zeros((length, feat.shape[1]), feat.dtype)
        out[:n] = feat
        return out, n
    nchunk = n // length + 1
    out = np.zeros((nchunk, length, feat.shape[1]), feat.dtype)
    flat = out.reshape(nchunk * length, feat.shape[1])
    flat[:n] = feat
    return out, n


def score_videos(model, videos: Sequence[Tuple[np.ndarray, np.ndarray]], maxlen: int = 256) -> List[np.ndarray]:
    """Per-video score vectors exactly as /root/reference/test.py:76-121 derives `prob1`:
    chunk, unsqueeze when len < maxlen, conditional nan_to_num, forward, flatten,
    slice [:len], sigmoid."""
    out = []
    for img, ev in videos:
        ci, n = process_split(img, maxlen)
        ce, _ = process_split(ev, maxlen)
        ti, te = torch.tensor(ci), torch.tensor(ce)
        if n < maxlen:
            ti, te = ti.unsqueeze(0), te.unsqueeze(0)
        if torch.isnan(ti).any():
            ti = torch.nan_to_num(ti, nan=0.0)
        if torch.isnan(te).any():
            te = torch.nan_to_num(te, nan=0.0)
        o = model(ti, te, None, None, None)
        lg = o["logits"].reshape(-1, 1)
        out.append(torch.sigmoid(lg[0:n].squeeze(-1)).to(torch.float32).cpu().numpy())
    return out


# ----------------------------------------------------------------------------------------
# training-side loss terms, forward only -- checker for csrc/loss.h
# ----------------------------------------------------------------------------------------
def loss_terms(logits, image_mu, event_mu, image_logvar, event_logvar, labels, lengths, noise_model="StudentT", nu=8,
               lambda_reg=1.0, lambda_kl=1.0, dtype=torch.float64) -> Dict[str, float]:
    """The sum the reference's trainers form (/root/reference/train/ucf_train.py:68-101, xd_train.py:60-75), restated with
    explicit algebra: CLAS2 (/root/reference/train/loss.py:18-30: per video the mean of the int(len/16 + 1) largest
    sigmoid(logit[0:len]); BCE against 1 - labels[:, 0] with torch's log clamp at -100), the cosine + norm regulariser of the
    two mu tensors (ucf_train.py:75-82; F.normalize eps 1e-12, F.cosine_similarity eps 1e-8) and the Gaussian / Student-t KL
    terms (:84-98).  Pinned by tests/golden/loss_terms.npz (reference CLAS2 + the trainers' torch calls)."""
    lg = torch.as_tensor(logits).to(dtype)
    B = lg.shape[0]
    p = torch.sigmoid(lg.reshape(B, -1))
    y = 1.0 - torch.as_tensor(labels).to(dtype)[:, 0]
    inst = []
    for i in range(B):
        n = int(lengths[i])
        k = int(n / 16 + 1)
        v = torch.sort(p[i, :n], descending=True).values[:k]
        inst.append(v.sum() / k)
    inst = torch.stack(inst)
    bce = -(y * torch.clamp(torch.log(inst), min=-100.0) + (1 - y) * torch.clamp(torch.log(1 - inst), min=-100.0))
    cls = float(bce.mean())
    mi, me = torch.as_tensor(image_mu).to(dtype), torch.as_tensor(event_mu).to(dtype)
    ni, ne = torch.sqrt((mi * mi).sum(-1)), torch.sqrt((me * me).sum(-1))
    ui = mi / torch.clamp(ni, min=1e-12).unsqueeze(-1)
    ue = me / torch.clamp(ne, min=1e-12).unsqueeze(-1)
    qi = torch.clamp(torch.sqrt((ui * ui).sum(-1)), min=1e-8)
    qe = torch.clamp(torch.sqrt((ue * ue).sum(-1)), min=1e-8)
    cos = ((ui / qi.unsqueeze(-1)) * (ue / qe.unsqueeze(-1))).sum(-1)
    lcos, lnorm = float((1 - cos).mean()), float(torch.abs(ni - ne).mean())
    sh = math.log(nu / (nu + 1)) if noise_model == "StudentT" else 0.0
    kl = []
    for mu, lv in ((mi, torch.as_tensor(image_logvar).to(dtype)), (me, torch.as_tensor(event_logvar).to(dtype))):
        e = lv + sh
        kl.append(float(-0.5 * (1 + e - mu * mu - torch.exp(e)).mean()))
    reg, klsum = lcos + lnorm, kl[0] + kl[1]
    return {"classification": cls, "reg": reg, "cos": lcos, "norm": lnorm, "kl": klsum, "kl_image": kl[0], "kl_event": kl[1],
            "total": cls + lambda_reg * reg + lambda_kl * klsum}
