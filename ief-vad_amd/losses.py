"""Training-side loss head: device counterparts of the reference's loss terms with the reference's call
shapes, computed by libiefvad (`iefvad_loss_forward` / `iefvad_loss_backward`, csrc/loss.h) -- no torch op computes anything here.

  * `CLAS2(logits, labels, lengths, device)`   -- /root/reference/train/loss.py:18-30
  * `training_losses(outputs, labels, lengths, ...)` -- the sum the trainers form, /root/reference/train/ucf_train.py:68-101
    (lambda_reg = lambda_kl = 1) and train/xd_train.py:60-75 (0.01, 0.01): classification + lambda_reg * (cosine + norm
    regulariser of image_mu / event_mu) + lambda_kl * (Gaussian or Student-t KL of both modalities).

  * `training_loss(outputs, labels, lengths, ...)` -- the same total as ONE differentiable scalar: `.backward()` fills the
    `.grad` of `logits`, `image_mu`, `event_mu`, `image_logvar`, `event_logvar` (the gradients `loss.backward()` hands to the
    model's outputs in ucf_train.py:103), computed by `iefvad_loss_backward`.

  * `AdamW(params, lr)` -- torch.optim.AdamW's update as the trainers construct it (ucf_train.py:28), `iefvad_adamw_step`.

The model's own backward pass and the train-mode forward (attention dropout, imf_vad.py:70) are not part of this build: the
gradients stop at the tensors `iefvad_amd.MMFMIL` returns.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from . import lib as _lib

TERMS = ("classification", "reg", "cos", "norm", "kl", "kl_image", "kl_event", "total")


def _run(logits, heads, labels, lengths, noise_model, nu, lambda_reg, lambda_kl) -> torch.Tensor:
    if not logits.is_cuda:
        raise RuntimeError("iefvad_amd.losses runs on a HIP device only; there is no CPU fallback")
    dev = logits.device
    B = int(logits.shape[0])
    lg = logits.reshape(B, -1).float().contiguous()
    T = int(lg.shape[1])
    targets = (1 - labels[:, 0].reshape(B)).to(device=dev, dtype=torch.float32).contiguous()          # loss.py:20
    lens = torch.as_tensor(lengths).reshape(B).to(device=dev, dtype=torch.int32).contiguous()
    ptrs = [None] * 4
    keep = []
    if heads is not None:
        for i, t in enumerate(heads):
            t = t.reshape(B * T, -1).float().contiguous()
            if t.shape[1] != 768 or t.device != dev:
                raise ValueError("image_mu / event_mu / image_logvar / event_logvar must be [B, T, 768] tensors on the logits' device")
            keep.append(t)
            ptrs[i] = C.c_void_p(t.data_ptr())
    if noise_model not in ("Gaussian", "StudentT"):
        raise ValueError("Unsupported noise_model. Choose 'Gaussian' or 'StudentT'.")
    lib = _lib.load_library()
    need = lib.iefvad_loss_workspace_bytes(B, T)
    ws = torch.empty(max(need, 1), dtype=torch.uint8, device=dev)
    out = torch.empty(8, dtype=torch.float32, device=dev)
    with torch.cuda.device(dev):
        rc = lib.iefvad_loss_forward(C.c_void_p(lg.data_ptr()), ptrs[0], ptrs[1], ptrs[2], ptrs[3], C.c_void_p(lens.data_ptr()),
                                     C.c_void_p(targets.data_ptr()), B, T,
                                     _lib.NOISE_STUDENT_T if noise_model == "StudentT" else _lib.NOISE_GAUSSIAN, float(nu),
                                     float(lambda_reg), float(lambda_kl), C.c_void_p(out.data_ptr()), C.c_void_p(ws.data_ptr()),
                                     ws.numel(), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError("iefvad_loss_forward: " + _lib.last_error())
    return out


def CLAS2(logits: torch.Tensor, labels: torch.Tensor, lengths, device=None) -> torch.Tensor:
    """Top-k MIL binary cross entropy, same arguments as the reference's CLAS2 (loss.py:18): `logits` [B, T, 1] (or [B, T]),
    `labels` [B, C] one-hot with column 0 = normal, `lengths` [B].  Returns a 0-dim device tensor."""
    return _run(logits, None, labels, lengths, "Gaussian", 1.0, 0.0, 0.0)[0]


def training_losses(outputs: Dict[str, torch.Tensor], labels: torch.Tensor, lengths, noise_model: str = "StudentT", nu: float = 8,
                    lambda_reg: float = 1.0, lambda_kl: float = 1.0) -> Dict[str, torch.Tensor]:
    """Every term of the trainers' loss from the model's output dict (`logits`, `image_mu`, `event_mu`, `image_logvar`,
    `event_logvar`): a dict of 0-dim device tensors keyed by `TERMS`."""
    heads = (outputs["image_mu"], outputs["event_mu"], outputs["image_logvar"], outputs["event_logvar"])
    out = _run(outputs["logits"], heads, labels, lengths, noise_model, nu, lambda_reg, lambda_kl)
    return {k: out[i] for i, k in enumerate(TERMS)}


def _prep(logits, heads, labels, lengths):
    dev = logits.device
    B = int(logits.shape[0])
    lg = logits.detach().reshape(B, -1).float().contiguous()
    T = int(lg.shape[1])
    targets = (1 - labels[:, 0].reshape(B)).to(device=dev, dtype=torch.float32).contiguous()
    lens = torch.as_tensor(lengths).reshape(B).to(device=dev, dtype=torch.int32).contiguous()
    hs = [t.detach().reshape(B * T, -1).float().contiguous() for t in heads]
    return B, T, lg, targets, lens, hs


class _LossHead(torch.autograd.Function):
    """total = classification + lambda_reg * reg + lambda_kl * kl as one differentiable node (both passes in libiefvad)."""

    @staticmethod
    def forward(ctx, logits, image_mu, event_mu, image_logvar, event_logvar, labels, lengths, noise_model, nu, lambda_reg, lambda_kl):
        heads = (image_mu, event_mu, image_logvar, event_logvar)
        out = _run(logits.detach(), tuple(t.detach() for t in heads), labels, lengths, noise_model, nu, lambda_reg, lambda_kl)
        ctx.save_for_backward(logits, image_mu, event_mu, image_logvar, event_logvar, labels)
        ctx.meta = (lengths, noise_model, float(nu), float(lambda_reg), float(lambda_kl))
        return out[7].clone()

    @staticmethod
    def backward(ctx, grad_out):
        logits, image_mu, event_mu, image_logvar, event_logvar, labels = ctx.saved_tensors
        lengths, noise_model, nu, lambda_reg, lambda_kl = ctx.meta
        heads = (image_mu, event_mu, image_logvar, event_logvar)
        B, T, lg, targets, lens, hs = _prep(logits, heads, labels, lengths)
        dev = logits.device
        need = list(ctx.needs_input_grad[:5])
        grads = [torch.empty_like(lg) if need[0] else None] + [torch.empty_like(hs[i]) if need[1 + i] else None for i in range(4)]
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        lib = _lib.load_library()
        with torch.cuda.device(dev):
            rc = lib.iefvad_loss_backward(ptr(lg), ptr(hs[0]), ptr(hs[1]), ptr(hs[2]), ptr(hs[3]), ptr(lens), ptr(targets), B, T,
                                          _lib.NOISE_STUDENT_T if noise_model == "StudentT" else _lib.NOISE_GAUSSIAN, nu, lambda_reg,
                                          lambda_kl, float(grad_out), ptr(grads[0]), ptr(grads[1]), ptr(grads[2]), ptr(grads[3]),
                                          ptr(grads[4]), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
        if rc != 0:
            raise RuntimeError("iefvad_loss_backward: " + _lib.last_error())
        shaped = [g.reshape(x.shape).to(x.dtype) if g is not None else None
                  for g, x in zip(grads, (logits, image_mu, event_mu, image_logvar, event_logvar))]
        return (*shaped, None, None, None, None, None, None)


def training_loss(outputs: Dict[str, torch.Tensor], labels: torch.Tensor, lengths, noise_model: str = "StudentT", nu: float = 8,
                  lambda_reg: float = 1.0, lambda_kl: float = 1.0) -> torch.Tensor:
    """The trainers' total loss (ucf_train.py:100-102) as a differentiable 0-dim device tensor; see the module docstring."""
    if noise_model not in ("Gaussian", "StudentT"):
        raise ValueError("Unsupported noise_model. Choose 'Gaussian' or 'StudentT'.")
    if not outputs["logits"].is_cuda:
        raise RuntimeError("iefvad_amd.losses runs on a HIP device only; there is no CPU fallback")
    return _LossHead.apply(outputs["logits"], outputs["image_mu"], outputs["event_mu"], outputs["image_logvar"], outputs["event_logvar"],
                           labels, lengths, noise_model, nu, lambda_reg, lambda_kl)


class AdamW:
    """torch.optim.AdamW's update (the optimiser of /root/reference/train/ucf_train.py:28, xd_train.py:25) on device tensors,
    one `iefvad_adamw_step` launch per tensor: `AdamW(params, lr)`, `.step()` reads each parameter's `.grad`, `.zero_grad()`.
    State (`exp_avg`, `exp_avg_sq`, the per-parameter step counts) lives here; amsgrad, maximize and parameter groups are not offered."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        self.params = [p for p in params]
        if any((not p.is_cuda) or p.dtype != torch.float32 or not p.is_contiguous() for p in self.params):
            raise ValueError("AdamW: contiguous fp32 tensors on a HIP device only; there is no CPU fallback")
        self.lr, self.betas, self.eps, self.weight_decay = float(lr), (float(betas[0]), float(betas[1])), float(eps), float(weight_decay)
        self.state = [(torch.zeros_like(p), torch.zeros_like(p)) for p in self.params]
        self.steps = [0] * len(self.params)          # per parameter, as torch counts them: only steps that saw a gradient

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    @torch.no_grad()
    def step(self):
        lib = _lib.load_library()
        for i, (p, (m, v)) in enumerate(zip(self.params, self.state)):
            if p.grad is None:
                continue
            self.steps[i] += 1
            g = p.grad.contiguous().float()
            with torch.cuda.device(p.device):
                rc = lib.iefvad_adamw_step(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(m.data_ptr()),
                                           C.c_void_p(v.data_ptr()), p.numel(), self.lr, self.betas[0], self.betas[1], self.eps,
                                           self.weight_decay, self.steps[i], C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
            if rc != 0:
                raise RuntimeError("iefvad_adamw_step: " + _lib.last_error())
