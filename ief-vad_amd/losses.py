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

In train mode `iefvad_amd.MMFMIL` returns differentiable tensors (model.py: `_TrainForward`), so `training_loss(...).backward()`
reaches every parameter: loss head backward here, the model's backward pass in `iefvad_train_backward`.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import numpy as np
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
        # the upstream gradient stays on the device: the kernels read the scalar themselves (no .item() sync per step)
        gscale = grad_out.detach().reshape(1).to(device=dev, dtype=torch.float32).contiguous()
        lib = _lib.load_library()
        with torch.cuda.device(dev):
            rc = lib.iefvad_loss_backward(ptr(lg), ptr(hs[0]), ptr(hs[1]), ptr(hs[2]), ptr(hs[3]), ptr(lens), ptr(targets), B, T,
                                          _lib.NOISE_STUDENT_T if noise_model == "StudentT" else _lib.NOISE_GAUSSIAN, nu, lambda_reg,
                                          lambda_kl, 1.0, ptr(grads[0]), ptr(grads[1]), ptr(grads[2]), ptr(grads[3]),
                                          ptr(grads[4]), ptr(gscale), C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
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


class AdamW(torch.optim.Optimizer):
    """torch.optim.AdamW's update (the optimiser of /root/reference/train/ucf_train.py:28, xd_train.py:25) with the arithmetic in
    libiefvad: ONE `iefvad_adamw_step_multi` launch for all parameter tensors of a group (`iefvad_adamw_step` per tensor when their step
    counts or devices differ), torch's operation order, hyper-parameter scalars formed in
    double on the host as torch forms them.  A `torch.optim.Optimizer` subclass, so what the trainers do with their optimiser
    works unchanged: `MultiStepLR(optimizer, ...)` (ucf_train.py:29-33), `optimizer.state_dict()` in the checkpoint
    (ucf_train.py:141-149), `load_state_dict`, parameter groups, `zero_grad()`.  State per parameter as torch.optim.AdamW keeps
    it (`step` as a CPU float tensor, `exp_avg`, `exp_avg_sq`), so a state_dict moves between the two optimisers.
    amsgrad / maximize / capturable / fused are not offered."""

    def __init__(self, params, lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8, weight_decay: float = 1e-2):
        if lr < 0 or eps < 0 or not (0 <= betas[0] < 1) or not (0 <= betas[1] < 1) or weight_decay < 0:
            raise ValueError("AdamW: invalid hyper-parameter")
        defaults = dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False, foreach=None,
                        capturable=False, differentiable=False, fused=None)
        super().__init__(params, defaults)
        for group in self.param_groups:
            for p in group["params"]:
                if (not p.is_cuda) or p.dtype != torch.float32 or not p.is_contiguous():
                    raise ValueError("AdamW: contiguous fp32 tensors on a HIP device only; there is no CPU fallback")

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        lib = _lib.load_library()
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("iefvad_amd.losses.AdamW: amsgrad / maximize are not offered")
            b1, b2 = group["betas"]
            todo = []
            for p in group["params"]:
                if p.grad is None:
                    continue
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)           # torch keeps it on the CPU for the default path
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                g = p.grad
                if g.dtype != torch.float32 or not g.is_contiguous():
                    g = g.contiguous().float()
                todo.append((p, g, st))
            if not todo:
                continue
            # one launch for all tensors that share a device and a step count (in the trainers: all of them); stragglers one by one
            steps = {int(st["step"].item()) for _, _, st in todo}
            devs = {p.device for p, _, _ in todo}
            if len(steps) == 1 and len(devs) == 1 and len(todo) > 1:
                dev, step = next(iter(devs)), next(iter(steps))
                tab = np.zeros((len(todo), 6), dtype=np.uint64)
                chunk = 0
                for i, (p, g, st) in enumerate(todo):
                    tab[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel(), chunk)
                    chunk += (p.numel() + 4095) // 4096
                host = self.__dict__.setdefault("_table_host", {}).get(dev)
                if host is None or host.shape[0] < len(todo):
                    host = torch.empty(max(len(todo), 128), 6, dtype=torch.int64).pin_memory()
                    self._table_host[dev] = host
                    self.__dict__.setdefault("_table_dev", {})[dev] = torch.empty_like(host, device=dev)
                tdev = self._table_dev[dev]
                with torch.cuda.device(dev):
                    # the pinned table is rewritten every step: wait until the previous step's copy has left it
                    ev = self.__dict__.setdefault("_table_event", {}).get(dev)
                    if ev is not None:
                        ev.synchronize()
                    host[:len(todo)].copy_(torch.from_numpy(tab.view(np.int64)))
                    tdev[:len(todo)].copy_(host[:len(todo)], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record()
                    self._table_event[dev] = ev
                    rc = lib.iefvad_adamw_step_multi(C.c_void_p(tdev.data_ptr()), len(todo), chunk, float(group["lr"]), float(b1), float(b2),
                                                     float(group["eps"]), float(group["weight_decay"]), step,
                                                     C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
                if rc != 0:
                    raise RuntimeError("iefvad_adamw_step_multi: " + _lib.last_error())
                self._keep = [g for _, g, _ in todo]          # converted gradients stay alive until the next step's launch has them
                for p, _, _ in todo:
                    torch.autograd.graph.increment_version(p)  # the kernel wrote through raw pointers: autograd and the model's weight cache must notice
                continue
            for p, g, st in todo:
                with torch.cuda.device(p.device):
                    rc = lib.iefvad_adamw_step(C.c_void_p(p.data_ptr()), C.c_void_p(g.data_ptr()), C.c_void_p(st["exp_avg"].data_ptr()),
                                               C.c_void_p(st["exp_avg_sq"].data_ptr()), p.numel(), float(group["lr"]), float(b1), float(b2),
                                               float(group["eps"]), float(group["weight_decay"]), int(st["step"].item()),
                                               C.c_void_p(torch.cuda.current_stream(p.device).cuda_stream))
                if rc != 0:
                    raise RuntimeError("iefvad_adamw_step: " + _lib.last_error())
                # the kernel wrote through the raw pointer: tell autograd (and iefvad_amd.MMFMIL's weight cache) that p changed
                torch.autograd.graph.increment_version(p)
        return loss
