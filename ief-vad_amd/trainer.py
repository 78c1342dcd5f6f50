"""The reference's training loops over the MI355X path: counterparts of
/root/reference/train/ucf_train.py:16-156 (paired normal / abnormal batches: UCF-Crime, ShanghaiTech, MSAD) and
/root/reference/train/xd_train.py:14-129 (one loader: XD-Violence), with the same call signatures minus wandb.

One step = what ucf_train.py:43-106 does: `model.train()`, the conditional `nan_to_num` of the batch (:50-53), the forward
(`iefvad_train_forward`), CLAS2 + cosine / norm regulariser + Gaussian / Student-t KL (`iefvad_loss_forward`), `zero_grad`,
`loss.backward()` (`iefvad_loss_backward` then `iefvad_train_backward`), `optimizer.step()` (`iefvad_adamw_step`).  Around it the
loops keep the reference's bookkeeping: evaluation every `print_steps` samples through `harness.ucf_test` / `harness.xd_test`, the best checkpoint
(`{'epoch', 'model_state_dict', 'optimizer_state_dict', 'ap'}`, :141-149), `scheduler.step()` and the reload of the best
checkpoint at every epoch end (:151-153), the final rewrite as a bare state_dict (:155-156).  Logging goes to a callback instead
of wandb.  No torch op computes a loss or a gradient here.
"""
from __future__ import annotations

import os
from typing import Callable, Dict, Iterable, Optional, Sequence

import numpy as np
import torch
from torch.optim.lr_scheduler import MultiStepLR

from . import harness, losses


def get_prompt_text(label_map: dict) -> list:
    """train/utils.py:53-58: the label map's values, in order."""
    return list(label_map.values())


def get_batch_label(texts: Sequence[str], prompt_text: Sequence[str], label_map: dict) -> torch.Tensor:
    """One-hot (multi-hot for XD's 'a-b' labels) class vectors, the four cases of train/utils.py:5-50 keyed on the size of the label
    map exactly as the reference keys them: 17 entries (ShanghaiTech: 'normal' -> column 0, anything else column 1, two columns),
    2 entries (MSAD: 'Normal' -> 0), 7 entries (XD-Violence: every '-'-separated part that is in the map), otherwise (UCF-Crime:
    the column of the label's prompt text)."""
    n = len(label_map)
    if n == 17 or n == 2:
        normal = 'normal' if n == 17 else 'Normal'
        out = torch.zeros(len(texts), 2)
        for i, t in enumerate(texts):
            out[i, 0 if t == normal else 1] = 1
        return out
    out = torch.zeros(len(texts), len(prompt_text))
    for i, t in enumerate(texts):
        for part in (t.split('-') if n == 7 else [t]):
            if part in label_map:
                out[i, list(prompt_text).index(label_map[part])] = 1
    return out


def _nan_rule(x: torch.Tensor) -> torch.Tensor:
    return torch.nan_to_num(x, nan=0.0) if bool(torch.isnan(x).any()) else x          # ucf_train.py:50-53


_NAN_FLAGS = {}


def _nan_rule_pair(img: torch.Tensor, ev: torch.Tensor):
    """ucf_train.py:50-53 for both inputs of a step.  fp32 device tensors go through `iefvad_nan_rule`: one scan + one repair launch,
    the flag never read by the host (torch's form costs an isnan pass, a reduction and a device-to-host wait per tensor and step);
    a tensor that does hold a NaN is repaired IN PLACE (the loops hand over per-step temporaries).  Anything else: the torch form."""
    ok = all(t.is_cuda and t.dtype == torch.float32 and t.is_contiguous() and t.numel() % 4 == 0 and t.data_ptr() % 16 == 0 for t in (img, ev))
    if not ok or img.numel() != ev.numel() or img.device != ev.device:
        return _nan_rule(img), _nan_rule(ev)
    import ctypes as C
    from . import lib as _lib
    dev = img.device
    flags = _NAN_FLAGS.get(dev)
    if flags is None:
        flags = _NAN_FLAGS[dev] = torch.zeros(2, dtype=torch.int32, device=dev)
    with torch.cuda.device(dev):
        rc = _lib.load_library().iefvad_nan_rule(C.c_void_p(img.data_ptr()), C.c_void_p(ev.data_ptr()), img.numel(), C.c_void_p(flags.data_ptr()),
                                                 C.c_void_p(torch.cuda.current_stream(dev).cuda_stream))
    if rc != 0:
        raise RuntimeError("iefvad_nan_rule: " + _lib.last_error())
    return img, ev


def train_step(model, optimizer, img: torch.Tensor, ev: torch.Tensor, labels: torch.Tensor, lengths: torch.Tensor,
               noise_model: str = "StudentT", lambda_reg: float = 1.0, lambda_kl: float = 1.0, nan_to_num: bool = True,
               want_terms: bool = True) -> Optional[Dict[str, torch.Tensor]]:
    """One optimiser step on a device batch; returns the eight loss terms (`losses.TERMS`) as 0-dim device tensors -- nothing
    is read back to the host -- or None with `want_terms=False` (the loops ask for them only on the steps they log:
    ucf_train.py:108-128 prints every `print_steps` samples)."""
    model.train()
    if nan_to_num:
        img, ev = _nan_rule_pair(img, ev)
    out = model(img, ev, None, None, lengths)
    nu = model.temporal.nu                                                             # ucf_train.py:94-95 reads it there
    total = losses.training_loss(out, labels, lengths, noise_model, nu, lambda_reg, lambda_kl)
    optimizer.zero_grad()
    total.backward()
    optimizer.step()
    if not want_terms:
        return None
    with torch.no_grad():
        terms = losses.training_losses({k: v.detach() for k, v in out.items()}, labels, lengths, noise_model, nu, lambda_reg, lambda_kl)
    return terms


def _save_best(path, epoch, model, optimizer, metric):
    os.makedirs(os.path.dirname(path) or ".", exist_ok=True)
    torch.save({'epoch': epoch, 'model_state_dict': model.state_dict(), 'optimizer_state_dict': optimizer.state_dict(), 'ap': metric}, path)


def _epoch_end(path, model, scheduler):
    scheduler.step()
    if os.path.exists(path):                     # the reference reloads the best weights after every epoch (ucf_train.py:151-153)
        ck = torch.load(path, weights_only=True)
        model.load_state_dict(ck['model_state_dict'])


def _finish(path):
    if os.path.exists(path):                     # ucf_train.py:155-156: the file ends up holding the bare state_dict
        ck = torch.load(path, weights_only=True)
        if 'model_state_dict' in ck:
            torch.save(ck['model_state_dict'], path)


def train_paired(args, model, normal_loader, abnormal_loader, test_loader, label_map, device, gt: Optional[np.ndarray] = None,
                 log: Optional[Callable[[dict], None]] = None, optimizer=None, eval_batch_chunks: int = 64):
    """Counterpart of /root/reference/train/ucf_train.py:train (same positional arguments; `gt` defaults to np.load(args.gt_path)).
    Each step concatenates a normal and an abnormal batch (:44-48), lambda_reg = lambda_kl = 1 (:100-101); the model is evaluated
    every `args.print_steps` samples and kept when its AUC improves.  Returns the best AUC."""
    model.to(device)
    if gt is None:
        gt = np.load(args.gt_path)
    optimizer = optimizer or losses.AdamW(model.parameters(), lr=args.lr)
    scheduler = MultiStepLR(optimizer, args.scheduler_milestones, args.scheduler_rate)
    prompt_text = get_prompt_text(label_map)
    path = os.path.join('checkpoints', f'{args.exp_name}.pth')
    best = 0.0
    for e in range(args.max_epoch):
        n_it, a_it = iter(normal_loader), iter(abnormal_loader)
        for i in range(min(len(normal_loader), len(abnormal_loader))):
            n_img, n_ev, n_lab, n_len = next(n_it)
            a_img, a_ev, a_lab, a_len = next(a_it)
            img = torch.cat([n_img, a_img], dim=0).to(device)
            ev = torch.cat([n_ev, a_ev], dim=0).to(device)
            lengths = torch.cat([n_len, a_len], dim=0).to(device)
            labels = get_batch_label(list(n_lab) + list(a_lab), prompt_text, label_map).to(device)
            step = i * normal_loader.batch_size * 2                                    # ucf_train.py:42,106
            due = step % args.print_steps == 0 and step != 0
            terms = train_step(model, optimizer, img, ev, labels, lengths, args.noise_model, 1.0, 1.0, want_terms=due)
            if due:
                rec = {f'train/loss_{k}' if k != 'total' else 'train/loss': float(v) for k, v in terms.items()}
                auc, ap = harness.ucf_test(args, model, test_loader, args.visual_length, prompt_text, gt, device,   # ucf_train.py:130-139
                                           vis=False, batch_chunks=eval_batch_chunks)
                rec.update(epoch=e, step=step, auc=auc, ap=ap)
                if log:
                    log(rec)
                if auc > best:
                    best = auc
                    _save_best(path, e, model, optimizer, best)
        _epoch_end(path, model, scheduler)
    _finish(path)
    return best


def train_single(args, model, train_loader, test_loader, label_map, device, gt: Optional[np.ndarray] = None,
                 log: Optional[Callable[[dict], None]] = None, optimizer=None, eval_batch_chunks: int = 64):
    """Counterpart of /root/reference/train/xd_train.py:train: one loader, lambda_reg = lambda_kl = 0.01 (:73-74), the Student-t
    shift of the KL terms whatever `args.noise_model` says (:67-70 apply it unconditionally), best checkpoint by AP (:114)."""
    model.to(device)
    if gt is None:
        gt = np.load(args.gt_path)
    optimizer = optimizer or losses.AdamW(model.parameters(), lr=args.lr)
    scheduler = MultiStepLR(optimizer, args.scheduler_milestones, args.scheduler_rate)
    prompt_text = get_prompt_text(label_map)
    path = os.path.join('checkpoints', f'{args.exp_name}.pth')
    best = 0.0
    for e in range(args.max_epoch):
        for i, (img, ev, text_labels, lengths) in enumerate(train_loader):
            labels = get_batch_label(list(text_labels), prompt_text, label_map).to(device)
            step = i * train_loader.batch_size
            due = step % args.print_steps == 0 and step != 0
            terms = train_step(model, optimizer, img.to(device), ev.to(device), labels, lengths.to(device), "StudentT", 0.01, 0.01,
                               nan_to_num=False, want_terms=due)                        # xd_train.py has no NaN rule
            if due:
                rec = {f'train/loss_{k}' if k != 'total' else 'train/loss': float(v) for k, v in terms.items()}
                auc, ap = harness.xd_test(args, model, test_loader, args.visual_length, prompt_text, gt, device, label_map,   # xd_train.py:102-112
                                          vis=False, batch_chunks=eval_batch_chunks)
                rec.update(epoch=e, step=step, auc=auc, ap=ap)
                if log:
                    log(rec)
                if ap > best:
                    best = ap
                    _save_best(path, e, model, optimizer, best)
        _epoch_end(path, model, scheduler)
    _finish(path)
    return best
