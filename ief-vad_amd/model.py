"""Drop-in `MMFMIL` for the reference's evaluation loops, backed by libiefvad.so.

Mirrors the interface of /root/reference/model/imf_vad.py:
  * constructor signature of `MMFMIL` (:6-18) and the hyper-parameters it forwards (:30-38);
  * `state_dict()` keys and shapes so `load_state_dict(torch.load(ckpt))`
    (/root/reference/test.py:377-378) works unchanged;
  * `forward(img, ev, padding_mask, text, lengths, return_attn=False) -> dict` with the eight keys of
    :152-161; `padding_mask`, `text`, `lengths`, `return_attn` are accepted and ignored (:40-44);
  * attribute `model.temporal.nu` read by the training loss (/root/reference/train/ucf_train.py:94-95);
  * `ValueError` for an unsupported `noise_model`, raised at forward time (:137-138).

The modules below only HOLD parameters (same names, shapes and default initialisation as the
torch.nn modules the reference instantiates); no torch op computes the forward.  Inputs must live
on a HIP device: there is no CPU path, a CPU tensor raises.

`model.train()` switches `forward` to the train-mode path (attention dropout, activations kept for the backward): the returned
tensors are differentiable with respect to every parameter, `loss.backward()` runs `iefvad_train_backward`.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, Optional

import torch
import torch.nn as nn

from . import lib as _lib

_IN_DTYPES = {torch.float32: _lib.IN_F32, torch.float16: _lib.IN_F16, torch.bfloat16: _lib.IN_BF16}
OUTPUT_KEYS = ("fused", "logits", "image_mu", "event_mu", "image_logvar", "event_logvar", "w_i", "w_e")


class _LinearParams(nn.Module):
    """Parameter holder with nn.Linear's names, shapes and default init (kaiming-uniform a=sqrt(5))."""

    def __init__(self, in_features: int, out_features: int):
        super().__init__()
        self.weight = nn.Parameter(torch.empty(out_features, in_features))
        self.bias = nn.Parameter(torch.empty(out_features))
        nn.init.kaiming_uniform_(self.weight, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(in_features)
        nn.init.uniform_(self.bias, -bound, bound)


class _LayerNormParams(nn.Module):
    def __init__(self, dim: int):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(dim))
        self.bias = nn.Parameter(torch.zeros(dim))


class _AttentionParams(nn.Module):
    """Parameter holder with nn.MultiheadAttention's packed layout: in_proj_weight [3D, D] in (q, k, v)
    order, in_proj_bias [3D], out_proj.{weight,bias}; same init order as torch (out_proj first, then
    xavier-uniform in_proj, zero biases) so a seeded construction reproduces the reference's weights."""

    def __init__(self, dim: int, dropout: float = 0.0):
        super().__init__()
        self.dropout = dropout          # nn.MultiheadAttention.dropout: applied to the attention weights in train mode
        self.in_proj_weight = nn.Parameter(torch.empty(3 * dim, dim))
        self.in_proj_bias = nn.Parameter(torch.empty(3 * dim))
        self.out_proj = _LinearParams(dim, dim)
        nn.init.xavier_uniform_(self.in_proj_weight)
        nn.init.constant_(self.in_proj_bias, 0.0)
        nn.init.constant_(self.out_proj.bias, 0.0)


class _RefinementBlockParams(nn.Module):
    """Sequential(Linear, ReLU, Linear): parameters live at indices 0 and 2 (imf_vad.py:97-104)."""

    def __init__(self, dim: int):
        super().__init__()
        self.add_module("0", _LinearParams(dim, dim))
        self.add_module("2", _LinearParams(dim, dim))


class FusionParams(nn.Module):
    """Parameters and hyper-parameters of MultiModal_Fusion_Attn_Iter (imf_vad.py:48-107), registered
    in the reference's order."""

    def __init__(self, embed_dim, num_layers=2, num_heads=8, dropout=0.1, num_refinement_steps=3,
                 lambda_ref=0.5, noise_model="StudentT", nu=5, epsilon=1e-8):
        super().__init__()
        self.embed_dim = embed_dim
        self.num_layers = num_layers
        self.num_heads = num_heads
        self.num_refinement_steps = num_refinement_steps
        self.lambda_ref = lambda_ref
        self.noise_model = noise_model
        self.nu = nu
        self.epsilon = epsilon
        self.image_attn_layers = nn.ModuleList([_AttentionParams(embed_dim, dropout) for _ in range(num_layers)])
        self.image_norms = nn.ModuleList([_LayerNormParams(embed_dim) for _ in range(num_layers)])
        self.event_attn_layers = nn.ModuleList([_AttentionParams(embed_dim, dropout) for _ in range(num_layers)])
        self.event_norms = nn.ModuleList([_LayerNormParams(embed_dim) for _ in range(num_layers)])
        self.whiten_image = _LayerNormParams(embed_dim)
        self.whiten_event = _LayerNormParams(embed_dim)
        self.image_mu = _LinearParams(embed_dim, embed_dim)
        self.event_mu = _LinearParams(embed_dim, embed_dim)
        self.image_logvar = _LinearParams(embed_dim, embed_dim)
        self.event_logvar = _LinearParams(embed_dim, embed_dim)
        if num_refinement_steps == 0:
            self.refinement_blocks = nn.ModuleList([nn.Identity()])
        else:
            self.refinement_blocks = nn.ModuleList(
                [_RefinementBlockParams(embed_dim) for _ in range(num_refinement_steps)])
        self.classifier = _LinearParams(embed_dim, 1)


class _TrainForward(torch.autograd.Function):
    """The train-mode forward of the whole model as ONE autograd node: forward = `iefvad_train_forward` (keeps the activations in
    a buffer this node owns), backward = `iefvad_train_backward` (every parameter gradient).  The parameters are inputs of the
    node, so `loss.backward()` accumulates into their `.grad` exactly as with the reference's modules."""

    @staticmethod
    def forward(ctx, model, img, ev, *params):
        lib = _lib.load_library()
        t = model.temporal
        device = img.device
        B, T, D = img.shape
        img = model._prepare_input(img)
        ev = model._prepare_input(ev)
        if ev.dtype != img.dtype:
            img, ev = img.to(torch.float), ev.to(torch.float)
        opt = _lib.TrainOptions()
        for m, name in enumerate(("image", "event")):
            for l, a in enumerate(getattr(t, f"{name}_attn_layers")):
                opt.dropout_p[m][l] = float(a.dropout)
        seed = model.__dict__.get("dropout_seed")
        if seed is None:      # a fresh stream per step, reproducible under torch.manual_seed
            seed = int(torch.empty((), dtype=torch.int64).random_().item())
        opt.seed = seed & 0xFFFFFFFFFFFFFFFF
        mask = model.__dict__.get("dropout_mask")
        if mask is not None:
            want = (2, t.num_layers, B, t.num_heads, T, T)
            if tuple(mask.shape) != want or mask.dtype != torch.uint8 or mask.device != device or not mask.is_contiguous():
                raise ValueError(f"dropout_mask must be a contiguous uint8 tensor of shape {want} on {device}")
            opt.keep_mask = mask.data_ptr()
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            model._ensure_handle(device)
            model._ensure_weights(device, stream)
            need = lib.iefvad_train_workspace_bytes(model._handle, B)
            if need == 0:
                raise RuntimeError(f"iefvad_train_workspace_bytes: unsupported batch size {B}")
            ws = torch.empty(need, dtype=torch.uint8, device=device)
            f32 = dict(dtype=torch.float32, device=device)
            res = {k: torch.empty(B, T, 1 if k == "logits" else D, **f32) for k in OUTPUT_KEYS}
            o = _lib.Outputs()
            for k in OUTPUT_KEYS:
                setattr(o, k, res[k].data_ptr())
            rc = lib.iefvad_train_forward(model._handle, C.c_void_p(img.data_ptr()), C.c_void_p(ev.data_ptr()), _IN_DTYPES[img.dtype], B,
                                          C.byref(opt), C.c_void_p(ws.data_ptr()), ws.numel(), C.byref(o), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("iefvad_train_forward: " + _lib.last_error())
        ctx.model, ctx.ws, ctx.B, ctx.mask = model, ws, B, mask
        ctx.params = params
        ctx.set_materialize_grads(False)     # backward takes None as "no gradient": no zero-filled [B,T,768] tensors for unused outputs
        outs = tuple(res[k] for k in OUTPUT_KEYS)
        # the library wrote five of the outputs in place and reads them, and fp32 inputs, again in the backward (include/iefvad.h):
        # saved here so that they stay alive and an in-place edit between the two passes raises as it would in the reference
        ctx.save_for_backward(img, ev, *(res[k] for k in ("fused", "image_mu", "event_mu", "image_logvar", "event_logvar")))
        return outs

    @staticmethod
    def backward(ctx, *gouts):
        lib = _lib.load_library()
        model, ws = ctx.model, ctx.ws
        if ws is None:
            raise RuntimeError("iefvad_amd.MMFMIL: backward through the same forward twice (the activations were released)")
        _ = ctx.saved_tensors               # version check of the tensors the library reads again
        device = ws.device
        keep = []
        dout = _lib.OutputGrads()
        for k, g in zip(OUTPUT_KEYS, gouts):
            if g is not None:
                g = g.contiguous().float()
                keep.append(g)
                setattr(dout, k, g.data_ptr())
        needs = ctx.needs_input_grad[3:]
        grads = [torch.empty_like(p, memory_format=torch.contiguous_format) if n else None for p, n in zip(ctx.params, needs)]
        by_id = {id(p): g for p, g in zip(ctx.params, grads)}
        dw = _lib.WeightGrads()
        model._fill_weight_struct(dw, lambda p: by_id[id(p)].data_ptr() if by_id.get(id(p)) is not None else None)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            rc = lib.iefvad_train_backward(model._handle, ctx.B, C.c_void_p(ws.data_ptr()), ws.numel(), C.byref(dout), C.byref(dw),
                                           C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("iefvad_train_backward: " + _lib.last_error())
        ctx.ws = None
        return (None, None, None, *grads)


class MMFMIL(nn.Module):
    """Same constructor as the reference's MMFMIL (/root/reference/model/imf_vad.py:6-18).

    Extra, keyword-only knobs (not in the reference):
      outputs      "full" (default; all eight tensors, the drop-in behaviour), "scores"
                   (only `logits` plus the per-row means `w_i_mean`, `w_e_mean`; nothing 768-wide
                   is written to HBM) or "weights" (the scores set plus the full `w_i`, `w_e`: what the
                   robustness sweep reads, test2.py:65-68).
      micro_batch  chunks per internal pass of the library (0 = library default).
      graph_chunks calls with B <= graph_chunks replay a cached hipGraph of the forward instead of launching its ~31
                   kernels one by one (0 = library default, 8; negative = never).  Same bits either way.
      compute      "f32" (default): exact-fp32 MFMA projections, the parity mode;
                   "bf16": bf16 MFMA operands with fp32 accumulation in the dense projections and in the two
                   attention products; softmax, LayerNorm, the residual stream, the fusion and the refinement
                   state stay fp32 (BASELINE config 3);
                   "bf16x6": the fp32 path with every dense projection computed as six bf16 MFMA products of the exact
                 three-term bf16 split of both fp32 operands (csrc/gemm_split.h): fp32-accurate -- held to the same
                 tolerances as "f32" -- at the bf16 matrix-core rate.  Batches too small to fill the chip run on the
                 "f32" kernels, so scores are fp32-accurate but not bit-identical across batch sizes in this mode.
                   "fp16x3" (opt-in, near-fp32): the "bf16x6" flow with two fp16 terms per operand and three products per
                 multiply-add (22-bit products, half the MFMAs); operands are scaled by powers of two from running max |.|
                 words the producing kernels maintain (range-safe).  Per-projection error vs fp64 ~1.8x the fp32 path's.
    """

    def __init__(self, num_class: int, embed_dim: int, visual_length: int, visual_width: int, visual_head: int,
                 visual_layers: int, attn_window: int, prompt_prefix: int, prompt_postfix: int, device, args,
                 *, outputs: str = "full", micro_batch: int = 0, compute: str = "f32", graph_chunks: int = 0):
        super().__init__()
        self.num_class = num_class
        self.visual_length = visual_length
        self.visual_width = visual_width
        self.embed_dim = embed_dim
        self.attn_window = attn_window
        self.prompt_prefix = prompt_prefix
        self.prompt_postfix = prompt_postfix
        self.device = device
        self.temporal = FusionParams(embed_dim, num_layers=args.visual_layers, num_heads=args.visual_head,
                                     num_refinement_steps=args.num_refinement_steps, lambda_ref=args.lambda_ref,
                                     noise_model=args.noise_model, nu=args.nu)
        if outputs not in ("full", "scores", "weights"):
            raise ValueError("outputs must be 'full', 'scores' or 'weights'")
        if compute not in _lib.COMPUTE_CODES:
            raise ValueError("compute must be 'f32', 'bf16', 'bf16x6' or 'fp16x3'")
        self.outputs = outputs
        self.micro_batch = micro_batch
        self.compute = compute
        self.graph_chunks = graph_chunks
        self._handle: Optional[C.c_void_p] = None
        self._handle_key = None
        self._weights_sig = None
        self._workspace: Optional[torch.Tensor] = None
        self.last_stage_times: Optional[dict] = None

    # ------------------------------------------------------------------ library plumbing
    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def _release(self):
        if self._handle is not None:
            _lib.load_library().iefvad_destroy(self._handle)
            self._handle = None
            self._handle_key = None
            self._weights_sig = None

    def _noise_code(self) -> int:
        nm = self.temporal.noise_model
        if nm == "Gaussian":
            return _lib.NOISE_GAUSSIAN
        if nm == "StudentT":
            return _lib.NOISE_STUDENT_T
        raise ValueError("Unsupported noise_model. Choose 'Gaussian' or 'StudentT'.")   # imf_vad.py:137-138

    def _ensure_handle(self, device: torch.device):
        t = self.temporal
        key = (device.index, t.embed_dim, self.visual_length, t.num_heads, t.num_layers, t.num_refinement_steps,
               self._noise_code(), float(t.lambda_ref), float(t.nu), float(t.epsilon), int(self.micro_batch), self.compute,
               int(self.graph_chunks))
        if self._handle is not None and key == self._handle_key:
            return
        self._release()
        lib = _lib.load_library()
        cfg = _lib.Config(abi_version=_lib.ABI_VERSION, embed_dim=t.embed_dim, seq_len=self.visual_length,
                          num_heads=t.num_heads, num_layers=t.num_layers, num_steps=t.num_refinement_steps,
                          noise_model=self._noise_code(),
                          compute=_lib.COMPUTE_CODES[self.compute], lambda_ref=float(t.lambda_ref),
                          nu=float(t.nu), epsilon=float(t.epsilon), micro_batch=int(self.micro_batch),
                          graph_chunks=int(self.graph_chunks))
        h = C.c_void_p()
        with torch.cuda.device(device):
            rc = lib.iefvad_create(C.byref(cfg), C.byref(h))
        if rc != 0:
            raise RuntimeError("iefvad_create: " + _lib.last_error())
        self._handle, self._handle_key = h, key

    def _ensure_weights(self, device: torch.device, stream: int):
        t = self.temporal
        # (owner's _parameters dict, name) of every parameter, resolved once: walking the module tree on every call costs
        # more host time than the rest of a one-chunk forward's enqueue; looking the slots up each time still sees a
        # Parameter that was re-assigned
        slots = self.__dict__.get("_param_slots")
        if slots is None:
            slots = [(m._parameters, n) for m in t.modules() for n in m._parameters if m._parameters[n] is not None]
            self.__dict__["_param_slots"] = slots
        params = [d[n] for d, n in slots]
        sig = tuple((p.data_ptr(), p._version) for p in params)
        if sig == self._weights_sig:
            return
        for p in params:
            if p.device != device or p.dtype != torch.float32 or not p.is_contiguous():
                raise RuntimeError("MMFMIL parameters must be contiguous fp32 tensors on the input's device "
                                   f"({device}); call model.to(device)")
        w = _lib.Weights()
        self._fill_weight_struct(w, lambda p: p.data_ptr())
        rc = _lib.load_library().iefvad_set_weights(self._handle, C.byref(w), C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("iefvad_set_weights: " + _lib.last_error())
        self._weights_sig = sig

    def _fill_weight_struct(self, w, ptr):
        """Fill an `iefvad_weights`-shaped ctypes structure (Weights or WeightGrads) with `ptr(parameter)` per field."""
        t = self.temporal
        for m, name in enumerate(("image", "event")):
            attn = getattr(t, f"{name}_attn_layers")
            norms = getattr(t, f"{name}_norms")
            for l in range(t.num_layers):
                w.in_proj_w[m][l] = ptr(attn[l].in_proj_weight)
                w.in_proj_b[m][l] = ptr(attn[l].in_proj_bias)
                w.out_proj_w[m][l] = ptr(attn[l].out_proj.weight)
                w.out_proj_b[m][l] = ptr(attn[l].out_proj.bias)
                w.norm_w[m][l] = ptr(norms[l].weight)
                w.norm_b[m][l] = ptr(norms[l].bias)
            wh = getattr(t, f"whiten_{name}")
            w.whiten_w[m], w.whiten_b[m] = ptr(wh.weight), ptr(wh.bias)
            mu, lv = getattr(t, f"{name}_mu"), getattr(t, f"{name}_logvar")
            w.mu_w[m], w.mu_b[m] = ptr(mu.weight), ptr(mu.bias)
            w.logvar_w[m], w.logvar_b[m] = ptr(lv.weight), ptr(lv.bias)
        for k in range(t.num_refinement_steps):
            blk = t.refinement_blocks[k]
            l1, l2 = getattr(blk, "0"), getattr(blk, "2")
            w.ref_w1[k], w.ref_b1[k] = ptr(l1.weight), ptr(l1.bias)
            w.ref_w2[k], w.ref_b2[k] = ptr(l2.weight), ptr(l2.bias)
        w.cls_w, w.cls_b = ptr(t.classifier.weight), ptr(t.classifier.bias)

    def lanes(self, n: int):
        """`n` forward lanes over ONE set of parameters: lane 0 is this module, the others are shallow copies that share
        `self.temporal` (the same Parameter objects) and own a library handle, repacked weights and workspace each, so that
        forwards issued on different HIP streams can run at the same time (harness.score_loader(lanes=...): the one-chunk
        forwards of the reference's per-video loop leave most of the chip idle).  Cached on the module."""
        import copy
        have = self.__dict__.setdefault("_lane_copies", [])
        while len(have) < n - 1:
            c = copy.copy(self)
            c.__dict__["_lane_copies"] = []
            c.__dict__["_stagers"] = {}          # a lane stages through buffers of its own (harness.score_loader), never lane 0's
            c.__dict__.pop("_param_slots", None)
            c._handle = c._handle_key = c._weights_sig = c._workspace = None
            have.append(c)
        return [self] + have[:max(n - 1, 0)]

    def refresh_weights(self):
        """Force the library to re-read (and re-pack / re-split) the parameters on the next forward.  The shim notices
        `load_state_dict`, `.to()`, optimizer steps and any in-place op on a Parameter (they change the storage pointer
        or the tensor version), but NOT writes made through `param.data` (e.g. `p.data.mul_(0.999)` in an EMA or a
        clipping utility): those bypass the version counter, so call this after them."""
        self._weights_sig = None
        for c in self.__dict__.get("_lane_copies", []):
            c._weights_sig = None

    def load_state_dict(self, *args, **kw):
        self.refresh_weights()
        return super().load_state_dict(*args, **kw)

    def _apply(self, fn, *args, **kw):
        self.refresh_weights()
        return super()._apply(fn, *args, **kw)

    # ------------------------------------------------------------------ forward
    def _prepare_input(self, x: torch.Tensor) -> torch.Tensor:
        if x.dtype not in _IN_DTYPES:
            x = x.to(torch.float)        # imf_vad.py:41-42 for fp64 / integer inputs
        return x.contiguous()

    def forward(self, img_visual, ev_visual, padding_mask=None, text=None, lengths=None, return_attn=False,
                *, timed: bool = False, row_scale=None) -> Dict[str, torch.Tensor]:
        """The reference's call (imf_vad.py:40-44; `padding_mask`, `text`, `lengths`, `return_attn` accepted and ignored as there).
        Keyword-only extras: `timed` (per-stage device times in `last_stage_times`), `row_scale=(s_img, s_ev)`: fp32 DEVICE vectors
        of [B*T] (either may be None) that multiply the input rows inside the library's input load (`iefvad_forward_scaled`: the
        robustness sweep's `x[:, idx] * 0.01`, test2.py:71-77, without touching the caller's tensors)."""
        self._noise_code()   # ValueError for an unsupported noise_model, as the reference raises
        if self.training:
            return self._forward_train(img_visual, ev_visual)
        if not (img_visual.is_cuda and ev_visual.is_cuda):
            raise RuntimeError("iefvad_amd.MMFMIL runs on a HIP device only; there is no CPU fallback "
                               "(move the inputs with .to('cuda'))")
        if img_visual.shape != ev_visual.shape or img_visual.dim() != 3:
            raise ValueError(f"expected two [B, T, D] tensors of equal shape, got {tuple(img_visual.shape)} and "
                             f"{tuple(ev_visual.shape)}")
        B, T, D = img_visual.shape
        if T != self.visual_length or D != self.temporal.embed_dim:
            raise ValueError(f"expected [B, {self.visual_length}, {self.temporal.embed_dim}] inputs, got [B, {T}, {D}]")
        device = img_visual.device
        img = self._prepare_input(img_visual)
        ev = self._prepare_input(ev_visual)
        if ev.dtype != img.dtype:
            img, ev = img.to(torch.float), ev.to(torch.float)
        lib = _lib.load_library()
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            self._ensure_handle(device)
            self._ensure_weights(device, stream)
            need = lib.iefvad_workspace_bytes(self._handle, B)
            if self._workspace is None or self._workspace.device != device or self._workspace.numel() < need:
                self._workspace = None
                self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
            N = B * T
            f32 = dict(dtype=torch.float32, device=device)
            res: Dict[str, torch.Tensor] = {}
            o = _lib.Outputs()
            res["logits"] = torch.empty(B, T, 1, **f32)
            o.logits = res["logits"].data_ptr()
            if self.outputs == "full":
                for k in OUTPUT_KEYS:
                    if k != "logits":
                        res[k] = torch.empty(B, T, D, **f32)
                        setattr(o, k, res[k].data_ptr())
            else:
                if self.outputs == "weights":       # the sweep's set (test2.py:65-68,86-87): both weight tensors, and their row means
                    for k in ("w_i", "w_e"):
                        res[k] = torch.empty(B, T, D, **f32)
                        setattr(o, k, res[k].data_ptr())
                res["w_i_mean"] = torch.empty(B, T, **f32)
                res["w_e_mean"] = torch.empty(B, T, **f32)
                o.w_i_mean, o.w_e_mean = res["w_i_mean"].data_ptr(), res["w_e_mean"].data_ptr()
            args = (self._handle, C.c_void_p(img.data_ptr()), C.c_void_p(ev.data_ptr()), _IN_DTYPES[img.dtype], B,
                    C.c_void_p(self._workspace.data_ptr()), self._workspace.numel(), C.byref(o), C.c_void_p(stream))
            if row_scale is not None and (row_scale[0] is not None or row_scale[1] is not None):
                sp = []
                for sv in row_scale:
                    if sv is not None and (sv.dtype != torch.float32 or sv.device != device or sv.numel() != N or not sv.is_contiguous()):
                        raise ValueError(f"row_scale vectors must be contiguous fp32 tensors of {N} elements on {device}")
                    sp.append(C.c_void_p(sv.data_ptr()) if sv is not None else None)
                rc = lib.iefvad_forward_scaled(*args[:5], sp[0], sp[1], *args[5:])
            elif timed:
                st = _lib.StageTimes()
                rc = lib.iefvad_forward_timed(*args, C.byref(st))
                self.last_stage_times = st.as_dict()
            else:
                rc = lib.iefvad_forward(*args)
            if rc != 0:
                raise RuntimeError("iefvad_forward: " + _lib.last_error())
        if self.outputs == "full":
            return {k: res[k] for k in OUTPUT_KEYS}   # the reference's key order
        return res

    def forward_videos_host(self, imgs, evs, lengths, nan_to_num: bool = True, batch_chunks: int = 128, host_threads: int = 0,
                            wire_dtype: Optional[torch.dtype] = None) -> Dict[str, torch.Tensor]:
        """A whole list of videos in ONE library call (`iefvad_forward_videos_host`): `imgs[v]`, `evs[v]` are contiguous HOST tensors
        of one dtype whose first `lengths[v]` rows ([..., D]) are video v's features -- e.g. the padded tensors a DataLoader
        delivers (data/dataset.py:34-52).  The library packs whole videos into passes of >= `batch_chunks` chunks, stages and sends
        pass k + 1 while pass k computes, and returns DEVICE vectors `logits`, `w_i_mean`, `w_e_mean` of [sum(lengths)] in list
        order (stream-ordered on the current stream).  Same results as `forward_videos` on the same batches.
        `wire_dtype=torch.bfloat16` (fp32 features, compute="bf16" only): the gather threads round the rows to bf16 while staging, so
        half the bytes cross PCIe; same results as `forward_videos` on `rows.to(torch.bfloat16)` (include/iefvad.h)."""
        if self.training:
            raise RuntimeError("iefvad_amd.MMFMIL.forward_videos_host is an evaluation entry point; call model.eval() first")
        self._noise_code()
        lens = [int(n) for n in lengths]
        if not lens or min(lens) < 1 or len(imgs) != len(lens) or len(evs) != len(lens):
            raise ValueError("imgs, evs and lengths must list the same videos, every video at least one snippet")
        dt = imgs[0].dtype
        D = self.temporal.embed_dim
        if dt not in _IN_DTYPES:
            raise ValueError(f"feature dtype {dt} is not supported by the list entry (fp32, fp16 or bf16)")
        wire = dt if wire_dtype is None else wire_dtype
        if wire not in _IN_DTYPES:
            raise ValueError(f"wire dtype {wire} is not supported (the feature dtype, or torch.bfloat16 for fp32 features)")
        n = len(lens)
        # validation as C-speed sweeps (a per-video Python loop costs more than the library call on lists of short videos)
        for parts in (imgs, evs):
            if any(t.dtype != dt for t in parts) or any(t.is_cuda for t in parts) or not all(t.is_contiguous() for t in parts) \
                    or any(t.shape[-1] != D for t in parts) or any(t.numel() < ln * D for t, ln in zip(parts, lens)):
                raise ValueError(f"expected contiguous host tensors of dtype {dt} with at least lengths[v] rows of {D} each")
        pi = (C.c_void_p * n)(*[t.data_ptr() for t in imgs])
        pe = (C.c_void_p * n)(*[t.data_ptr() for t in evs])
        device = next(self.temporal.parameters()).device
        if device.type != "cuda":
            raise RuntimeError("iefvad_amd.MMFMIL runs on a HIP device only; there is no CPU fallback (call model.to('cuda'))")
        total = sum(lens)
        lib = _lib.load_library()
        larr = (C.c_int32 * n)(*lens)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            self._ensure_handle(device)
            self._ensure_weights(device, stream)
            f32 = dict(dtype=torch.float32, device=device)
            res = {"logits": torch.empty(total, **f32), "w_i_mean": torch.empty(total, **f32), "w_e_mean": torch.empty(total, **f32)}
            rc = lib.iefvad_forward_videos_host(self._handle, pi, pe, _IN_DTYPES[dt], _IN_DTYPES[wire], larr, n, 1 if nan_to_num else 0, int(batch_chunks),
                                                int(host_threads), C.c_void_p(res["logits"].data_ptr()),
                                                C.c_void_p(res["w_i_mean"].data_ptr()), C.c_void_p(res["w_e_mean"].data_ptr()),
                                                C.c_void_p(stream))
        if rc != 0:
            raise RuntimeError("iefvad_forward_videos_host: " + _lib.last_error())
        return res

    # ------------------------------------------------------------------ train mode
    def _forward_train(self, img_visual, ev_visual) -> Dict[str, torch.Tensor]:
        """`model.train()` forward (/root/reference/train/ucf_train.py:43,60-66): the same dict, differentiable with respect to
        every parameter -- `iefvad_train_forward` keeps the activations, `loss.backward()` reaches `iefvad_train_backward`
        through `_TrainForward`.  Attention dropout (imf_vad.py:70) uses each layer's `.dropout` (as nn.MultiheadAttention
        keeps it); the mask comes from the library's counter-based generator seeded from torch's RNG (`self.dropout_seed` pins
        it), or from `self.dropout_mask` (uint8 [2, L, B, 8, T, T], 1 = keep) when a test injects one."""
        if self.compute not in ("f32", "bf16x6"):
            raise RuntimeError("iefvad_amd.MMFMIL trains in the fp32-accurate arithmetics only: compute='f32' or 'bf16x6'")
        if not (img_visual.is_cuda and ev_visual.is_cuda):
            raise RuntimeError("iefvad_amd.MMFMIL runs on a HIP device only; there is no CPU fallback "
                               "(move the inputs with .to('cuda'))")
        if img_visual.shape != ev_visual.shape or img_visual.dim() != 3:
            raise ValueError(f"expected two [B, T, D] tensors of equal shape, got {tuple(img_visual.shape)} and "
                             f"{tuple(ev_visual.shape)}")
        B, T, D = img_visual.shape
        if T != self.visual_length or D != self.temporal.embed_dim:
            raise ValueError(f"expected [B, {self.visual_length}, {self.temporal.embed_dim}] inputs, got [B, {T}, {D}]")
        slots = self.__dict__.get("_param_slots")
        if slots is None:
            slots = [(m._parameters, n) for m in self.temporal.modules() for n in m._parameters if m._parameters[n] is not None]
            self.__dict__["_param_slots"] = slots
        params = [d[n] for d, n in slots]
        outs = _TrainForward.apply(self, img_visual.detach(), ev_visual.detach(), *params)
        return dict(zip(OUTPUT_KEYS, outs))

    def forward_videos(self, img_rows: torch.Tensor, ev_rows: torch.Tensor, lengths, nan_to_num: bool = True
                       ) -> Dict[str, torch.Tensor]:
        """Scores of whole videos (`iefvad_forward_videos`, include/iefvad.h): `img_rows`, `ev_rows` are the videos' VALID
        feature rows concatenated in list order, [sum(lengths), D] on the device; `lengths` the snippets per video.  The
        chunker (tools.py:100-114), the conditional nan_to_num of test.py:90-95 and the `[0:len]` slicing of
        test.py:119-121,131-138 happen on the device.  Returns `logits`, `w_i_mean`, `w_e_mean`, each [sum(lengths)]."""
        if self.training:
            raise RuntimeError("iefvad_amd.MMFMIL.forward_videos is an evaluation entry point; call model.eval() first")
        self._noise_code()
        if not (img_rows.is_cuda and ev_rows.is_cuda):
            raise RuntimeError("iefvad_amd.MMFMIL runs on a HIP device only; there is no CPU fallback")
        lens = [int(n) for n in lengths]
        total = sum(lens)
        D = self.temporal.embed_dim
        if img_rows.shape != ev_rows.shape or img_rows.dim() != 2 or tuple(img_rows.shape) != (total, D):
            raise ValueError(f"expected two [{total}, {D}] tensors (sum of lengths x D), got {tuple(img_rows.shape)} and "
                             f"{tuple(ev_rows.shape)}")
        if not lens or min(lens) < 1:
            raise ValueError("every video needs at least one snippet")
        device = img_rows.device
        img = self._prepare_input(img_rows)
        ev = self._prepare_input(ev_rows)
        if ev.dtype != img.dtype:
            img, ev = img.to(torch.float), ev.to(torch.float)
        lib = _lib.load_library()
        larr = (C.c_int32 * len(lens))(*lens)
        with torch.cuda.device(device):
            stream = torch.cuda.current_stream(device).cuda_stream
            self._ensure_handle(device)
            self._ensure_weights(device, stream)
            need = lib.iefvad_videos_workspace_bytes(self._handle, larr, len(lens))
            if self._workspace is None or self._workspace.device != device or self._workspace.numel() < need:
                self._workspace = None
                self._workspace = torch.empty(need, dtype=torch.uint8, device=device)
            f32 = dict(dtype=torch.float32, device=device)
            res = {"logits": torch.empty(total, **f32), "w_i_mean": torch.empty(total, **f32), "w_e_mean": torch.empty(total, **f32)}
            rc = lib.iefvad_forward_videos(self._handle, C.c_void_p(img.data_ptr()), C.c_void_p(ev.data_ptr()),
                                           _IN_DTYPES[img.dtype], larr, len(lens), 1 if nan_to_num else 0,
                                           C.c_void_p(self._workspace.data_ptr()), self._workspace.numel(),
                                           C.c_void_p(res["logits"].data_ptr()), C.c_void_p(res["w_i_mean"].data_ptr()),
                                           C.c_void_p(res["w_e_mean"].data_ptr()), C.c_void_p(stream))
            if rc != 0:
                raise RuntimeError("iefvad_forward_videos: " + _lib.last_error())
        return res
