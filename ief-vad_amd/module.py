"""Host-side mirrors of /root/reference/model/module.py: `LayerNorm`, `QuickGELU`, `ResidualAttentionBlock`,
`Transformer` with the reference's constructor arguments, parameter names (state_dict keys `attn.in_proj_weight`, `ln_1.weight`,
`mlp.c_fc.weight`, ...) and the `(x, padding_mask)` tuple plumbing of its forward.  A block's forward is one library call
(`iefvad_resblock_forward`, csrc/vadclip.h); the parameter holders below compute nothing.  HIP tensors only."""
import ctypes as C
from collections import OrderedDict

import torch
import torch.nn as nn

from . import lib as _lib
from .layers import _check, _f32, _need_cuda, _p, _stream, _ws


class LayerNorm(nn.LayerNorm):
    """module.py:7-12: a parameter holder here (the block's kernels normalise in fp32, as the reference's subclass forces)."""


class QuickGELU(nn.Module):
    """module.py:15-17: x * sigmoid(1.702 x); inside a block it is the epilogue of the c_fc product."""

    def forward(self, x: torch.Tensor):
        raise RuntimeError("QuickGELU is fused into iefvad_resblock_forward / iefvad_gcn_forward; it is not called on its own")


class ResidualAttentionBlock(nn.Module):
    """module.py:20-43.  `forward((x [T, B, d_model], padding_mask [B, T] or None)) -> (x, padding_mask)`."""

    def __init__(self, d_model: int, n_head: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.attn = nn.MultiheadAttention(d_model, n_head)
        self.ln_1 = LayerNorm(d_model)
        self.mlp = nn.Sequential(OrderedDict([("c_fc", nn.Linear(d_model, d_model * 4)), ("gelu", QuickGELU()),
                                              ("c_proj", nn.Linear(d_model * 4, d_model))]))
        self.ln_2 = LayerNorm(d_model)
        self.attn_mask = attn_mask
        self.d_model, self.n_head = d_model, n_head

    def forward(self, x):
        x, padding_mask = x
        _need_cuda(x, self.attn.in_proj_weight)
        xin = _f32(x)
        T, B, D = xin.shape
        dev = xin.device
        am = None
        if self.attn_mask is not None:
            am = self.attn_mask.to(dev)
            am = torch.zeros_like(am, dtype=torch.float32).masked_fill_(am, float("-inf")) if am.dtype == torch.bool else _f32(am)
        kp = padding_mask.to(device=dev, dtype=torch.bool).to(torch.uint8).contiguous() if padding_mask is not None else None
        w = _lib.ResblockWeights()
        keep = []
        for name, t in (("in_proj_w", self.attn.in_proj_weight), ("in_proj_b", self.attn.in_proj_bias), ("out_proj_w", self.attn.out_proj.weight),
                        ("out_proj_b", self.attn.out_proj.bias), ("ln_1_w", self.ln_1.weight), ("ln_1_b", self.ln_1.bias), ("ln_2_w", self.ln_2.weight),
                        ("ln_2_b", self.ln_2.bias), ("c_fc_w", self.mlp.c_fc.weight), ("c_fc_b", self.mlp.c_fc.bias), ("c_proj_w", self.mlp.c_proj.weight),
                        ("c_proj_b", self.mlp.c_proj.bias)):
            t = _f32(t)
            keep.append(t)
            setattr(w, name, t.data_ptr())
        lib = _lib.load_library()
        with torch.cuda.device(dev):
            out = torch.empty(T, B, D, dtype=torch.float32, device=dev)
            ws = _ws(lib.iefvad_resblock_workspace_bytes(T, B, self.n_head), dev)
            _check(lib.iefvad_resblock_forward(_p(xin), C.byref(w), _p(am), _p(kp), T, B, D, self.n_head, _p(out), _p(ws), ws.numel(), _stream(dev)),
                   "iefvad_resblock_forward")
        return (out.to(x.dtype), padding_mask)


class Transformer(nn.Module):
    """module.py:46-54: `layers` blocks in sequence."""

    def __init__(self, width: int, layers: int, heads: int, attn_mask: torch.Tensor = None):
        super().__init__()
        self.width = width
        self.layers = layers
        self.resblocks = nn.Sequential(*[ResidualAttentionBlock(width, heads, attn_mask) for _ in range(layers)])

    def forward(self, x: torch.Tensor):
        return self.resblocks(x)
