"""CPU oracle for the VadCLIP-residue modules of the reference.  TEST INFRASTRUCTURE ONLY: only tests/
may import it; the product (`iefvad_amd.layers`, `iefvad_amd.module` -> libiefvad.so) never does.

Plain-tensor restatements -- no nn.MultiheadAttention / nn.LayerNorm / nn.Linear / nn.Conv1d / F.softmax -- of
  /root/reference/model/layers.py:12-49    GraphAttentionLayer.forward (eval)
  /root/reference/model/layers.py:64-111   GraphConvolution.forward
  /root/reference/model/layers.py:114-163  SimilarityAdj.forward (weight0 used for theta AND phi, :132-133)
  /root/reference/model/layers.py:166-179  DistanceAdj.forward
  /root/reference/model/module.py:20-43    ResidualAttentionBlock.forward (sequence-first, QuickGELU MLP)
taking the reference's state_dict tensors by their own names.

Parity pin: the reference holds no test or fixture for these classes (nothing even imports them).  All but DistanceAdj import on the CPU
in the build container, so tests/golden/make_golden.py runs the reference classes themselves on seeded inputs and stores their outputs
(tests/golden/vadclip_*.npz); tests/test_oracle_golden.py checks this file against them.  DistanceAdj hard-codes `.to('cuda')`
(layers.py:176,178) and cannot run there: PARITY UNPINNED for that one class -- its formula is restated from the source text.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence

import torch


def _softmax_rows(x: torch.Tensor) -> torch.Tensor:
    e = torch.exp(x - x.max(dim=-1, keepdim=True).values)
    return e / e.sum(dim=-1, keepdim=True)


def _layer_norm(x, g, b, eps=1e-5):
    c = x - x.mean(dim=-1, keepdim=True)
    return c / torch.sqrt((c * c).mean(dim=-1, keepdim=True) + eps) * g + b


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * (1.0 / (1.0 + torch.exp(-1.702 * x)))                    # module.py:15-17


def similarity_adj(sd: Dict[str, torch.Tensor], x: torch.Tensor, seq_len: Optional[Sequence[int]]) -> torch.Tensor:
    theta = x @ sd["weight0"]                                           # layers.py:132 (and :133: phi is the same product)
    sim = theta @ theta.transpose(1, 2)
    n = torch.sqrt((theta * theta).sum(dim=2, keepdim=True))
    sim = sim / (n * n.transpose(1, 2) + 1e-20)
    out = torch.zeros_like(sim)
    for i in range(sim.shape[0]):
        m = sim.shape[1] if seq_len is None else int(seq_len[i])
        t = sim[i, :m, :m]
        t = torch.where(t > 0.7, t, torch.zeros_like(t))               # F.threshold(adj2, 0.7, 0)
        out[i, :m, :m] = _softmax_rows(t)
    return out


def distance_adj(batch_size: int, max_seqlen: int, dtype=torch.float32) -> torch.Tensor:
    idx = torch.arange(max_seqlen, dtype=dtype)
    dist = (idx[:, None] - idx[None, :]).abs()                          # pdist(cityblock) of the positions, squareform
    e1 = torch.exp(torch.tensor(1.0, dtype=dtype))
    return torch.exp(-dist / e1).unsqueeze(0).repeat(batch_size, 1, 1)


def graph_convolution(sd: Dict[str, torch.Tensor], x: torch.Tensor, adj: torch.Tensor, residual: bool = True) -> torch.Tensor:
    W = sd["weight"]
    out = adj @ (x @ W)
    if "bias" in sd and sd["bias"] is not None:
        out = out + sd["bias"]
    if not residual:
        return out
    if W.shape[0] == W.shape[1]:
        return out + x
    cw, cb = sd["residual.weight"], sd["residual.bias"]                 # Conv1d(in, out, 5, padding=2) over time, layers.py:84,100-104
    B, T, _ = x.shape
    xp = torch.zeros(B, T + 4, x.shape[2], dtype=x.dtype)
    xp[:, 2:T + 2] = x
    res = cb.expand(B, T, -1).clone()
    for d in range(5):
        res = res + xp[:, d:d + T] @ cw[:, :, d].t()
    return out + res


def graph_attention(sd: Dict[str, torch.Tensor], x: torch.Tensor, adj: torch.Tensor, alpha: float, concat: bool = True) -> torch.Tensor:
    h = x @ sd["W"]
    F_ = h.shape[1]
    a = sd["a"].reshape(-1)
    e = (h @ a[:F_])[:, None] + (h @ a[F_:])[None, :]                    # matmul([h_i | h_j], a), layers.py:32-33
    e = torch.where(e > 0, e, alpha * e)
    att = _softmax_rows(torch.where(adj > 0, e, torch.full_like(e, -9e15)))
    hp = att @ h
    return torch.where(hp > 0, hp, torch.expm1(hp)) if concat else hp


def residual_attention_block(sd: Dict[str, torch.Tensor], x: torch.Tensor, n_head: int, attn_mask: Optional[torch.Tensor] = None,
                             padding_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """x [T, B, D] sequence-first; attn_mask [T, T] additive; padding_mask [B, T] bool (True = padding)."""
    T, B, D = x.shape
    dh = D // n_head

    def attention(y):
        qkv = y @ sd["attn.in_proj_weight"].t() + sd["attn.in_proj_bias"]              # [T, B, 3D]
        q, k, v = (t.reshape(T, B, n_head, dh).permute(1, 2, 0, 3) for t in qkv.split(D, dim=-1))      # [B, H, T, dh]
        s = (q * (1.0 / math.sqrt(dh))) @ k.transpose(-1, -2)
        if attn_mask is not None:
            s = s + attn_mask
        if padding_mask is not None:
            s = s.masked_fill(padding_mask[:, None, None, :], float("-inf"))
        a = (_softmax_rows(s) @ v).permute(2, 0, 1, 3).reshape(T, B, D)
        return a @ sd["attn.out_proj.weight"].t() + sd["attn.out_proj.bias"]

    x = x + attention(_layer_norm(x, sd["ln_1.weight"], sd["ln_1.bias"]))                # module.py:40
    y = _layer_norm(x, sd["ln_2.weight"], sd["ln_2.bias"])
    hfc = quick_gelu(y @ sd["mlp.c_fc.weight"].t() + sd["mlp.c_fc.bias"])
    return x + hfc @ sd["mlp.c_proj.weight"].t() + sd["mlp.c_proj.bias"]                 # module.py:41
