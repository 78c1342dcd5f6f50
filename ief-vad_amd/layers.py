"""Host-side mirrors of /root/reference/model/layers.py: same class names, constructor arguments, parameter
names / shapes / initialisers and call signatures; every forward is ONE call into libiefvad.so (csrc/vadclip.h: MFMA GEMMs plus a
few row kernels).  Upstream nothing imports that file (residue of a deleted model/VADCLIP.py): the classes are provided at module
level so that a model variant that instantiates them finds them.  HIP tensors only -- no CPU fallback.

Eval semantics: `GraphAttentionLayer`'s attention dropout (layers.py:38) is active in train() upstream; here train() raises."""
import ctypes as C
import math
from typing import Optional

import numpy as np
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import lib as _lib


def _need_cuda(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("iefvad_amd.layers runs on a HIP device only; there is no CPU fallback (move the module and its inputs with .to('cuda'))")


def _f32(t: torch.Tensor) -> torch.Tensor:
    return t.detach().to(torch.float32).contiguous()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


def _p(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _check(rc: int, who: str):
    if rc != 0:
        raise RuntimeError(f"{who}: {_lib.last_error()}")


class GraphAttentionLayer(nn.Module):
    """layers.py:12-49.  `forward(input [N, in_features], adj [N, N]) -> [N, out_features]`."""

    def __init__(self, in_features, out_features, dropout, alpha, concat=True):
        super().__init__()
        self.dropout, self.in_features, self.out_features, self.alpha, self.concat = dropout, in_features, out_features, alpha, concat
        self.W = nn.Parameter(nn.init.xavier_uniform_(torch.empty(in_features, out_features), gain=np.sqrt(2.0)), requires_grad=True)
        self.a = nn.Parameter(nn.init.xavier_uniform_(torch.empty(2 * out_features, 1), gain=np.sqrt(2.0)), requires_grad=True)

    def forward(self, input, adj):
        if self.training and self.dropout > 0:
            raise RuntimeError("GraphAttentionLayer: attention dropout (layers.py:38) is not built; call .eval()")
        _need_cuda(input, adj, self.W)
        x, A, W, a = _f32(input), _f32(adj), _f32(self.W), _f32(self.a).reshape(-1)
        N = x.shape[0]
        lib = _lib.load_library()
        with torch.cuda.device(x.device):
            out = torch.empty(N, self.out_features, dtype=torch.float32, device=x.device)
            ws = _ws(lib.iefvad_gat_workspace_bytes(N, self.out_features), x.device)
            _check(lib.iefvad_gat_forward(_p(x), _p(A), _p(W), _p(a), C.c_float(self.alpha), 1 if self.concat else 0, N, self.in_features,
                                          self.out_features, _p(out), _p(ws), ws.numel(), _stream(x.device)), "iefvad_gat_forward")
        return out

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class GraphConvolution(nn.Module):
    """layers.py:64-111.  `forward(input [B, T, in_features], adj [B, T, T]) -> [B, T, out_features]`: adj (input W) (+ bias) + residual,
    the residual being the identity for equal widths and `nn.Conv1d(in, out, kernel_size=5, padding=2)` over time otherwise.
    `act="quick_gelu"` (not in the reference's class) folds the QuickGELU its VadCLIP caller applied behind it into the product."""

    def __init__(self, in_features, out_features, bias=False, residual=True, act: Optional[str] = None):
        super().__init__()
        self.in_features, self.out_features, self.act = in_features, out_features, act
        self.weight = Parameter(torch.empty(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.empty(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()
        if not residual:
            self.residual = None
            self._res_kind = 0
        elif in_features == out_features:
            self.residual = None
            self._res_kind = 1
        else:
            self.residual = nn.Conv1d(in_channels=in_features, out_channels=out_features, kernel_size=5, padding=2)
            self._res_kind = 2

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight)
        if self.bias is not None:
            self.bias.data.fill_(0.1)

    def forward(self, input, adj):
        _need_cuda(input, adj, self.weight)
        x, A = _f32(input), _f32(adj)
        B, T, _ = x.shape
        lib = _lib.load_library()
        cw = _f32(self.residual.weight) if self._res_kind == 2 else None
        cb = _f32(self.residual.bias) if self._res_kind == 2 else None
        bias = _f32(self.bias) if self.bias is not None else None
        W = _f32(self.weight)
        with torch.cuda.device(x.device):
            out = torch.empty(B, T, self.out_features, dtype=torch.float32, device=x.device)
            ws = _ws(lib.iefvad_gcn_workspace_bytes(B, T, self.in_features, self.out_features, self._res_kind), x.device)
            _check(lib.iefvad_gcn_forward(_p(x), _p(A), _p(W), _p(bias), _p(cw), _p(cb), self._res_kind, 1 if self.act == "quick_gelu" else 0, B, T,
                                          self.in_features, self.out_features, _p(out), _p(ws), ws.numel(), _stream(x.device)), "iefvad_gcn_forward")
        return out

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class SimilarityAdj(nn.Module):
    """layers.py:114-163.  `forward(input [B, T, in_features], seq_len) -> [B, T, T]`.  `weight1` exists and is initialised as upstream
    but -- as upstream, which multiplies by weight0 twice (:132-133) -- never read."""

    def __init__(self, in_features, out_features):
        super().__init__()
        self.in_features, self.out_features = in_features, out_features
        self.weight0 = Parameter(torch.empty(in_features, out_features))
        self.weight1 = Parameter(torch.empty(in_features, out_features))
        self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.xavier_uniform_(self.weight0)
        nn.init.xavier_uniform_(self.weight1)

    def forward(self, input, seq_len):
        _need_cuda(input, self.weight0)
        x, W = _f32(input), _f32(self.weight0)
        B, T, _ = x.shape
        lens = None
        if seq_len is not None:
            lens = torch.as_tensor([int(v) for v in seq_len], dtype=torch.int32)
            if lens.numel() != B:
                raise ValueError("seq_len must hold one length per sequence of the batch")
            lens = lens.to(x.device)
        lib = _lib.load_library()
        with torch.cuda.device(x.device):
            adj = torch.empty(B, T, T, dtype=torch.float32, device=x.device)
            ws = _ws(lib.iefvad_similarity_adj_workspace_bytes(B, T, self.out_features), x.device)
            _check(lib.iefvad_similarity_adj(_p(x), _p(W), _p(lens), B, T, self.in_features, self.out_features, _p(adj), _p(ws), ws.numel(),
                                             _stream(x.device)), "iefvad_similarity_adj")
        return adj

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'


class DistanceAdj(nn.Module):
    """layers.py:166-179.  `forward(batch_size, max_seqlen) -> [batch_size, max_seqlen, max_seqlen]`, exp(-|i - j| / e) on the module's
    device; `sigma` is a parameter upstream never reads.  Parity unpinned: the reference hard-codes `.to('cuda')` (:176,178) and
    cannot run in the build container; the formula is restated from its source."""

    def __init__(self):
        super().__init__()
        self.sigma = Parameter(torch.empty(1))
        self.sigma.data.fill_(0.1)

    def forward(self, batch_size, max_seqlen):
        _need_cuda(self.sigma)
        dev = self.sigma.device
        lib = _lib.load_library()
        with torch.cuda.device(dev):
            out = torch.empty(batch_size, max_seqlen, max_seqlen, dtype=torch.float32, device=dev)
            _check(lib.iefvad_distance_adj(batch_size, max_seqlen, _p(out), _stream(dev)), "iefvad_distance_adj")
        self.dist = out
        return out
