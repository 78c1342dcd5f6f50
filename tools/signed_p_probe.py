"""Probe of the train forward's sign-carrying attention probabilities (attention_split.h TRAIN): reads P of (image, layer 0) out of the
training buffer and checks |P| against softmax(q k^T) of the stored q | k, the sign bits against common.h's dropout_bits re-derived in
torch, and the stored attention output against dropout(P) v.  Found the hipcc vector-element bit_cast miscompile (common.h).
    python3 tools/signed_p_probe.py"""
import os, sys, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_train as T
img, ev, labels, lengths = T.batch(52, 3)
model, _ = T.make_model(41, 2, 2, "StudentT", 8, "bf16x6", 0.1)
model.dropout_seed = 4321
model.train()
out = model(img, ev, None, None, lengths)
ws = out["logits"].grad_fn.ws.view(torch.float32)
B = 3; U = B * 256 * 768; PU = B * 8 * 256 * 256
qkv = ws[3 * U:6 * U].view(B, 256, 3, 8, 96)
att = ws[6 * U:7 * U].view(B, 256, 8, 96)
P = ws[8 * U:8 * U + PU].view(B, 8, 256, 256)
neg = (P.view(torch.int32) < 0)
print("fraction of sign bits set", float(neg.float().mean()))
print("row sums of |P|: min %.6f max %.6f" % (float(P.abs().sum(-1).min()), float(P.abs().sum(-1).max())))
q = qkv[:, :, 0].permute(0, 2, 1, 3).double(); k = qkv[:, :, 1].permute(0, 2, 1, 3).double(); v = qkv[:, :, 2].permute(0, 2, 1, 3).double()
Pref = torch.softmax(q @ k.transpose(-1, -2), -1)
print("max | |P| - softmax(q k^T) |", float((P.abs().double() - Pref).abs().max()))
Pd = torch.where(neg, torch.zeros_like(P), P * (1.0 / 0.9)).double()
a2 = (Pd @ v).permute(0, 2, 1, 3)
print("max |att - Pd v|", float((att.double() - a2).abs().max()), "max |att|", float(att.abs().max()))
a3 = ((P.abs().double() * (1 / 0.9)) @ v).permute(0, 2, 1, 3)
print("max |att - |P| v / 0.9| (no mask)", float((att.double() - a3).abs().max()))
# per key-tile error
err = (att.double() - a2).abs()
print("err by head", [float(err[:, :, h].max()) for h in range(8)])
# expected mask from the counter-based generator (common.h dropout_bits)
M32 = 0xFFFFFFFF
def fmix32(x):
    x = x & M32
    x ^= x >> 16; x = (x * 0x85EBCA6B) & M32
    x ^= x >> 13; x = (x * 0xC2B2AE35) & M32
    return x ^ (x >> 16)
MAXL = int(os.environ.get("MAXL", "4"))
seed = (4321 * 0x100000001B3 + (0 * MAXL + 0 + 1) * 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
idx = torch.arange(PU, dtype=torch.int64, device="cuda")
lo = idx & M32; hi = idx >> 32
a = fmix32(lo ^ (seed & M32))
bits = fmix32((a + (seed >> 32) + hi * 0x9E3779B1) & M32) >> 8
thr = int(0.1 * 16777216.0)
exp_drop = (bits < thr).view(B, 8, 256, 256)
print("expected dropped fraction", float(exp_drop.float().mean()))
print("sign set but not expected", int((neg & ~exp_drop).sum()), " expected but not set", int((exp_drop & ~neg).sum()))
bad = (neg & ~exp_drop)
w = bad.nonzero()
print("first unexpected:", w[:12].tolist())
ratio = (P.abs().double() / Pref)
print("ratio |P| / softmax: min %.4f max %.4f" % (float(ratio.min()), float(ratio.max())))
rb = ((ratio - 1).abs() > 1e-4)
print("elements with |ratio - 1| > 1e-4:", int(rb.sum()), "of", rb.numel(), "first", rb.nonzero()[:12].tolist())
