"""The arithmetic behind compute="bf16x6" (csrc/gemm_split.h), restated with torch CPU tensors: every finite fp32 value is
the exact sum of three bf16 values (round-to-nearest of the running remainder), and a dot product rebuilt from the six
largest of the nine bf16 x bf16 partial products is as close to the exact result as an fp32 dot product is.  No GPU."""
import numpy as np
import torch


def split3(x: torch.Tensor):
    p1 = x.to(torch.bfloat16).float()
    r1 = x - p1                       # exact in fp32
    p2 = r1.to(torch.bfloat16).float()
    r2 = r1 - p2                      # exact in fp32
    p3 = r2.to(torch.bfloat16).float()
    return p1, p2, p3


def test_three_bf16_terms_hold_every_fp32_bit():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(1 << 18, generator=g) * torch.exp(torch.empty(1 << 18).uniform_(-40, 40, generator=g))
    x[:6] = torch.tensor([0.0, 1.0, -1.0, 3.0e38, 1.1754944e-38, 7.0e-41])      # incl. a denormal
    p1, p2, p3 = split3(x)
    s = p1.double() + p2.double() + p3.double()
    normal = (x == 0) | (x.abs() >= 1.1754944e-38)
    assert torch.equal(s[normal], x.double()[normal])                 # exact for every normal fp32 value
    assert float((s - x.double())[~normal].abs().max()) < 1e-40       # fp32 denormals: below the smallest bf16 denormal
    nz = x != 0
    assert bool((p2[nz].abs().double() <= x[nz].abs().double() * 2.0 ** -8).all())
    assert bool((p3[nz].abs().double() <= x[nz].abs().double() * 2.0 ** -16).all())


def test_six_products_are_fp32_accurate():
    g = torch.Generator().manual_seed(1)
    M, N, K = 64, 96, 768
    a = torch.randn(M, K, generator=g) * 1.3
    w = (torch.rand(N, K, generator=g) * 2 - 1) / np.sqrt(K)
    exact = a.double() @ w.double().t()
    a1, a2, a3 = (t.double() for t in split3(a))
    w1, w2, w3 = (t.double() for t in split3(w))
    kept = a3 @ w1.t() + a1 @ w3.t() + a2 @ w2.t() + a2 @ w1.t() + a1 @ w2.t() + a1 @ w1.t()
    dropped = a2 @ w3.t() + a3 @ w2.t() + a3 @ w3.t()
    assert torch.allclose(kept + dropped, exact, rtol=0, atol=1e-12)              # nine products are the exact result
    scale = (a.abs().double() @ w.abs().double().t())                             # sum |a w| per output
    assert bool((dropped.abs() <= scale * 2.0 ** -24).all())                      # truncation: worst case 2^-23 per product, far less on random data
    # against an fp32 GEMM: the truncation error of the six-term form is far below fp32 accumulation error
    err_fp32 = (a @ w.t()).double().sub(exact).abs()
    assert float(dropped.abs().max()) < 0.1 * float(err_fp32.max())
    # and with fp32 accumulation of the six-term products (what the MFMA accumulator does, up to ordering)
    acc = torch.zeros(M, N)
    for x, y in ((a3, w1), (a1, w3), (a2, w2), (a2, w1), (a1, w2), (a1, w1)):
        acc = acc + (x.float() @ y.float().t())
    assert float(acc.double().sub(exact).abs().max()) <= 2.0 * float(err_fp32.max()) + 1e-7
