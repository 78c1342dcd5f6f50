#!/usr/bin/env python3
"""Reference point for DESIGN 4.3: the vendor library's bf16 GEMM (torch.mm -> hipBLASLt / rocBLAS) at the forward's
projection shapes, M = 262,144 rows, K = 768, bf16 result (torch has no fp32-result form of a bf16 mm).  Not used by the
product; prints TFLOP/s per shape."""
import torch

def bench(M, N, K, iters=20):
    a = torch.randn(M, K, device="cuda", dtype=torch.bfloat16)
    w = torch.randn(N, K, device="cuda", dtype=torch.bfloat16)
    b = torch.randn(N, device="cuda", dtype=torch.bfloat16)
    for fn, name in ((lambda: torch.mm(a, w.t()), "mm"), (lambda: torch.addmm(b, a, w.t()), "addmm(bias)"),
                     (lambda: torch.nn.functional.linear(a, w, b), "linear")):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for _ in range(iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); fn(); e1.record(); torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
        ts.sort()
        med = ts[len(ts) // 2]
        print(f"M={M} N={N} K={K} {name:12s} median {med:.3f} ms  {2.0 * M * N * K / med / 1e9:.1f} TFLOP/s (bf16 in, bf16 out)")

for N in (768, 2304, 1536):
    bench(262144, N, 768)
bench(8192, 8192, 8192, 5)
