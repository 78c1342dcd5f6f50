"""iefvad_auc_ap (csrc/metrics.h) at BASELINE config 4's size -- 2,097,152 snippet scores, 33.5 M frames of ground truth -- for the
rocprofv3 kernel trace of tools/collect_profiles.sh: five calls; prints the result, the wall time per call and the achieved
HBM rate against the algorithmic bytes of the call (pairs kernel 4 + 16 B / snippet in, 8 out; four sort passes of 8 + 16 + 8 B;
scan / apply / groups passes 8 + 8 + 16 B: 124 B per snippet)."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from iefvad_amd import harness, synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192 * 256
gen = torch.Generator(device="cuda").manual_seed(5)
scores = torch.sigmoid(torch.randn(n, device="cuda", generator=gen) * 2)
gt = torch.from_numpy(synth.make_gt(5, n)).to(torch.uint8).cuda()
harness.device_auc_ap(scores, gt)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    auc, ap = harness.device_auc_ap(scores, gt)
dt = (time.perf_counter() - t0) / 5
print(f"iefvad_auc_ap n = {n}: AUC {auc:.12f} AP {ap:.12f}; {dt * 1e3:.3f} ms per call incl. the read-back; "
      f"{124 * n / dt / 1e9:.0f} GB/s of algorithmic traffic (124 B per snippet)")
