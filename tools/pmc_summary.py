#!/usr/bin/env python3
"""Per-kernel PMC table from several rocprofv3 `--kernel-trace --pmc ...` passes (one directory per pass).
usage: pmc_summary.py <pass dir> [<pass dir> ...]   (each holding *_counter_collection.csv and *_kernel_trace.csv)
mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (duration x clock x 1024 SIMDs), clock = GRBM_GUI_ACTIVE / 8 / duration;
wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES; HBM GB/s = (2 x FETCH_SIZE + WRITE_SIZE) x 1024 / duration (FETCH_SIZE doubled:
gfx950 half-count, MI355X_MICROARCH.md HBM section)."""
import csv
import glob
import sys
from collections import defaultdict

cnt = defaultdict(lambda: defaultdict(list))     # kernel -> counter -> values
dur = defaultdict(list)                          # kernel -> durations (ns)
for d in sys.argv[1:]:
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            dur[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
mean = lambda v: sum(v) / len(v) if v else float("nan")
print(f"{'kernel':44s} {'calls':>6s} {'avg us':>9s} {'clock GHz':>9s} {'mfma_util':>9s} {'wait_any':>8s} {'lds_conf':>9s} {'HBM GB/s':>9s}")
for k in sorted(cnt, key=lambda k: -sum(dur[k])):
    name = k[5:] if k.startswith("void ") else k      # template instantiations are listed as "void name<...>(...)"
    if not name.startswith("iefvad_"):
        continue
    c = cnt[k]
    t_ns = mean(dur[k])
    clock = mean(c.get("GRBM_GUI_ACTIVE", [])) / 8 / t_ns
    util = mean(c.get("SQ_VALU_MFMA_BUSY_CYCLES", [])) / (t_ns * clock * 1024)
    wait = mean(c.get("SQ_WAIT_ANY", [])) / mean(c.get("SQ_WAVE_CYCLES", [])) if c.get("SQ_WAVE_CYCLES") else float("nan")
    hbm = (2 * mean(c.get("FETCH_SIZE", [])) + mean(c.get("WRITE_SIZE", []))) * 1024 / t_ns
    ncalls = max(len(v) for v in c.values())
    print(f"{name.split('(')[0].replace(', ', ',')[:44]:44s} {ncalls:6d} {t_ns / 1e3:9.1f} {clock:9.2f} {util:9.3f} {wait:8.3f} {mean(c.get('SQ_LDS_BANK_CONFLICT', [])):9.3g} {hbm:9.0f}")
