#!/usr/bin/env python3
"""HBM-side bytes per launch of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected
separately, as MI355X_MICROARCH.md's HBM section prescribes), calibrated on iefvad_layernorm_kernel, which streams a
known byte count in the same passes (gfx950: FETCH_SIZE tallies 128-B requests at 64 B -> doubled when the calibration
shows 1/2).  The kernel argument may name several substrings joined by '|' (two tilings of one GEMM): the mean is over all
their launches.   usage: hbm_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <kernel substring>
                                   <rows per launch> <out.json> "<source description>" [<algorithmic read bytes per launch>
                                   <algorithmic write bytes per launch> "<how they are counted>"]
The JSON names the kernels it covers and the git head it was collected at: bench.py refuses a figure whose kernel list does
not match the kernels the build dispatches."""
import csv
import json
import sys

fetch_csv, write_csv, kname, rows, out, source = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5], sys.argv[6]


def means(path, counter):
    acc = {}
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        acc.setdefault(r["Kernel_Name"], []).append(float(r["Counter_Value"]))
    out = {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
    hits = [x for k, v in acc.items() if any(sub in k for sub in kname.split("|")) for x in v]
    if hits:
        out[kname] = (sum(hits) / len(hits), len(hits))
    return out


f, w = means(fetch_csv, "FETCH_SIZE"), means(write_csv, "WRITE_SIZE")
pick = lambda d, sub: next((v for k, v in d.items() if sub in k), None)
ln_f, ln_w = pick(f, "iefvad_layernorm_kernel"), pick(w, "iefvad_layernorm_kernel")
cal_name = "iefvad_layernorm_kernel"
ln_bytes = 2 * rows * 768 * 4                                  # both modalities, one fp32 tensor in, one out
if ln_f is None:        # bf16 mode with the fused out_proj + LayerNorm kernel: calibrate on the scorer (reads z once)
    cal_name = "iefvad_scorer_kernel"
    ln_f, ln_w = pick(f, cal_name), pick(w, cal_name)
    ln_bytes = rows * 768 * 4
if ln_f is None:        # ... and with the refinement chain (scorer inside it): the input cast reads both fp32 blocks once
    cal_name = "iefvad_cast_kernel"
    def biggest(path, counter):      # the full-size launch (fp32 inputs of the micro-batch), not set_weights' small conversions
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter and cal_name in r["Kernel_Name"]]
        return (max(vals), 1)
    ln_f, ln_w = biggest(fetch_csv, "FETCH_SIZE"), biggest(write_csv, "WRITE_SIZE")
    ln_bytes = 2 * rows * 768 * 4
ratio = ln_bytes / (ln_f[0] * 1024)
if 0.8 <= ratio <= 2.5:
    fetch_scale = round(ratio)                                  # 2 on gfx950
    cal_text = (f"{cal_name} streams {ln_bytes} B in: WRITE_SIZE reads {ln_w[0]:.0f} KB, "
                f"FETCH_SIZE reads {ln_f[0]:.0f} KB -> FETCH_SIZE x {fetch_scale}")
else:
    # no launch of known streamed bytes in this mode's passes (bf16 mode: LayerNorm, scorer and the input cast are all inside
    # the row-block kernels): the factor every calibrated pass of this chip has shown, e.g. the bf16x6 pass of the same collection
    fetch_scale = 2
    cal_text = ("no stand-alone streaming kernel in this mode's passes; FETCH_SIZE x 2 as calibrated on iefvad_layernorm_kernel in "
                "the bf16x6 pass of the same collection (gemm_split_hbm_traffic.json)")
kf, kw = pick(f, kname), pick(w, kname)
res = {"source": source, "kernel": kname, "launches_measured": kf[1], "rows_per_launch": rows,
       "FETCH_SIZE_KB_mean": kf[0], "WRITE_SIZE_KB_mean": kw[0],
       "calibration": cal_text,
       "read_bytes_per_launch": kf[0] * 1024 * fetch_scale, "write_bytes_per_launch": kw[0] * 1024,
       "traffic_bytes_per_launch": kf[0] * 1024 * fetch_scale + kw[0] * 1024}
subs = kname.split("|")
res["kernels"] = subs
res["per_kernel_bytes_per_launch"] = {sub: {"read": pick(f, sub)[0] * 1024 * fetch_scale, "write": pick(w, sub)[0] * 1024, "launches": pick(f, sub)[1]}
                                      for sub in subs if pick(f, sub) and pick(w, sub)}
res["head"] = __import__("os").environ.get("IEFVAD_HEAD")      # the commit the box's snapshot was taken from (no .git on the box)
if len(sys.argv) > 9:
    ar, aw = float(sys.argv[7]), float(sys.argv[8])
    res["algorithmic_bytes_per_launch"] = {"read": ar, "write": aw, "note": sys.argv[9]}
    res["traffic_over_algorithmic"] = res["traffic_bytes_per_launch"] / (ar + aw)
json.dump(res, open(out, "w"), indent=1)
print(json.dumps(res, indent=1))
