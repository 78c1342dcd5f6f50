"""bench.py's output contract, on the GPU: ONE JSON line on stdout carrying the metric of BASELINE.json, the roofline
block on SURVEY.md 8(d)'s algorithmic FLOPs, the CPU-oracle baseline, the side modes and the UCF-sized evaluation block.
A reduced batch keeps this a ~30 s test; the numbers are not checked, the structure and the internal consistency are."""
import json
import math
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_default_line_structure_and_consistency():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--chunks", "1024", "--steps", "2", "--warmup", "1",
                        "--cpu-seconds", "2"], env=env, capture_output=True, text=True, timeout=560)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout[:2000]
    d = json.loads(lines[0])
    assert d["metric"].startswith("snippets/sec") and d["unit"] == "snippets/s" and d["higher_is_better"] is True
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert d["config"]["chunks_total"] == 1024 and d["config"]["snippets_per_step"] == 1024 * 256
    assert abs(d["value"] - 1024 * 256 / (d["ms_per_step"] * 1e-3)) < 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and roof["unit"] == "TFLOP/s" and roof["peak"] == 2500.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-12
    # achieved = algorithmic FLOPs per launch / average launch duration (47,185,920 GEMM FLOPs per snippet)
    assert abs(roof["algorithmic_flops_per_launch"] * roof["launches_per_step"] - 47_185_920 * 1024 * 256) < 1e3
    assert abs(roof["achieved"] - roof["algorithmic_flops_per_launch"] / (roof["avg_launch_ms"] * 1e-3) / 1e12) < 1e-6 * roof["achieved"]
    assert abs(roof["mfma_pipe_util"] - 6 * roof["frac"]) < 1e-12            # bf16x6: six executed products per algorithmic one
    gemm_ms = sum(d["stage_ms_per_step"][k] for k in ("qkv_gemm_ms", "out_gemm_ms", "head_gemm_ms", "refine_gemm_ms"))
    assert gemm_ms <= d["ms_per_step"] * 1.02                                # the dominant kernel's time fits inside the step
    for key, peak in (("f32_mfma_mode", 157.3), ("bf16_mode", 2500.0), ("fp16x3_mode", 2500.0)):
        m = d[key]
        assert m["value"] > 0 and m["roofline"]["peak"] == peak and 0 < m["roofline"]["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "sample" in cb
    u = d["ucf_eval"]
    assert u["videos"] == 290 and 60000 < u["snippets"] < 80000
    for k in ("per_video_f32", "per_video_4lanes_f32", "batched_f32", "batched_bf16x6"):
        assert u[k]["snippets_per_s"] > 0
        assert u[k]["max_abs_score_diff_vs_oracle_on_sample"] <= 2e-6       # fp32 gate on sigmoid(logit)
        assert u[k]["abs_auc_diff_vs_oracle_on_sample"] <= 1e-4             # north star: AUC within 1e-4
    assert u["per_video_f32"]["x_cpu_oracle"] >= 10                          # north star: >= 10x the reference CPU path
    assert u["per_video_4lanes_f32"]["bit_identical_to_per_video_f32"] is True
    for key, nvid in (("xd_eval", 753), ("shang_msad_eval", 438)):           # BASELINE configs 3 and 5: bf16 projections
        e = d[key]
        assert e["videos"] == nvid and e["compute"] == "bf16" and e["snippets_per_s"] > 0
        c = e["vs_fp32_cpu_oracle_on_sample"]
        assert c["max_abs_score_diff"] <= 5e-3                               # bf16 gate on sigmoid(logit), tests/test_gpu_bf16.py
        assert c["abs_auc_diff"] <= 1e-4 and c["abs_ap_diff"] <= 1e-4
        w = e["bf16_wire"]                                                   # the opt-in narrowed wire: beside the fp32-wire figure, same gates
        assert w["snippets_per_s"] > 0 and w["max_abs_score_diff_vs_fp32_wire"] <= 5e-3
        cw = w["vs_fp32_cpu_oracle_on_sample"]
        assert cw["max_abs_score_diff"] <= 5e-3 and cw["abs_auc_diff"] <= 1e-4 and cw["abs_ap_diff"] <= 1e-4
    t = d["train_step"]                                                      # both arithmetic modes of the training step
    assert "compute=bf16x6" in t["workload"] and "compute=f32" in t["f32_mode"]["workload"]
    for m in (t, t["f32_mode"]):
        assert m["snippets_per_s"] > 0 and abs(m["snippets_per_s"] - 128 * 256 / (m["ms_per_step"] * 1e-3)) < 1e-6 * m["snippets_per_s"]
        assert math.isfinite(m["loss_total"])
        assert 0 < m["frac"] < 1 and abs(m["frac"] - m["achieved_tflops"] / m["peak_tflops"]) < 1e-9 and 0 < m["mfma_pipe_util"] <= 1
    assert t["peak_tflops"] == 2500.0 and t["f32_mode"]["peak_tflops"] == 157.3           # each mode against the pipe it issues on
    mt = d["metric_tail"]                                                                # iefvad_auc_ap on config 4's 2.1 M scores
    assert mt["snippets"] == 8192 * 256 and mt["device_ms"] > 0 and 0 < mt["auc"] < 1 and 0 < mt["ap"] < 1
    assert mt["sklearn_sample"]["abs_auc_diff"] < 1e-12 and mt["sklearn_sample"]["abs_ap_diff"] < 1e-12
