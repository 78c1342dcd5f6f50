#!/bin/bash
# Measurement set of a round, run on the GPU box from the repo root:  bash tools/collect_profiles.sh [r03] [head]
# Writes everything under gpurun_out/<round>/; the summaries worth keeping are copied to profiles/<round>_* afterwards
# (tools/keep_profiles.sh).  `head` = the commit the box's snapshot was taken from (there is no .git on the box).
set -u
RND="${1:-r05}"
HEAD="${2:-unknown}"
export IEFVAD_HEAD="$HEAD"
R="$PWD"
O="$R/gpurun_out/$RND"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
QUIET="--no-extra-modes --no-ucf-eval --no-cpu-baseline"

echo "[1] default bench line"; T0=$(date +%s); python3 "$B" > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "rc=$? wall=$(( $(date +%s) - T0 )) s"

for mode in bf16x6 bf16 f32; do
  echo "[2] kernel trace, $mode"
  rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_$mode" -o t -- python3 "$B" --steps 3 --warmup 1 --compute $mode $QUIET \
      > "$O/bench_${mode}_under_rocprof.json" 2> "$O/trace_$mode.log"
  python3 "$R/tools/summarize_profile.py" "$(find "$O/trace_$mode" -name '*kernel_stats.csv' | head -1)" "$O/bench_${mode}_kernel_stats.csv" > /dev/null
done

# PMC passes (each in its own run, --kernel-trace only beside --pmc): one 1024-chunk micro-batch per launch
for mode in bf16 bf16x6; do
  echo "[3] PMC, $mode"
  rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv \
      -d "$O/pmc_sq_$mode" -o p -- python3 "$B" --steps 1 --warmup 1 --chunks 1024 --compute $mode $QUIET > /dev/null 2> "$O/pmc_sq_$mode.log"
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv \
      -d "$O/pmc_fetch_$mode" -o p -- python3 "$B" --steps 1 --warmup 1 --chunks 1024 --compute $mode $QUIET > /dev/null 2> "$O/pmc_fetch_$mode.log"
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv \
      -d "$O/pmc_write_$mode" -o p -- python3 "$B" --steps 1 --warmup 1 --chunks 1024 --compute $mode $QUIET > /dev/null 2> "$O/pmc_write_$mode.log"
  python3 "$R/tools/pmc_summary.py" "$O/pmc_sq_$mode" "$O/pmc_fetch_$mode" "$O/pmc_write_$mode" > "$O/pmc_summary_$mode.txt" 2>&1
done
SRC="rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on python3 bench.py --steps 1 --warmup 1 --chunks 1024"
K="iefvad_inproj_chain_f32in_kernel|iefvad_inproj_chain_bf16_kernel|iefvad_refine_chain_bf16_kernel|iefvad_heads_pchain_bf16_kernel|iefvad_outproj_ln_pchain_bf16_kernel"
# algorithmic bytes of the six projection launches of a 262,144-row pass (bf16 mode, outputs=scores), KB per row, both modalities:
# in_proj x2: read 6 (fp32 rows, layer 0) / 3 (bf16 rows, layer 1), write 9 (bf16 q|k|v); out_proj+LN x2: read 3 + 6 (fp32 residual),
# write 6 fp32 + 3 bf16 (layer 0) / 3 bf16 (layer 1); heads+fusion: read 3, write 3 (fp32 z) + 0.2; refinement chain: read 3 (z), write 0.004
python3 "$R/tools/hbm_traffic.py" "$(find "$O/pmc_fetch_bf16" -name '*counter_collection.csv' | head -1)" "$(find "$O/pmc_write_bf16" -name '*counter_collection.csv' | head -1)" \
    "$K" 262144 "$O/gemm_bf16_hbm_traffic.json" "$SRC --compute bf16 (one micro-batch of 1024 chunks = 262144 rows per launch), MI355X, $RND, head $HEAD" \
    $(python3 -c "print(262144*1024*(6+3+3+6+3+6+3+3)/6, 262144*1024*(9+9+9+3+3.2+0.004)/6)") \
    "mean over the 6 projection launches of one pass: 2 x in_proj (6 / 3 KB/row read, 9 written), 2 x out_proj+LayerNorm (9 read; 9 / 3 written), heads+fusion (3 / 3.2), refinement chain of 2K projections + scorer (3 / 0.004); weights <= 24 MB per launch" > /dev/null 2>&1
K=iefvad_gemm_split_n128_kernel
python3 "$R/tools/hbm_traffic.py" "$(find "$O/pmc_fetch_bf16x6" -name '*counter_collection.csv' | head -1)" "$(find "$O/pmc_write_bf16x6" -name '*counter_collection.csv' | head -1)" \
    $K 262144 "$O/gemm_split_hbm_traffic.json" "$SRC --compute bf16x6 (262144 rows per launch), MI355X, $RND, head $HEAD" \
    1432400000 1288400000 "mean over the 25 GEMM launches of one 262144-row pass: A (fp32) + residual reads and C writes; W is 3 bf16 planes (<= 10.6 MB per launch)" > /dev/null 2>&1

echo "[4] split GEMM: random vs all-zero operands (the power-envelope probe), with the in-kernel clock"
for z in 0 1; do
  if [ $z = 1 ]; then export GS_ZERO=1; else unset GS_ZERO; fi
  "$R/tools/gemm_tune_split" > "$O/gemm_split_tune_zero$z.log" 2>&1
  "$R/tools/gemm_tune_split_clk" > "$O/gemm_split_clk_zero$z.log" 2>&1
done
unset GS_ZERO

echo "[5] per-video pattern"
python3 "$R/tools/latency_probe.py" > "$O/latency.log" 2>&1
echo "[6] BASELINE config 5 (K=5, Shang+MSAD lists, bf16): kernel stats + PMC counters"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_cfg5" -o t -- python3 "$R/tools/config5_profile.py" 5 > "$O/config5_run.log" 2> "$O/trace_cfg5.log"
python3 "$R/tools/summarize_profile.py" "$(find "$O/trace_cfg5" -name '*kernel_stats.csv' | head -1)" "$O/config5_kernel_stats.csv" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv \
    -d "$O/pmc_sq_cfg5" -o p -- python3 "$R/tools/config5_profile.py" 2 > /dev/null 2> "$O/pmc_sq_cfg5.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_cfg5" -o p -- python3 "$R/tools/config5_profile.py" 2 > /dev/null 2> "$O/pmc_fetch_cfg5.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_cfg5" -o p -- python3 "$R/tools/config5_profile.py" 2 > /dev/null 2> "$O/pmc_write_cfg5.log"
python3 "$R/tools/pmc_summary.py" "$O/pmc_sq_cfg5" "$O/pmc_fetch_cfg5" "$O/pmc_write_cfg5" > "$O/pmc_summary_config5.txt" 2>&1
echo "[7] packed evaluation loop: where the wall clock goes"
python3 "$R/tools/ragged_profile.py" > "$O/ragged_profile.log" 2>&1
echo "[8] training step (ucf_train.py's configuration, B = 128): kernel stats"
python3 "$R/tools/train_step_probe.py" --compute bf16x6 > "$O/train_step.json" 2> "$O/train_step.err"
python3 "$R/tools/train_step_probe.py" --compute f32 >> "$O/train_step.json" 2>> "$O/train_step.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_train" -o t -- python3 "$R/tools/train_step_probe.py" --compute bf16x6 --steps 3 --warmup 1 \
    > "$O/train_step_under_rocprof.json" 2> "$O/trace_train.log"
python3 "$R/tools/summarize_profile.py" "$(find "$O/trace_train" -name '*kernel_stats.csv' | head -1)" "$O/train_step_kernel_stats.csv" > /dev/null
echo "[9] the evaluation list walked inside the library: phases of configs 3 and 5"
IEFVAD_HOSTPIPE_TRACE=1 python3 "$R/tools/host_list_probe.py" > "$O/host_list_probe.log" 2> "$O/host_list_trace.log"
grep hostpipe "$O/host_list_trace.log" | tail -16 >> "$O/host_list_probe.log"
echo "== wire_dtype = BF16 (fp32 host rows rounded to bf16 by the staging threads)" >> "$O/host_list_probe.log"
IEFVAD_HOSTPIPE_TRACE=1 python3 "$R/tools/host_list_probe.py" --wire-bf16 >> "$O/host_list_probe.log" 2> "$O/host_list_trace_wire.log"
grep hostpipe "$O/host_list_trace_wire.log" | tail -16 >> "$O/host_list_probe.log"
echo "[10] persistent out_proj + LayerNorm kernel: phase stamps (diag build)"
# the diag library is rebuilt from THIS tree (a stale one lacks the newer exports and lib.py refuses it)
mkdir -p "$R/build" && make -s -B -C "$R/ief-vad_amd/csrc" EXTRA=-DOC_DIAG OUT=../../build/libiefvad_ocdiag.so > "$O/ocdiag_build.log" 2>&1
if [ -f "$R/build/libiefvad_ocdiag.so" ]; then IEFVAD_LIB="$R/build/libiefvad_ocdiag.so" python3 "$R/tools/outproj_pdiag.py" > "$O/outproj_pchain_phase_stamps.log" 2>&1; fi
for pz in 1 0; do
  IEFVAD_PERSIST=$pz python3 "$B" --compute bf16 --steps 3 --warmup 1 $QUIET 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read()); print('IEFVAD_PERSIST=$pz', round(d['value']), 'snippets/s', round(d['ms_per_step'],2), 'ms/step', {k:round(v,2) for k,v in d['stage_ms_per_step'].items()})" >> "$O/persist_ab.log"
done
echo "[11] metric tail (iefvad_auc_ap at config 4's size): kernel stats"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_metric" -o t -- python3 "$R/tools/metric_probe.py" > "$O/metric_tail.log" 2> "$O/trace_metric.log"
python3 "$R/tools/summarize_profile.py" "$(find "$O/trace_metric" -name '*kernel_stats.csv' | head -1)" "$O/metric_tail_kernel_stats.csv" > /dev/null
# keep the merge-back small: the raw traces are large
find "$O" -name '*kernel_trace.csv' -size +8M -delete
find "$O" -name '*.db' -delete
du -sh "$O"
echo done
