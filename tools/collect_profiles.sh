#!/bin/bash
# Round-2 measurement set, run on the GPU box from the repo root:  bash tools/collect_profiles.sh
# Writes everything under gpurun_out/r02/; the summaries worth keeping are copied to profiles/ by hand afterwards.
set -u
R="$PWD"
O="$R/gpurun_out/r02"
mkdir -p "$O"
cd /tmp && export TMPDIR=/tmp
B="$R/bench.py"
QUIET="--no-extra-modes --no-ucf-eval --no-cpu-baseline"

echo "[1] default bench line"; python3 "$B" > "$O/bench_default.json" 2> "$O/bench_default.err"; echo "rc=$?"

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
K="iefvad_gemm_bf16_pipe_kernel|iefvad_gemm_bf16_w256_kernel|iefvad_heads_fused_bf16_kernel|iefvad_outproj_ln_bf16_kernel"
python3 "$R/tools/hbm_traffic.py" "$(find "$O/pmc_fetch_bf16" -name '*counter_collection.csv' | head -1)" "$(find "$O/pmc_write_bf16" -name '*counter_collection.csv' | head -1)" \
    "$K" 262144 "$O/gemm_bf16_hbm_traffic.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on python3 bench.py --steps 1 --warmup 1 --chunks 1024 --compute bf16 (one micro-batch of 1024 chunks = 262144 rows per launch), MI355X, round 2" > /dev/null 2>&1
K=iefvad_gemm_split_n128_kernel
python3 "$R/tools/hbm_traffic.py" "$(find "$O/pmc_fetch_bf16x6" -name '*counter_collection.csv' | head -1)" "$(find "$O/pmc_write_bf16x6" -name '*counter_collection.csv' | head -1)" \
    $K 262144 "$O/gemm_split_hbm_traffic.json" "rocprofv3 --kernel-trace --pmc FETCH_SIZE / --pmc WRITE_SIZE (two separate passes) on python3 bench.py --steps 1 --warmup 1 --chunks 1024 --compute bf16x6 (262144 rows per launch), MI355X, round 2" > /dev/null 2>&1

echo "[4] per-video pattern"
python3 "$R/tools/latency_probe.py" > "$O/latency.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_b1" -o t -- python3 "$R/tools/b1_loop.py" > /dev/null 2> "$O/trace_b1.log"
python3 "$R/tools/summarize_profile.py" "$(find "$O/trace_b1" -name '*kernel_stats.csv' | head -1)" "$O/b1_kernel_stats.csv" > /dev/null
echo "[5] BASELINE config 5 (K=5, Shang+MSAD-sized, bf16): kernel stats + PMC counters"
rocprofv3 --kernel-trace --stats --output-format csv -d "$O/trace_cfg5" -o t -- python3 "$R/tools/config5_profile.py" 5 > "$O/config5_run.log" 2> "$O/trace_cfg5.log"
python3 "$R/tools/summarize_profile.py" "$(find "$O/trace_cfg5" -name '*kernel_stats.csv' | head -1)" "$O/config5_kernel_stats.csv" > /dev/null
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --output-format csv \
    -d "$O/pmc_sq_cfg5" -o p -- python3 "$R/tools/config5_profile.py" 2 > /dev/null 2> "$O/pmc_sq_cfg5.log"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$O/pmc_fetch_cfg5" -o p -- python3 "$R/tools/config5_profile.py" 2 > /dev/null 2> "$O/pmc_fetch_cfg5.log"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$O/pmc_write_cfg5" -o p -- python3 "$R/tools/config5_profile.py" 2 > /dev/null 2> "$O/pmc_write_cfg5.log"
python3 "$R/tools/pmc_summary.py" "$O/pmc_sq_cfg5" "$O/pmc_fetch_cfg5" "$O/pmc_write_cfg5" > "$O/pmc_summary_config5.txt" 2>&1
# keep the merge-back small: the raw traces are large
find "$O" -name '*kernel_trace.csv' -size +8M -delete
find "$O" -name '*.db' -delete
du -sh "$O"
echo done
