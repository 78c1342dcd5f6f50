#!/bin/bash
# Copy the summaries of a collect_profiles.sh run (gpurun_out/<round>/) into profiles/<round>_* (tracked).
set -eu
RND="${1:-r05}"
O=gpurun_out/$RND
cp $O/bench_default.json profiles/${RND}_bench_default.json.log
for m in bf16x6 bf16 f32; do
  cp $O/bench_${m}_kernel_stats.csv profiles/${RND}_bench_${m}_kernel_stats.csv
  cp $O/bench_${m}_under_rocprof.json profiles/${RND}_bench_${m}_under_rocprof.json.log
done
cp $O/pmc_summary_bf16.txt profiles/${RND}_kernel_pmc_summary_bf16.txt
cp $O/pmc_summary_bf16x6.txt profiles/${RND}_kernel_pmc_summary_bf16x6.txt
cp $O/pmc_summary_config5.txt profiles/${RND}_kernel_pmc_summary_config5_k5_bf16.txt
cp $O/config5_kernel_stats.csv profiles/${RND}_config5_k5_bf16_kernel_stats.csv
cp $O/config5_run.log profiles/${RND}_config5_run.log
cp $O/gemm_split_hbm_traffic.json profiles/${RND}_gemm_split_hbm_traffic.json
cp $O/gemm_bf16_hbm_traffic.json profiles/${RND}_gemm_bf16_hbm_traffic.json
(echo "== random operands: tools/gemm_tune_split (M = 65536, K = 768)"; cat $O/gemm_split_tune_zero0.log; echo
 echo "== random operands: tools/gemm_tune_split_clk (in-kernel stamps, GB2_CLOCK_DIAG)"; cat $O/gemm_split_clk_zero0.log; echo
 echo "== GS_ZERO=1 (all-zero operands: no multiplier bits toggle): tools/gemm_tune_split"; cat $O/gemm_split_tune_zero1.log; echo
 echo "== GS_ZERO=1: tools/gemm_tune_split_clk"; cat $O/gemm_split_clk_zero1.log) > profiles/${RND}_gemm_split_zero_vs_random_operands.log
cp $O/latency.log profiles/${RND}_small_batch_latency.log
cp $O/ragged_profile.log profiles/${RND}_packed_eval_loop_profile.log
cp $O/train_step.json profiles/${RND}_train_step.json.log
cp $O/train_step_kernel_stats.csv profiles/${RND}_train_step_kernel_stats.csv
cp $O/host_list_probe.log profiles/${RND}_host_list_probe.log
[ -f $O/outproj_pchain_phase_stamps.log ] && cp $O/outproj_pchain_phase_stamps.log profiles/${RND}_outproj_pchain_phase_stamps.log
cp $O/persist_ab.log profiles/${RND}_persistent_rowblock_ab.log
[ -f $O/metric_tail.log ] && cp $O/metric_tail.log profiles/${RND}_metric_tail.log
[ -f $O/metric_tail_kernel_stats.csv ] && cp $O/metric_tail_kernel_stats.csv profiles/${RND}_metric_tail_kernel_stats.csv
echo kept
