cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/trace_train_tn -o t -- python3 $GRAFT_REPO_ROOT/tools/train_step_probe.py --compute bf16x6 --steps 3 --warmup 1 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/summarize_profile.py "$(find $GRAFT_REPO_ROOT/gpurun_out/trace_train_tn -name '*kernel_stats.csv' | head -1)" $GRAFT_REPO_ROOT/gpurun_out/train_tn_kernel_stats.csv > /dev/null
find $GRAFT_REPO_ROOT/gpurun_out/trace_train_tn -name '*kernel_trace.csv' -delete; find $GRAFT_REPO_ROOT/gpurun_out/trace_train_tn -name '*.db' -delete
head -12 $GRAFT_REPO_ROOT/gpurun_out/train_tn_kernel_stats.csv | cut -c1-150
