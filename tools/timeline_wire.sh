cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/tl_wire -o t -- python3 $GRAFT_REPO_ROOT/tools/host_list_probe.py --wire-bf16 > /dev/null 2>&1
python3 $GRAFT_REPO_ROOT/tools/timeline.py $GRAFT_REPO_ROOT/gpurun_out/tl_wire 40 > $GRAFT_REPO_ROOT/gpurun_out/tl_wire.txt
find $GRAFT_REPO_ROOT/gpurun_out/tl_wire -name '*.csv' -size +4M -delete; find $GRAFT_REPO_ROOT/gpurun_out/tl_wire -name '*.db' -delete
wc -l $GRAFT_REPO_ROOT/gpurun_out/tl_wire.txt
