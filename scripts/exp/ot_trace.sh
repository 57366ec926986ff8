cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ottrace -- python3 $GRAFT_REPO_ROOT/scripts/gpu_ot_probe.py > $GRAFT_REPO_ROOT/gpurun_out/ottrace.log 2>&1
cat $GRAFT_REPO_ROOT/gpurun_out/ottrace.log | tail -4
