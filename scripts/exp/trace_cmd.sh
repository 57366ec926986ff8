cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/ktrace -- python3 $GRAFT_REPO_ROOT/scripts/gpu_probe.py trace > $GRAFT_REPO_ROOT/gpurun_out/ktrace.log 2>&1
echo done
