cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/c5trace -- python3 $GRAFT_REPO_ROOT/tests/tools/gpu_c5_sweep.py --check 0 > $GRAFT_REPO_ROOT/gpurun_out/c5trace.log 2>&1
tail -1 $GRAFT_REPO_ROOT/gpurun_out/c5trace.log
