cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq1 -- python $GRAFT_REPO_ROOT/scripts/gpu_probe.py big > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq1.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2 -- python $GRAFT_REPO_ROOT/scripts/gpu_probe.py big > $GRAFT_REPO_ROOT/gpurun_out/pmc_sq2.log 2>&1
echo done
