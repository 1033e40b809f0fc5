#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
ctrs="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE"
rm -rf $R/kp3
(cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/$R/kp3" -o k -- python "$OLDPWD/scripts/bench_ops.py" 256 3 wgrad ${LAYER:-l3.conv}) > $R/kp3.log 2>&1; echo "pass exit $?"
python scripts/pmc_kernel.py $(find $R/kp3 -name "*counter_collection.csv" | head -1) wgrad
find $R/kp3 -name "*kernel_trace*" -delete
ctrs="SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"
rm -rf $R/kp4
(cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/$R/kp4" -o k -- python "$OLDPWD/scripts/bench_ops.py" 256 3 wgrad ${LAYER:-l3.conv}) > $R/kp4.log 2>&1; echo "pass exit $?"
python scripts/pmc_kernel.py $(find $R/kp4 -name "*counter_collection.csv" | head -1) wgrad
find $R/kp4 -name "*kernel_trace*" -delete
