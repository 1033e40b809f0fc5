#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
i=0
for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_LDS_ADDR_CONFLICT GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf $R/kp$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/$R/kp$i" -o k -- python "$OLDPWD/scripts/bench_ops.py" 256 3 fwd ${LAYER:-l3.conv}) > $R/kp$i.log 2>&1; echo "pass $i exit $?"
  python scripts/pmc_kernel.py $(find $R/kp$i -name "*counter_collection.csv" | head -1) conv_glds
  find $R/kp$i -name "*kernel_trace*" -delete
done
