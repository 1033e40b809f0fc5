#!/bin/bash
# Round 3, GPU call 3: hardware counters of conv_hdma_k (layer-3 shape, batch 256) for the baseline and three ablations
# (LBC_HDMA_DIAG: 15 = MFMA only, 16 = everything but the MFMAs, 2 = no fragment reads) -- clocks (GRBM_GUI_ACTIVE / duration),
# wait / issue-stall shares, LDS activity.  Counters only together with --kernel-trace.
cd "${GRAFT_REPO_ROOT:-/root/repo}"; export TMPDIR=/tmp; mkdir -p gpurun_out; R=gpurun_out
for D in 0 15 16 2 8; do
  i=0
  for ctrs in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE"; do
    i=$((i+1)); rm -rf $R/kp_${D}_$i
    (cd /tmp && LBC_HDMA_CFG=1 LBC_HDMA_DIAG=$D timeout 200 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d "$OLDPWD/$R/kp_${D}_$i" -o k -- python "$OLDPWD/scripts/bench_ops.py" 256 3 fwd l3.conv) > $R/kp_${D}_$i.log 2>&1; echo "diag $D pass $i exit $?"
    python scripts/pmc_kernel.py $(find $R/kp_${D}_$i -name "*counter_collection.csv" | head -1) conv_hdma
    python - "$R/kp_${D}_$i" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)
if f:
    d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) for r in csv.DictReader(open(f[0])) if "conv_hdma" in r["Kernel_Name"]]
    if d: print("   kernel durations (ns): n=%d mean %.0f min %d" % (len(d), sum(d) / len(d), min(d)))
PY
    find $R/kp_${D}_$i -name "*kernel_trace*" -delete
  done
done 2>&1 | tee $R/hdma_pmc.log
