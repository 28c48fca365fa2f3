#!/bin/bash
# rocprofv3 --pmc passes over the multi-scale loss probe (one counter group per pass) -> gpurun_out/<tag>_pmc.txt
TAG=${1:-ms}; B=${2:-12}; ROWS=${3:-0}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/${TAG}_pmc.txt; : > $OUT
i=0
for grp in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_VALU_MFMA_BUSY_CYCLES TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pmc_$i
  PROBE_TIMING=0 timeout 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i --output-format csv -- python -u $R/scripts/probe_loss_ms.py $B $ROWS 6 > /tmp/pmc_$i.log 2>&1
  echo "pass $i rc=$? ($grp)" >> $OUT
  CSV=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  python - "$CSV" >> $OUT <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "k_photo_ms" not in k: continue
    acc[k[:40]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    for c, v in sorted(d.items()):
        print("  %-40s %-28s mean %.6g  (n=%d)" % (k, c, sum(v) / len(v), len(v)))
PY
done
cat $OUT
