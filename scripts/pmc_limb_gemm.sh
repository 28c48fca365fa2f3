#!/bin/bash
# PMC passes over scripts/ubench/limb_gemm (split-precision GEMM exploration): where do the limb kernels' cycles go?   (GPU box)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE"; do
  i=$((i+1)); rm -rf /tmp/pl_$i
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pl_$i --output-format csv -- $R/scripts/ubench/limb_gemm > /tmp/pl_$i.log 2>&1
  echo "pass $i rc=$? ($grp)"
  CSV=$(find /tmp/pl_$i -name "*counter_collection.csv" | head -1)
  python - "$CSV" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"]
    if "k_gemm" not in k: continue
    if row.get("Grid_Size", "") and "4096" not in row.get("Grid_Size", "4096"): pass
    acc[k[:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k, d in acc.items():
    for c, v in sorted(d.items()):
        v = v[: len(v) // 2] if len(v) > 2 else v        # first shape (4096^3) only: its launches come first
        print("  %-60s %-26s mean %.6g (n=%d)" % (k, c, sum(v) / len(v), len(v)))
PY
done
