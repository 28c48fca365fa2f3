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
  PROBE_TIMING=0 timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pmc_$i --output-format csv -- python -u $R/scripts/probe_loss_ms.py $B $ROWS 6 > /tmp/pmc_$i.log 2>&1
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
# machine-readable traffic record for bench.py (carries the sha256 of the kernel source it was measured on)
python - <<PY
import csv, glob, json, hashlib, collections
tot = collections.defaultdict(lambda: collections.defaultdict(list))
for i, name in ((3, "FETCH_SIZE"), (4, "WRITE_SIZE")):
    for f in glob.glob("/tmp/pmc_%d/**/*counter_collection.csv" % i, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_photo_ms" in k and r["Counter_Name"] == name:
                kk = "k_photo_ms_bwd" if "k_photo_ms_bwd" in k else ("k_photo_ms_fin" if "k_photo_ms_fin" in k else "k_photo_ms")
                tot[kk][name].append(float(r["Counter_Value"]))
mean = lambda v: sum(v) / max(len(v), 1)
per = {k: {"FETCH_SIZE_KB": mean(d["FETCH_SIZE"]), "WRITE_SIZE_KB": mean(d["WRITE_SIZE"])} for k, d in tot.items()}
traffic = sum(int(p["FETCH_SIZE_KB"] * 1024 * 2 + p["WRITE_SIZE_KB"] * 1024) for p in per.values())
out = {"kernels": per, "fetch_correction": 2.0, "batch": $B, "traffic_bytes_per_launch": traffic,
       "collected": "rocprofv3 --pmc <one group per pass> --kernel-trace (scripts/pmc_loss_ms.sh), means over the probe's launches",
       "source_sha256": hashlib.sha256(open("$R/fusiondepth_amd/csrc/photometric_ms.hip", "rb").read()).hexdigest()}
json.dump(out, open("$R/gpurun_out/pmc_loss.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
