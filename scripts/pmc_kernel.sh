#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over any probe command, summarised per kernel whose
# name contains <match>:   scripts/pmc_kernel.sh <tag> <match> <skip-launches> -- <command...>   -> gpurun_out/<tag>_pmc.md
TAG=$1; MATCH=$2; SKIP=$3; shift 4
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "GRBM_GUI_ACTIVE FETCH_SIZE" "GRBM_GUI_ACTIVE WRITE_SIZE" \
           "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
           "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS" \
           "SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1)); rm -rf /tmp/pk_${TAG}_$i
  timeout -k 10 300 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pk_${TAG}_$i --output-format csv -- "$@" > /tmp/pk_${TAG}_$i.log 2>&1
  echo "pass $i rc=$? ($grp)"
done
python - "$TAG" "$MATCH" "$SKIP" "$*" > $R/gpurun_out/${TAG}_pmc.md <<'PY'
import csv, glob, re, sys, collections
tag, match, skip, cmd = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"\((?:[^()]|\([^()]*\))*\)$", "", n)
    return re.sub(r"^void ", "", n)[:60]
acc = collections.defaultdict(lambda: collections.defaultdict(list)); dur = collections.defaultdict(list)
for i in range(1, 6):
    for f in glob.glob("/tmp/pk_%s_%d/**/*counter_collection.csv" % (tag, i), recursive=True):
        per = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in per.items():
            for c, v in d.items():
                acc[k][c] = v[skip:]
    for f in glob.glob("/tmp/pk_%s_%d/**/*kernel_trace.csv" % (tag, i), recursive=True):
        per = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if match in r["Kernel_Name"]:
                per[short(r["Kernel_Name"])].append((float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3)
        for k, v in per.items():
            dur[k] += v[skip:]
print("# PMC counters of kernels matching `%s`\n\ncommand: `%s`; rocprofv3 --pmc, one group per pass, --kernel-trace only (scripts/pmc_kernel.sh); "
      "means per launch after skipping %d warm-up launches.\n" % (match, cmd, skip))
for k, d in acc.items():
    m = {c: sum(v) / max(len(v), 1) for c, v in d.items()}
    us = sum(dur[k]) / max(len(dur[k]), 1)
    print("## `%s`  (%d launches, %.1f us per launch under the counters)\n" % (k, len(dur[k]), us))
    print("| counter | mean per launch |\n|---|---:|")
    for c in sorted(m):
        print("| %s | %.6g |" % (c, m[c]))
    g = m.get
    print()
    if g("FETCH_SIZE") is not None:
        print("- HBM traffic: FETCH_SIZE %.1f KB x 2 (gfx950 correction, MI355X_MICROARCH.md) + WRITE_SIZE %.1f KB = %.2f MB per launch"
              % (g("FETCH_SIZE"), g("WRITE_SIZE", 0), (2 * g("FETCH_SIZE") + g("WRITE_SIZE", 0)) / 1024))
    if g("SQ_INSTS_MFMA") and g("SQ_INSTS_VALU"):
        print("- VALU : MFMA instructions = %.2f : 1 (SQ_INSTS_VALU includes the MFMA instructions)" % ((g("SQ_INSTS_VALU") - g("SQ_INSTS_MFMA")) / g("SQ_INSTS_MFMA")))
    if g("SQ_VALU_MFMA_BUSY_CYCLES") and g("SQ_BUSY_CU_CYCLES"):
        print("- matrix pipe busy = SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CU_CYCLES = %.3f" % (g("SQ_VALU_MFMA_BUSY_CYCLES") / g("SQ_BUSY_CU_CYCLES")))
    if g("SQ_ACTIVE_INST_VALU") and g("SQ_WAVE_CYCLES"):
        print("- per wave: VALU issue %.3f, LDS %.3f, VMEM %.3f, scalar %.3f of the wave cycles; waiting (any) %.3f" % tuple(
            g(x, 0) / g("SQ_WAVE_CYCLES") for x in ("SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_VMEM", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_ANY")))
    if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None:
        print("- L2 hit rate %.3f" % (g("TCC_HIT_sum") / max(g("TCC_HIT_sum") + g("TCC_MISS_sum"), 1)))
    print()
PY
cat $R/gpurun_out/${TAG}_pmc.md
