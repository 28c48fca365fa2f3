#!/bin/bash
# rocprofv3 --pmc passes (one counter group per pass, --kernel-trace only) over the bench.py conv probe kernel
# (scripts/probe_layer1.py) -> gpurun_out/pmc_probe_wino.json, which carries the sha256 of the kernel source it was measured on
# (bench.py refuses a stale file).  Copy to profiles/roundN_pmc_probe_wino.json.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pp_$i
  timeout -k 10 200 rocprofv3 --pmc $grp --kernel-trace -d /tmp/pp_$i --output-format csv -- python -u $R/scripts/probe_layer1.py 12 > /tmp/pp_$i.log 2>&1
  echo "pass $i rc=$? ($grp)"
done
python - <<PY
import csv, glob, json, hashlib, collections
acc = collections.defaultdict(list); dur = []
for i in (1, 2, 3, 4):
    for f in glob.glob("/tmp/pp_%d/**/*counter_collection.csv" % i, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_conv_wino" in r["Kernel_Name"]]
        byc = collections.defaultdict(list)
        for r in rows:
            byc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for c, v in byc.items():
            acc[c] += v[2:]                       # skip 2 warm-up launches
    for f in glob.glob("/tmp/pp_%d/**/*kernel_trace.csv" % i, recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "k_conv_wino" in r["Kernel_Name"]]
        dur += [(float(r["End_Timestamp"]) - float(r["Start_Timestamp"])) / 1e3 for r in rows][2:]
m = {c: sum(v) / len(v) for c, v in acc.items()}
names = sorted({r["Kernel_Name"].split("(")[0] for i in (1, 2, 3, 4) for f in glob.glob("/tmp/pp_%d/**/*kernel_trace.csv" % i, recursive=True)
                for r in csv.DictReader(open(f)) if "k_conv_wino" in r["Kernel_Name"]})
out = {"kernel": "%s: layer1 3x3 64->64 @48x160 batch 12 (scripts/probe_layer1.py)" % ", ".join(names),
       "collected": "rocprofv3 --pmc <one group per pass> --kernel-trace (scripts/pmc_probe.sh), mean of %d launches after 2 warm-up launches" % len(acc.get("FETCH_SIZE", [])),
       "source_sha256": hashlib.sha256(open("$R/fusiondepth_amd/csrc/conv_wino.hip", "rb").read()).hexdigest(),
       "FETCH_SIZE_KB": m.get("FETCH_SIZE"), "WRITE_SIZE_KB": m.get("WRITE_SIZE"), "fetch_correction": 2.0,
       "traffic_bytes_per_launch": int(m.get("FETCH_SIZE", 0) * 1024 * 2 + m.get("WRITE_SIZE", 0) * 1024),
       "algorithmic_bytes_per_launch": 4 * (12 * 64 * 48 * 160 * 2 + 64 * 64 * 9 * 4),
       "us_per_launch_under_pmc": sum(dur) / max(len(dur), 1)}
out.update({k: v for k, v in m.items() if k not in ("FETCH_SIZE", "WRITE_SIZE")})
json.dump(out, open("$R/gpurun_out/pmc_probe_wino.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
