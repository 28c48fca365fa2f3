cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout 300 python scripts/wino_ksweep.py 12 > gpurun_out/r3_ksweep.log 2>&1
timeout 300 python scripts/wino_ksweep.py 24 >> gpurun_out/r3_ksweep.log 2>&1
timeout 300 python scripts/wino_ksweep.py 8 >> gpurun_out/r3_ksweep.log 2>&1
cat gpurun_out/r3_ksweep.log
python -c "
import json; r=json.load(open('gpurun_out/r3_bench3.json')) if False else None"
