#!/bin/bash
# round 5, GPU call 3: the new tests (refine-input kernels, fused act' gradients, INTEGRATION stub), the Refiner step with them, its kernel trace
cd $GRAFT_REPO_ROOT; O=gpurun_out; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_refiner.py tests/test_gpu_losspath.py::test_integration_stub_functions tests/test_gpu_convstack.py -x -q -m gpu \
   -k "refine or masked or integration or fused_activation or input_activation or refiner" 2>&1 | tail -25 > $O/r5_job3_tests.log
cat $O/r5_job3_tests.log
for i in 1 2; do python bench.py --_other refiner_640x192 2>/dev/null | cut -c1-330; done
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/prof_ref
timeout -k 10 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_ref -- python -u $R/bench.py --_other refiner_640x192 --other_steps 4 > $R/$O/prof_ref.log 2>&1
echo "rocprof rc=$?"
DB=$(find /tmp/prof_ref -name "*_results.db" | head -1)
python $R/scripts/rocprof_summary.py $DB 20 70 k_adam_dev > $R/$O/round5_refiner_kernel_stats.md
head -75 $R/$O/round5_refiner_kernel_stats.md
