cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_ksweep_ablate.log; : > $L
for v in "" wnoloop wnostore wnoepi wnoproload wnoloop_noepi wnothing; do
  echo "=== variant '$v'" >> $L
  if [ -n "$v" ]; then export FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_$v.so; else unset FD_LIBFDHIP; fi
  timeout 200 python scripts/wino_ksweep.py 8 2>&1 | grep -v amdgpu.ids >> $L
done
unset FD_LIBFDHIP
cat $L
timeout 200 scripts/ubench/mfma_ablate2 rand > gpurun_out/r3_ablate2_rand.log 2>&1; cat gpurun_out/r3_ablate2_rand.log
