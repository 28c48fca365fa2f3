cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_ksweep_ablate2.log; : > $L
for v in "" woob wnolds woob_nolds wl1 wnoepi wnostore; do
  echo "=== variant '$v'" >> $L
  if [ -n "$v" ]; then export FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_$v.so; else unset FD_LIBFDHIP; fi
  timeout 200 python scripts/wino_ksweep.py 8 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
  timeout 200 python scripts/wino_ksweep.py 24 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
done
cat $L
