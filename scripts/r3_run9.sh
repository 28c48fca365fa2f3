cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
L=gpurun_out/r3_ksweep_ablate4.log; : > $L
for v in wnou wnov wnou_oob wnov_oob; do
  echo "=== variant '$v'" >> $L
  export FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_$v.so
  timeout 200 python scripts/wino_ksweep.py 24 2>&1 | grep -v amdgpu.ids | tail -3 >> $L
done
cat $L
