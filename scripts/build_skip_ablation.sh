#!/bin/bash
# libfdhip_skip.so = the whole library with -DFD_SKIP_ABLATION: launches whose "<file>:<line>:<kernel>" site matches $FD_SKIP are dropped
# (what-if timing of the step, results wrong by design).  Run with FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_skip.so FD_SKIP=norm.hip
cd "$(dirname "$0")/.."
O=/tmp/fd_skipobj; mkdir -p $O
pids=()
for s in fusiondepth_amd/csrc/*.hip; do
  b=$(basename $s .hip)
  EXTRA=$(head -1 $s | grep -o "FD_HIPCC_FLAGS:.*" | cut -d: -f2-)
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-pass-failed $EXTRA -DFD_SKIP_ABLATION -I include -I fusiondepth_amd/csrc \
      -I scripts/ubench -c $s -o $O/$b.o &
  pids+=($!)
done
rc=0; for p in "${pids[@]}"; do wait $p || rc=1; done
[ $rc = 0 ] || { echo compile failed; exit 1; }
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o fusiondepth_amd/libfdhip_skip.so $O/*.o && echo built fusiondepth_amd/libfdhip_skip.so
