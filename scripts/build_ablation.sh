#!/bin/bash
# Build libfdhip_<tag>.so = libfdhip.so with photometric_ms.hip recompiled with extra -D flags (timing experiments).
#   scripts/build_ablation.sh nobar -DFD_MS_ABLATE=1        then run with FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_nobar.so
TAG=$1; shift
cd "$(dirname "$0")/.."
O=fusiondepth_amd/csrc/_obj
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-pass-failed -fno-slp-vectorize -I include -I fusiondepth_amd/csrc "$@" \
    -c fusiondepth_amd/csrc/photometric_ms.hip -o /tmp/pm_$TAG.o || exit 1
OBJS=$(ls $O/*.o | grep -v photometric_ms.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o fusiondepth_amd/libfdhip_$TAG.so $OBJS /tmp/pm_$TAG.o && echo built fusiondepth_amd/libfdhip_$TAG.so
