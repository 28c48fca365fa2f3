#!/bin/bash
# Build libfdhip_<tag>.so = libfdhip.so with ONE source recompiled with extra -D flags (timing experiments).
#   scripts/build_ablation.sh nobar photometric_ms -DFD_MS_ABLATE=1      then run with FD_LIBFDHIP=$PWD/fusiondepth_amd/libfdhip_nobar.so
TAG=$1; SRC=$2; shift 2
cd "$(dirname "$0")/.."
O=fusiondepth_amd/csrc/_obj
EXTRA=$(head -1 fusiondepth_amd/csrc/$SRC.hip | grep -o "FD_HIPCC_FLAGS:.*" | cut -d: -f2-)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-comment -Wno-pass-failed $EXTRA -I include -I fusiondepth_amd/csrc "$@" \
    -c fusiondepth_amd/csrc/$SRC.hip -o /tmp/abl_$TAG.o || exit 1
OBJS=$(ls $O/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o fusiondepth_amd/libfdhip_$TAG.so $OBJS /tmp/abl_$TAG.o && echo built fusiondepth_amd/libfdhip_$TAG.so
