#!/bin/bash
# k_conv_wino2d_limb: more step pairs, other configurations
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  FD_WINO_FWD_LIMB=0 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
  FD_WINO_FWD_LIMB=1 timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-200
done
for i in 1 2; do
  FD_WINO_FWD_LIMB=0 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
  FD_WINO_FWD_LIMB=1 timeout 300 python scripts/secondary_ab.py r50 3 10 2>/dev/null | tail -1 | cut -c1-200
done
FD_WINO_FWD_LIMB=0 timeout 300 python scripts/secondary_ab.py r18big 3 10 2>/dev/null | tail -1 | cut -c1-200
FD_WINO_FWD_LIMB=1 timeout 300 python scripts/secondary_ab.py r18big 3 10 2>/dev/null | tail -1 | cut -c1-200
for i in 1 2; do
  FD_WINO_FWD_LIMB=0 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-60
  FD_WINO_FWD_LIMB=1 timeout 300 python bench.py --_other refiner_640x192 2>/dev/null | tail -1 | cut -c1-60
done
