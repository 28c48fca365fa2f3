#!/bin/bash
# the slice / split targets re-swept on the final build (the limb kernels changed what a workgroup costs)
cd $GRAFT_REPO_ROOT
run() { env "$@" timeout 300 python scripts/secondary_ab.py r18 5 20 2>/dev/null | tail -1 | cut -c1-60,140-200; }
run FD_NONE=1
run FD_WINO_WGRAD_TARGET=384
run FD_WINO_WGRAD_TARGET=512
run FD_LIMB_WGRAD_TARGET=128
run FD_LIMB_WGRAD_TARGET=512
run FD_NONE=1
run FD_LIMB_TARGET=128
run FD_LIMB_TARGET=512
run FD_WINO_TARGET=256
run FD_WINO_TARGET=512
run FD_NONE=1
