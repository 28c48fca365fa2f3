cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
timeout -k 10 900 bash scripts/round3_profiles.sh g 2>&1 | grep -v "^-rw" | tail -8
