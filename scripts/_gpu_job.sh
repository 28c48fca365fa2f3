bash $GRAFT_REPO_ROOT/scripts/round3_profiles.sh d
