cd "$GRAFT_REPO_ROOT"
bash tools/r02_round.sh v21 && bash tools/profile_round.sh v21
