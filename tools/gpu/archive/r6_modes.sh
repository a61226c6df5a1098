# usage: r6_modes.sh "<modes>": config 5 with the experiments variant in the given SMST_DEBUG_MODEs
cd $GRAFT_REPO_ROOT
MODES="$1" bash tools/gpu/r5_ablate_vocn.sh
