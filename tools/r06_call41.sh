set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r06_ax}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
bash tools/final_round6.sh $TAG
bash tools/prof_round6.sh $TAG > $OUT/prof.log 2>&1; tail -3 $OUT/prof.log
