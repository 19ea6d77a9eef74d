set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_t; mkdir -p $OUT
cd $R
for cut in 3 4 5; do for tie in 0 1; do echo "== cut $cut tie $tie"; timeout 200 python tools/dbg_scenario2.py lfuzz58 $cut $tie 2>&1 | tail -14; done; done > $OUT/dbg.log 2>&1
cat $OUT/dbg.log
