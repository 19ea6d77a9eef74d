set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_v; mkdir -p $OUT
cd $R
for cut in 1 2 3; do echo "== cut $cut"; timeout 200 python tools/dbg_scenario2.py fuzz319 $cut 0 2 2>&1 | tail -12; done > $OUT/dbg.log 2>&1
cat $OUT/dbg.log
