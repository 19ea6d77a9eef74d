set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_r; mkdir -p $OUT
cd $R
timeout 300 python tools/dbg_scenario.py fuzz319 > $OUT/dbg_fuzz319.log 2>&1; cat $OUT/dbg_fuzz319.log
timeout 300 python tools/dbg_scenario.py lfuzz58 > $OUT/dbg_lfuzz58.log 2>&1; cat $OUT/dbg_lfuzz58.log
