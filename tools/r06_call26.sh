set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_x; mkdir -p $OUT
cd $R
timeout 1500 python tools/gpu_vs_ref_fuzz.py mapping 0 80 > $OUT/vs_ref_mapping.log 2>&1; tail -2 $OUT/vs_ref_mapping.log
timeout 900 python tools/gpu_vs_ref_fuzz.py loc 0 40 > $OUT/vs_ref_loc.log 2>&1; tail -2 $OUT/vs_ref_loc.log
timeout 900 python tools/gpu_vs_ref_fuzz.py long 0 12 > $OUT/vs_ref_long.log 2>&1; tail -2 $OUT/vs_ref_long.log
timeout 300 python tools/gpu_vs_ref_fuzz.py mapping 319 1 > $OUT/vs_ref_tie.log 2>&1; tail -2 $OUT/vs_ref_tie.log
timeout 300 python tools/gpu_vs_ref_fuzz.py loc 58 1 >> $OUT/vs_ref_tie.log 2>&1; tail -2 $OUT/vs_ref_tie.log
