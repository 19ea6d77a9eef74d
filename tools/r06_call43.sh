set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_az}; mkdir -p $OUT
cd $R
timeout 60 tools/experiments/bin/lone_wave_issue_rate > $OUT/lone_wave_issue_rate.log 2>&1; cat $OUT/lone_wave_issue_rate.log
timeout 400 python tools/gpu_vs_ref_fuzz.py mapping 0 404 > $OUT/gpu_vs_ref_mapping_0_403.log 2>&1; tail -1 $OUT/gpu_vs_ref_mapping_0_403.log | cut -c1-400
timeout 120 python tools/gpu_vs_ref_fuzz.py loc 0 80 > $OUT/gpu_vs_ref_loc_0_79.log 2>&1; tail -1 $OUT/gpu_vs_ref_loc_0_79.log | cut -c1-300
timeout 120 python tools/gpu_vs_ref_fuzz.py long 0 60 > $OUT/gpu_vs_ref_long_0_59.log 2>&1; tail -1 $OUT/gpu_vs_ref_long_0_59.log | cut -c1-300
