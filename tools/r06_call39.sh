set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r06_av}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 120 tools/experiments/bin/lone_wave_issue_rate > $OUT/lone_wave_issue_rate.log 2>&1; cat $OUT/lone_wave_issue_rate.log
timeout 900 python tools/gpu_vs_ref_fuzz.py mapping 0 404 > $OUT/gpu_vs_ref_mapping_0_403.log 2>&1; tail -1 $OUT/gpu_vs_ref_mapping_0_403.log | cut -c1-300
timeout 600 python tools/gpu_vs_ref_fuzz.py loc 0 80 > $OUT/gpu_vs_ref_loc_0_79.log 2>&1; tail -1 $OUT/gpu_vs_ref_loc_0_79.log | cut -c1-300
timeout 900 python tools/gpu_vs_ref_fuzz.py long 0 60 > $OUT/gpu_vs_ref_long_0_59.log 2>&1; tail -1 $OUT/gpu_vs_ref_long_0_59.log | cut -c1-300
bash tools/final_round6.sh $TAG
bash tools/prof_round6.sh $TAG > $OUT/prof.log 2>&1; tail -5 $OUT/prof.log
