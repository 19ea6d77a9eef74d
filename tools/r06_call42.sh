set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_ay}; mkdir -p $OUT
cd $R
timeout 700 python tools/gpu_fuzz_replay.py mapping 0 200 > $OUT/fuzz_mapping_0_199.log 2>&1; tail -1 $OUT/fuzz_mapping_0_199.log | cut -c1-300
timeout 300 python tools/gpu_fuzz_replay.py loc 0 80 > $OUT/fuzz_loc_0_79.log 2>&1; tail -1 $OUT/fuzz_loc_0_79.log | cut -c1-300
timeout 300 python tools/gpu_fuzz_replay.py long 0 24 > $OUT/fuzz_long_0_23.log 2>&1; tail -1 $OUT/fuzz_long_0_23.log | cut -c1-300
timeout 100 python tools/gpu_fuzz_replay.py deg > $OUT/fuzz_deg.log 2>&1; tail -1 $OUT/fuzz_deg.log | cut -c1-300
