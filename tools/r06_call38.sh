set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_au}; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q > $OUT/gpu_pytest.log 2>&1; tail -5 $OUT/gpu_pytest.log
timeout 1500 python tools/gpu_fuzz_replay.py mapping 0 404 > $OUT/fuzz_mapping_0_403.log 2>&1; tail -2 $OUT/fuzz_mapping_0_403.log | cut -c1-300
timeout 900 python tools/gpu_fuzz_replay.py loc 0 80 > $OUT/fuzz_loc_0_79.log 2>&1; tail -2 $OUT/fuzz_loc_0_79.log | cut -c1-300
timeout 900 python tools/gpu_fuzz_replay.py long 0 60 > $OUT/fuzz_long_0_59.log 2>&1; tail -2 $OUT/fuzz_long_0_59.log | cut -c1-300
timeout 600 python tools/gpu_fuzz_replay.py deg > $OUT/fuzz_deg.log 2>&1; tail -2 $OUT/fuzz_deg.log | cut -c1-300
