set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_s; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_fuzz_replay.py -x -q -m gpu -k "tie" > $OUT/pytest_tie.log 2>&1; tail -15 $OUT/pytest_tie.log
timeout 600 python tools/gpu_fuzz_replay.py mapping 317 4 > $OUT/fuzz_tie_mapping.log 2>&1; tail -6 $OUT/fuzz_tie_mapping.log
timeout 600 python tools/gpu_fuzz_replay.py loc 57 3 > $OUT/fuzz_tie_loc.log 2>&1; tail -5 $OUT/fuzz_tie_loc.log
timeout 600 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -x -q -m gpu > $OUT/pytest_sort_vg.log 2>&1; tail -2 $OUT/pytest_sort_vg.log
timeout 300 python tools/gpu_vg_large.py 8 > $OUT/vg_large.json 2>&1; tail -1 $OUT/vg_large.json
