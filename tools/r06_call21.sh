set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_u; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_fuzz_replay.py -x -q -m gpu -k "tie" -s > $OUT/pytest_tie.log 2>&1; tail -8 $OUT/pytest_tie.log
timeout 600 python tools/gpu_fuzz_replay.py mapping 318 3 > $OUT/fuzz_tie_mapping.log 2>&1; tail -4 $OUT/fuzz_tie_mapping.log
timeout 600 python tools/gpu_fuzz_replay.py loc 57 3 > $OUT/fuzz_tie_loc.log 2>&1; tail -4 $OUT/fuzz_tie_loc.log
