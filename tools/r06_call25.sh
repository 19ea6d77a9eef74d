set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_w; mkdir -p $OUT
cd $R
L=funny_lidar_slam_amd
timeout 600 python tools/gpu_ab_libs.py 0 $L/libfls_reg_base.so $L/libfls_reg.so > $OUT/ab_icp.log 2>&1; cat $OUT/ab_icp.log
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "icp or config1 or batch" > $OUT/pytest_icp.log 2>&1; tail -3 $OUT/pytest_icp.log
timeout 600 python tools/gpu_fuzz_replay.py loc 0 8 > $OUT/fuzz_loc.log 2>&1; tail -1 $OUT/fuzz_loc.log
timeout 600 python tools/gpu_fuzz_replay.py mapping 0 8 > $OUT/fuzz_map.log 2>&1; tail -1 $OUT/fuzz_map.log
