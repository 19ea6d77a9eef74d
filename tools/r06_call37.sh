set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_at}; mkdir -p $OUT
cd $R
L=funny_lidar_slam_amd
timeout 300 python -m pytest tests/test_gpu_solver.py -x -q -s > $OUT/solver_tests.log 2>&1; tail -3 $OUT/solver_tests.log
for cid in 1 0 2 3; do timeout 600 python tools/gpu_ab_libs.py $cid $L/libfls_reg_prev.so $L/libfls_reg.so >> $OUT/ab_tail.log 2>&1; done; cat $OUT/ab_tail.log
for lib in libfls_reg_prev_timing.so libfls_reg_timing.so; do
  echo "== $lib" >> $OUT/stamps.log
  FLS_REG_LIB=$R/$L/$lib timeout 300 python tools/gpu_fanin_stamps.py "FLS_X=0" >> $OUT/stamps.log 2>&1
  FLS_REG_LIB=$R/$L/$lib timeout 300 python tools/gpu_icp_stamps.py icp >> $OUT/stamps.log 2>&1
  FLS_REG_LIB=$R/$L/$lib timeout 300 python tools/gpu_icp_stamps.py ndt >> $OUT/stamps.log 2>&1
done; cat $OUT/stamps.log
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/gpu_pytest.log 2>&1; tail -5 $OUT/gpu_pytest.log
