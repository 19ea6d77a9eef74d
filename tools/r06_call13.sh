set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_m; mkdir -p $OUT
cd $R
L=$R/funny_lidar_slam_amd
for lib in r05 wt0; do
  FLS_REG_LIB=$L/libfls_reg_$lib.so timeout 900 python tools/dbg_batch_stress.py 80 icp > $OUT/stress_icp_$lib.log 2>&1; echo $lib; tail -2 $OUT/stress_icp_$lib.log
done
timeout 900 python tools/dbg_batch_stress.py 80 icp > $OUT/stress_icp_cur.log 2>&1; echo cur; tail -2 $OUT/stress_icp_cur.log
FLS_REG_LIB=$L/libfls_reg_r05.so timeout 300 python tools/dbg_batch_stress.py 30 ndt > $OUT/stress_ndt_r05.log 2>&1; echo ndt r05; tail -3 $OUT/stress_ndt_r05.log
