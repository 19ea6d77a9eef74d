set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_f; mkdir -p $OUT
cd $R
L=$R/funny_lidar_slam_amd
timeout 600 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -x -q -m gpu > $OUT/pytest_sort_vg.log 2>&1; tail -2 $OUT/pytest_sort_vg.log
timeout 600 python tools/es_fuzz.py > $OUT/es_fuzz.log 2>&1; tail -2 $OUT/es_fuzz.log
for g in 256 128; do for big in 8192 2048; do
  FLS_ES_GRID=$g FLS_ES_LDS_BIG=$big timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_wt1_g${g}_b${big}.json 2>&1
  FLS_REG_LIB=$L/libfls_reg_wt0.so FLS_ES_GRID=$g FLS_ES_LDS_BIG=$big timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_wt0_g${g}_b${big}.json 2>&1
done; done
FLS_ES_GRID=32 timeout 200 python tools/gpu_vg_large.py 6 scan > $OUT/vg_wt1_g32_scan.json 2>&1
FLS_ES_DEBUG=1 timeout 200 python tools/gpu_vg_large.py 2 loam_planar,scan > $OUT/vg_stamps.log 2>&1
for i in 1 2; do
timeout 300 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1 >> $OUT/perf_vg_wt1.log
FLS_REG_LIB=$L/libfls_reg_wt0.so timeout 300 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1 >> $OUT/perf_vg_wt0.log
timeout 300 python tools/gpu_perf_voxelgrid.py icp 2>&1 | tail -1 >> $OUT/perf_vg_wt1.log
FLS_REG_LIB=$L/libfls_reg_wt0.so timeout 300 python tools/gpu_perf_voxelgrid.py icp 2>&1 | tail -1 >> $OUT/perf_vg_wt0.log
done
for f in $OUT/vg_wt*.json; do echo $f; tail -1 $f; done; cat $OUT/perf_vg_wt1.log $OUT/perf_vg_wt0.log; grep "fls exact sort" $OUT/vg_stamps.log | grep -v "global partition " | tail -22
