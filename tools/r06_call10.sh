set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_j; mkdir -p $OUT
cd $R
L=$R/funny_lidar_slam_amd
timeout 300 python tools/dbg_batch_icp.py > $OUT/dbg_default.log 2>&1
FLS_DEVICE_VOXELGRID=0 timeout 300 python tools/dbg_batch_icp.py > $OUT/dbg_hostfilter.log 2>&1
FLS_DEVICE_VOXELGRID=2 timeout 300 python tools/dbg_batch_icp.py > $OUT/dbg_indexorder.log 2>&1
for f in default hostfilter indexorder; do echo == $f; cat $OUT/dbg_$f.log | tail -12; done
for lib in carry0 carry16 carry32; do
 FLS_REG_LIB=$L/libfls_reg_$lib.so timeout 200 python tools/gpu_vg_large.py 8 > $OUT/vg_$lib.json 2>&1; tail -1 $OUT/vg_$lib.json
 for i in 1 2; do FLS_REG_LIB=$L/libfls_reg_$lib.so timeout 300 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1; FLS_REG_LIB=$L/libfls_reg_$lib.so timeout 300 python tools/gpu_perf_voxelgrid.py icp 2>&1 | tail -1; done
done
