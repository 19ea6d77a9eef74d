set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_d; mkdir -p $OUT
cd $R
L=funny_lidar_slam_amd
FLS_ES_DEBUG=1 timeout 200 python tools/gpu_vg_large.py 2 loam_planar > $OUT/vg_planar_stamps.log 2>&1
FLS_ES_DEBUG=1 FLS_ES_HANDOVER=131072 timeout 200 python tools/gpu_vg_large.py 2 loam_planar > $OUT/vg_planar_h131072_stamps.log 2>&1
FLS_ES_DEBUG=1 FLS_ES_LDS_BIG=2048 timeout 200 python tools/gpu_vg_large.py 2 loam_planar > $OUT/vg_planar_big2048_stamps.log 2>&1
timeout 600 python tools/gpu_ab_libs.py 0 $L/libfls_reg_base.so $L/libfls_reg.so > $OUT/ab_icp.log 2>&1
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "icp or config1" > $OUT/pytest_icp.log 2>&1; tail -3 $OUT/pytest_icp.log
cat $OUT/ab_icp.log; grep "fls exact sort" $OUT/vg_planar_stamps.log | grep -v "global partition " | tail -12
