set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_e; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -x -q -m gpu > $OUT/pytest_sort_vg.log 2>&1; tail -2 $OUT/pytest_sort_vg.log
timeout 300 python tools/gpu_vg_large.py 6 > $OUT/vg_default.json 2> $OUT/vg_default.err
FLS_ES_GRID=128 timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_grid128.json 2>&1
FLS_ES_GRID=64 timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_grid64.json 2>&1
FLS_ES_HANDOVER=131072 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_h131072.json 2>&1
FLS_ES_LDS_BIG=2048 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_big2048.json 2>&1
FLS_ES_LDS_BIG=4096 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_big4096.json 2>&1
FLS_ES_DEBUG=1 timeout 200 python tools/gpu_vg_large.py 2 loam_planar,scan > $OUT/vg_stamps.log 2>&1
timeout 300 python tools/gpu_perf_voxelgrid.py ndt > $OUT/perf_vg_ndt.log 2>&1
timeout 300 python tools/gpu_perf_voxelgrid.py icp > $OUT/perf_vg_icp.log 2>&1
cat $OUT/vg_*.json; tail -n 1 $OUT/perf_vg_*.log; grep "fls exact sort" $OUT/vg_stamps.log | grep -v "global partition " | tail -24
