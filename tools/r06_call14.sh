set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_n; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -x -q -m gpu > $OUT/pytest_sort_vg.log 2>&1; tail -2 $OUT/pytest_sort_vg.log
timeout 900 python tools/dbg_batch_stress.py 150 icp > $OUT/stress_icp.log 2>&1; tail -3 $OUT/stress_icp.log
timeout 900 python tools/dbg_batch_stress.py 40 ndt > $OUT/stress_ndt.log 2>&1; tail -6 $OUT/stress_ndt.log
timeout 300 python tools/gpu_vg_large.py 8 > $OUT/vg_large.json 2>&1; tail -1 $OUT/vg_large.json
