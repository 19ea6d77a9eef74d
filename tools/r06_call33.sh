set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_ah}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -q -x 2>&1 | tail -15 > $OUT/pytest_sort_vg.log; tail -4 $OUT/pytest_sort_vg.log | cut -c1-300
