set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_aa; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_exact_sort.py -q -x 2>&1 | tail -15 > $OUT/pytest_sort.log; tail -5 $OUT/pytest_sort.log
timeout 900 python -m pytest tests/test_gpu_voxelgrid.py -q -x 2>&1 | tail -8 > $OUT/pytest_vg.log; tail -3 $OUT/pytest_vg.log
for op in 0 1; do
  for k in ndt icp; do FLS_ES_ONEPASS=$op FLS_DEVICE_VOXELGRID=1 timeout 200 python tools/gpu_perf_voxelgrid.py $k 2>&1 | tail -1 | sed "s/^/onepass=$op /" >> $OUT/vg_call.log; done
  FLS_ES_ONEPASS=$op timeout 200 python tools/gpu_vg_large.py 8 2>&1 | tail -1 | sed "s/^/onepass=$op /" >> $OUT/vg_large.log
done
for it in 8; do
  FLS_ES_ITEMS=$it FLS_DEVICE_VOXELGRID=1 timeout 200 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1 | sed "s/^/items=$it /" >> $OUT/vg_call.log
done
FLS_DEVICE_VOXELGRID=2 timeout 200 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1 | sed "s/^/index-order /" >> $OUT/vg_call.log
cat $OUT/vg_call.log $OUT/vg_large.log
