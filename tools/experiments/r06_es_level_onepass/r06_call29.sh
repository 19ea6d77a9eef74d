set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_ac}; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_exact_sort.py -q -x 2>&1 | tail -15 > $OUT/pytest_sort.log; tail -5 $OUT/pytest_sort.log
timeout 900 python -m pytest tests/test_gpu_voxelgrid.py -q -x 2>&1 | tail -8 > $OUT/pytest_vg.log; tail -3 $OUT/pytest_vg.log
for op in 0 1; do
  for k in ndt icp; do FLS_ES_ONEPASS=$op FLS_DEVICE_VOXELGRID=1 timeout 200 python tools/gpu_perf_voxelgrid.py $k 2>&1 | tail -1 | sed "s/^/onepass=$op /" >> $OUT/vg_call.log; done
  FLS_ES_ONEPASS=$op timeout 200 python tools/gpu_vg_large.py 8 2>&1 | tail -1 | sed "s/^/onepass=$op /" >> $OUT/vg_large.log
done
FLS_ES_ITEMS=8 FLS_DEVICE_VOXELGRID=1 timeout 200 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1 | sed "s/^/items=8 /" >> $OUT/vg_call.log
FLS_DEVICE_VOXELGRID=2 timeout 200 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1 | sed "s/^/index-order /" >> $OUT/vg_call.log
cat $OUT/vg_call.log $OUT/vg_large.log
cd /tmp && export TMPDIR=/tmp
for w in loam_planar scan; do
rm -rf /tmp/p_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/tools/gpu_vg_large.py 8 $w > $OUT/under_trace_$w.log 2> $OUT/trace_$w.err
python $R/tools/trace_summary.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_summary_$w.txt 2>&1
cat $OUT/kernel_trace_summary_$w.txt | head -4
done
