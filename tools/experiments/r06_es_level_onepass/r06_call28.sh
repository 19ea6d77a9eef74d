set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_ab; mkdir -p $OUT
cd $R
for lib in "" lvl1 lvl2; do
  L=""; [ -n "$lib" ] && L=$R/funny_lidar_slam_amd/libfls_reg_$lib.so
  FLS_REG_LIB=$L FLS_DEVICE_VOXELGRID=1 timeout 200 python tools/gpu_perf_voxelgrid.py ndt 2>&1 | tail -1 | sed "s/^/lib=$lib /" >> $OUT/vg_call.log
  FLS_REG_LIB=$L timeout 200 python tools/gpu_vg_large.py 8 2>&1 | tail -1 | sed "s/^/lib=$lib /" >> $OUT/vg_large.log
done
cat $OUT/vg_call.log $OUT/vg_large.log
cd /tmp && export TMPDIR=/tmp
for w in loam_planar scan; do
rm -rf /tmp/p_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/tools/gpu_vg_large.py 8 $w > $OUT/under_trace_$w.log 2> $OUT/trace_$w.err
python $R/tools/trace_summary.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_summary_$w.txt 2>&1
cat $OUT/kernel_trace_summary_$w.txt | head -12
done
