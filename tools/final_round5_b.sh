set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_final; mkdir -p $OUT
cd $R
timeout 80 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -q 2>&1 | tail -4 > $OUT/gpu_pytest_sort_voxelgrid.log
timeout 130 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
timeout 60 python tools/gpu_perf_voxelgrid.py > $OUT/vg_after.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_trace; timeout 40 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/tools/gpu_perf_voxelgrid.py ndt > $OUT/vg_under_trace.log 2> $OUT/vg_trace.err
cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $OUT/vg_kernel_stats.csv 2>/dev/null
python $R/tools/trace_summary.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $OUT/vg_kernel_trace_summary.txt 2>&1
cat $OUT/gpu_pytest_sort_voxelgrid.log; tail -c 200 $OUT/bench_full.err; cat $OUT/vg_after.log; head -c 400 $OUT/bench_full.json
