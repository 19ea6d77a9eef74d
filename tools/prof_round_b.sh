#!/bin/bash
# Second profiling pass of a round: kernel traces of what was added after tools/prof_round.sh ran (device VoxelGrid source filter,
# NDT in-place image edits).  usage: bash tools/prof_round_b.sh r02b   -> gpurun_out/<tag>/...
set -u
TAG=${1:-r02b}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_vg; FLS_DEVICE_VOXELGRID=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_vg -- python $R/tools/gpu_perf_voxelgrid.py ndt > $OUT/voxelgrid_ndt.log 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_vg -name "*kernel_trace.csv" | head -1) 0.0 > $OUT/voxelgrid_kernel_trace_summary.txt 2>&1
python $R/tools/gpu_perf_voxelgrid.py > $OUT/voxelgrid_host_vs_device.txt 2>&1
rm -rf /tmp/p_ndt; FLS_HOST_TIMING=1 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_ndt -- python $R/tools/gpu_perf_ndt_update.py > $OUT/ndt_mapping_mode.log 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_ndt -name "*kernel_trace.csv" | head -1) 0.0 > $OUT/ndt_mapping_mode_kernel_trace_summary.txt 2>&1
FLS_DEVICE_VOXELGRID=1 FLS_HOST_TIMING=1 python $R/tools/gpu_perf_ndt_update.py > $OUT/ndt_mapping_mode_device_filter.log 2>&1
ls -la $OUT
