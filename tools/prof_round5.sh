#!/bin/bash
# Profiling pass of round 5 (run on the GPU box through gpurun).  usage: bash tools/prof_round5.sh <tag> [headline|kinds|vg|all]
#   headline  kernel trace + stats + four PMC passes (own runs, --kernel-trace only) of the bench command  -> gpurun_out/<tag>/headline/
#   kinds     the same for tools/kinds_trace.py (ten resident Matches of every kind)                       -> .../kinds/
#   vg        the same for fls_match from host buffers of NDT (source VoxelGrid + exact sort inside)       -> .../vg/
# Each directory holds kernel_trace.csv, kernel_stats.csv, kernel_trace_summary.txt and pmc<i>_counter_collection.csv: what
# tools/make_kernel_traffic_json.py turns into profiles/traffic_<kernel>.json.
set -u
TAG=${1:-r05}; WHAT=${2:-all}
R=${GRAFT_REPO_ROOT:-$PWD}
cd /tmp && export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1
pass() {  # <subdir> <command...>
  local sub=$1; shift
  local OUT=$R/gpurun_out/$TAG/$sub; mkdir -p $OUT
  rm -rf /tmp/p_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- "$@" > $OUT/under_trace.log 2> $OUT/trace.err
  cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
  cp $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv 2>/dev/null
  python $R/tools/trace_summary.py $OUT/kernel_trace.csv > $OUT/kernel_trace_summary.txt 2>&1
  local i=0
  for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
    i=$((i+1)); rm -rf /tmp/p_pmc
    timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/p_pmc -- "$@" > /dev/null 2> $OUT/pmc$i.err
    cp $(find /tmp/p_pmc -name "*counter_collection.csv" | head -1) $OUT/pmc${i}_counter_collection.csv 2>/dev/null
    echo "== --pmc $PMC" >> $OUT/pmc_summary.txt
    python $R/tools/pmc_summary.py $OUT/pmc${i}_counter_collection.csv >> $OUT/pmc_summary.txt 2>&1
  done
}
case $WHAT in headline|all) pass headline python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-extras;; esac
case $WHAT in kinds|all) pass kinds python $R/tools/kinds_trace.py;; esac
case $WHAT in vg|all) pass vg python $R/tools/gpu_perf_voxelgrid.py ndt; pass vg_icp python $R/tools/gpu_perf_voxelgrid.py icp;; esac
ls $R/gpurun_out/$TAG
