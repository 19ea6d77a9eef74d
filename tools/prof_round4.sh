#!/bin/bash
# Profiling pass of round 4 (run on the GPU box through gpurun): the full bench line, a kernel trace + stats of the headline command, four
# PMC passes in their OWN runs (no trace domain besides --kernel-trace), the production call launch by launch, the VoxelGrid legs.
# usage: bash tools/prof_round4.sh r04x   -> gpurun_out/<tag>/...
set -u
TAG=${1:-r04}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 python $R/bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-extras"
rm -rf /tmp/p_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.log
cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python $R/tools/trace_summary.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_summary.txt 2>&1
python $R/tools/trace_timeline.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_timeline.txt 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/p_pmc$i
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/p_pmc$i -- $BENCH > /dev/null 2> $OUT/pmc$i.log
  f=$(find /tmp/p_pmc$i -name "*counter_collection.csv" | head -1)
  cp $f $OUT/pmc${i}_counter_collection.csv 2>/dev/null
  echo "== --pmc $PMC" >> $OUT/pmc_summary.txt
  python $R/tools/pmc_summary.py $f >> $OUT/pmc_summary.txt 2>&1
done
timeout 200 bash $R/tools/prof_prod_call.sh ${TAG}_prod > /dev/null 2>&1
timeout 200 python $R/tools/gpu_perf_voxelgrid.py > $OUT/voxelgrid_host_vs_device.txt 2>&1
ls -la $OUT
