#!/bin/bash
# Profiling pass of round 3 (run on the GPU box through gpurun): kernel trace + stats of the headline bench command, the five PMC
# passes in their OWN runs (no trace domain besides --kernel-trace), kernel traces of every kind, the kd-tree kinds' mapping mode and
# the loop-closure matcher.  usage: bash tools/prof_round3.sh r03x   -> gpurun_out/<tag>/...
set -u
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
BENCH="python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-batch --no-extras"
$BENCH > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rm -rf /tmp/p_trace; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- $BENCH > $OUT/bench_under_trace.json 2> $OUT/trace.log
cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
python $R/tools/trace_summary.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_summary.txt 2>&1
python $R/tools/trace_timeline.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) > $OUT/kernel_trace_timeline.txt 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1))
  rm -rf /tmp/p_pmc$i
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/p_pmc$i -- $BENCH > /dev/null 2> $OUT/pmc$i.log
  f=$(find /tmp/p_pmc$i -name "*counter_collection.csv" | head -1)
  cp $f $OUT/pmc${i}_counter_collection.csv 2>/dev/null
  echo "== --pmc $PMC" >> $OUT/pmc_summary.txt
  python $R/tools/pmc_summary.py $f >> $OUT/pmc_summary.txt 2>&1
done
rm -rf /tmp/p_kinds; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kinds -- python $R/tools/kinds_trace.py > $OUT/kinds.log 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_kinds -name "*kernel_trace.csv" | head -1) 2.0 > $OUT/all_kinds_kernel_trace_summary.txt 2>&1
rm -rf /tmp/p_kd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kd -- python $R/tools/gpu_kd_mapping.py > $OUT/kd_mapping_mode.json 2> $OUT/kd_mapping.err
python $R/tools/trace_summary.py $(find /tmp/p_kd -name "*kernel_trace.csv" | head -1) 0.0 > $OUT/kd_mapping_kernel_trace_summary.txt 2>&1
rm -rf /tmp/p_loop; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_loop -- python $R/tools/gpu_loop.py > $OUT/loop_closure.log 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_loop -name "*kernel_trace.csv" | head -1) 0.0 > $OUT/loop_closure_kernel_trace_summary.txt 2>&1
rm -rf /tmp/p_upd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_upd -- python $R/tools/gpu_perf_update.py > $OUT/mapping_mode.log 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_upd -name "*kernel_trace.csv" | head -1) 1.0 > $OUT/mapping_mode_kernel_trace_summary.txt 2>&1
ls -la $OUT
