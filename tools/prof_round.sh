#!/bin/bash
# Profiling pass of one round (run on the GPU box through gpurun): kernel trace + stats of the default-ish bench command, PMC passes in
# their OWN runs (no trace domains besides --kernel-trace), kernel traces of every kind / the feature front-end / mapping mode.
# usage: bash tools/prof_round.sh r02a        -> gpurun_out/<tag>/...
set -u
TAG=${1:-r02}
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
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES" "GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rm -rf /tmp/p_pmc$i
  rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/p_pmc$i -- $BENCH > /dev/null 2> $OUT/pmc$i.log
  f=$(find /tmp/p_pmc$i -name "*counter_collection.csv" | head -1)
  echo "== --pmc $PMC" >> $OUT/pmc_summary.txt
  python $R/tools/pmc_summary.py $f >> $OUT/pmc_summary.txt 2>&1
done
rm -rf /tmp/p_kinds; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_kinds -- python $R/tools/kinds_trace.py > $OUT/kinds.log 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_kinds -name "*kernel_trace.csv" | head -1) 2.0 > $OUT/all_kinds_kernel_trace_summary.txt 2>&1
rm -rf /tmp/p_feat; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_feat -- python $R/tools/gpu_features.py > $OUT/features_timing.txt 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_feat -name "*kernel_trace.csv" | head -1) 2.0 > $OUT/features_kernel_trace_summary.txt 2>&1
rm -rf /tmp/p_upd; rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_upd -- python $R/tools/gpu_perf_update.py > $OUT/mapping_mode.log 2>&1
python $R/tools/trace_summary.py $(find /tmp/p_upd -name "*kernel_trace.csv" | head -1) 1.0 > $OUT/mapping_mode_kernel_trace_summary.txt 2>&1
python $R/tools/gpu_pcie_inclusive.py > $OUT/pcie_inclusive.txt 2>&1
ls -la $OUT
