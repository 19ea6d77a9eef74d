#!/bin/bash
# Profiling pass of round 6 (one gpurun call): tools/prof_round5.sh's three workloads (headline / kinds / vg: kernel trace + stats + four PMC passes each, every
# --pmc group in a run of its own) plus the map-side filter of the LoamFull planar keyframe deque (1.55 M points) -> gpurun_out/<tag>/{headline,kinds,vg,vg_icp,vg_large}/
set -u
TAG=${1:-r06_z}
R=${GRAFT_REPO_ROOT:-$PWD}
bash $R/tools/prof_round5.sh $TAG all > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
OUT=$R/gpurun_out/$TAG/vg_large; mkdir -p $OUT
rm -rf /tmp/p_trace; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_trace -- python $R/tools/gpu_vg_large.py 8 loam_planar > $OUT/under_trace.log 2> $OUT/trace.err
cp $(find /tmp/p_trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
cp $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) $OUT/kernel_trace.csv 2>/dev/null
python $R/tools/trace_summary.py $OUT/kernel_trace.csv > $OUT/kernel_trace_summary.txt 2>&1
i=0
for PMC in "FETCH_SIZE" "WRITE_SIZE" "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_BUSY_CYCLES SQ_WAVES"; do
  i=$((i+1)); rm -rf /tmp/p_pmc
  timeout 300 rocprofv3 --kernel-trace --pmc $PMC --output-format csv -d /tmp/p_pmc -- python $R/tools/gpu_vg_large.py 8 loam_planar > /dev/null 2> $OUT/pmc$i.err
  cp $(find /tmp/p_pmc -name "*counter_collection.csv" | head -1) $OUT/pmc${i}_counter_collection.csv 2>/dev/null
  echo "== --pmc $PMC" >> $OUT/pmc_summary.txt
  python $R/tools/pmc_summary.py $OUT/pmc${i}_counter_collection.csv >> $OUT/pmc_summary.txt 2>&1
done
# traffic records (written into profiles/ of the box's copy, then brought home under gpurun_out/<tag>/traffic/)
D=$R/gpurun_out/$TAG; T=$R/tools/make_kernel_traffic_json.py
python $T $D/headline ivox_knn_kernel ivox_knn "bench.py --steps 20 --warmup 5 (BASELINE configs[1]: 115,200-pt scan, 1e6-pt iVox map)" > $D/traffic_jsons.log 2>&1
python $T $D/kinds p2plane_fit_solve_kernel p2plane_fit_solve "tools/kinds_trace.py: ten resident Matches of every kind" >> $D/traffic_jsons.log 2>&1
python $T $D/kinds ndt_lanes_kernel ndt_lanes "tools/kinds_trace.py" >> $D/traffic_jsons.log 2>&1
python $T $D/kinds icp_knn_fit_kernel icp_knn_fit "tools/kinds_trace.py" >> $D/traffic_jsons.log 2>&1
python $T $D/kinds grid_knn_dual_kernel grid_knn_dual "tools/kinds_trace.py" >> $D/traffic_jsons.log 2>&1
python $T $D/kinds feature_fit_dual_kernel feature_fit_dual "tools/kinds_trace.py" >> $D/traffic_jsons.log 2>&1
bash $R/tools/make_vg_traffic_jsons.sh $D >> $D/traffic_jsons.log 2>&1
python $T $D/vg_large es_task_kernel es_task_planar_deque "map-side VoxelGrid of the LoamFull planar keyframe deque: exact sort of 1,555,200 records (task kernel behind 14 pre-enqueued levels; ticket queue, LDS ranges of 4,096 records)" >> $D/traffic_jsons.log 2>&1
python $T $D/vg_large es_count_scatter_kernel es_count_scatter_planar_deque "the same call: stop lists of a level" >> $D/traffic_jsons.log 2>&1
python $T $D/vg_large vg_centroid_plan vg_centroid_planar_deque "the same call: leaf sums (runs of 128+ points by a wave)" >> $D/traffic_jsons.log 2>&1
mkdir -p $D/traffic; cp $R/profiles/traffic_*.json $D/traffic/
rm -f $R/gpurun_out/$TAG/*/kernel_trace.csv   # (tens of MB each; the summaries and the counter files are what is kept)
ls $R/gpurun_out/$TAG/*
