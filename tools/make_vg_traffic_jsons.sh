#!/bin/bash
# profiles/traffic_{es,vg}_*.json from the two `tools/prof_round5.sh <tag> vg` directories (NDT call: <dir>/vg, ICP call: <dir>/vg_icp).
# usage: bash tools/make_vg_traffic_jsons.sh gpurun_out/r05_z
set -eu
D=$1
T=$(dirname "$0")/make_kernel_traffic_json.py
python $T $D/vg es_count_scatter_kernel es_count_scatter "NDT exact sort, top levels (stop lists of a level in one launch: look-back over the tiles' published counts)"
python $T $D/vg es_level_begin es_level_begin "NDT exact sort, top levels"
python $T $D/vg es_swap_kernel es_swap "NDT exact sort, top levels"
python $T $D/vg es_task_kernel es_task_ndt "NDT source VoxelGrid: exact sort of 115,200 records (task kernel behind six pre-enqueued levels; LDS ranges of 2,048 records)"
python $T $D/vg_icp es_task_kernel es_task_icp "ICP source VoxelGrid: exact sort of 14,400 records"
python $T $D/vg vg_centroid_plan vg_centroid "NDT source VoxelGrid"
python $T $D/vg vg_heads_plan vg_heads "NDT source VoxelGrid"
python $T $D/vg vg_minmax_plan vg_minmax "NDT source VoxelGrid"
