# usage: bash tools/prof_vg_call.sh <tag> ["ENV=1 ENV2=0"]  -- fls_match from host buffers for ICP / NDT (the source VoxelGrid is inside the call):
# kernel + copy trace read launch by launch, per-kernel stats, exact-sort stage stamps, for FLS_DEVICE_VOXELGRID = 1 (std::sort order) and 2 (index order)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-vg}; shift || true
for kv in ${1:-}; do export $kv; done
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python -c "import torch" > /dev/null 2>&1   # page the image in before anything is timed
for kind in ndt icp; do
  for mode in 1 2; do
    FLS_DEVICE_VOXELGRID=$mode timeout 300 python $R/tools/gpu_perf_voxelgrid.py $kind 2>&1 | tail -1 >> $OUT/untraced.log
    rm -rf /tmp/p_vg
    FLS_DEVICE_VOXELGRID=$mode timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/p_vg -- python $R/tools/gpu_perf_voxelgrid.py $kind > $OUT/${kind}_m${mode}_under_trace.log 2>&1
    K=$(find /tmp/p_vg -name "*kernel_trace.csv" | head -1); M=$(find /tmp/p_vg -name "*memory_copy_trace.csv" | head -1); S=$(find /tmp/p_vg -name "*kernel_stats.csv" | head -1)
    python $R/tools/trace_sequence.py $K $M --skip 0.7 --n 90 > $OUT/${kind}_m${mode}_sequence.txt 2>&1
    cp $S $OUT/${kind}_m${mode}_kernel_stats.csv 2>/dev/null
  done
  FLS_ES_DEBUG=1 FLS_DEVICE_VOXELGRID=1 timeout 300 python $R/tools/gpu_perf_voxelgrid.py $kind 2>&1 | tail -4 > $OUT/${kind}_es_stamps.log
done
cat $OUT/untraced.log
