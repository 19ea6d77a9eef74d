#!/bin/bash
# End-of-round evidence of round 5 in ONE gpurun call: the full GPU test suite, the full bench line, kernel trace + PMC passes of the headline and of
# the ICP / NDT calls from host buffers, the VoxelGrid table.  Fills gpurun_out/r05_z/ (what gets copied to profiles/r05_z_*).
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_z; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests -m gpu -q 2>&1 | tail -12 > $OUT/gpu_pytest.log
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
timeout 200 python tools/gpu_perf_voxelgrid.py > $OUT/vg_after.log 2>&1
for k in ndt icp; do FLS_ES_DEBUG=1 timeout 100 python tools/gpu_perf_voxelgrid.py $k 2>&1 | tail -12 > $OUT/${k}_exact_sort_stamps.log; done
timeout 500 bash tools/prof_round5.sh r05_z headline > /dev/null 2>&1
timeout 400 bash tools/prof_round5.sh r05_z vg > /dev/null 2>&1
cat $OUT/gpu_pytest.log | tail -3; tail -c 300 $OUT/bench_full.err; cat $OUT/vg_after.log
