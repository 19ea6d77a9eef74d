# round 6, GPU call 2: run-time LDS cap of the exact sort (regime by cloud size) -- correctness, the large-cloud filters again, the bench legs that regressed, scenario re-runs
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_b; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -x -q -m gpu > $OUT/pytest_sort_vg.log 2>&1; tail -3 $OUT/pytest_sort_vg.log
timeout 300 python tools/gpu_vg_large.py 6 > $OUT/vg_default.json 2> $OUT/vg_default.err
FLS_ES_LDS_BIG=4096 timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_big4096.json 2>&1
FLS_ES_LDS_SMALL=1024 timeout 200 python tools/gpu_vg_large.py 6 scan > $OUT/vg_small1024.json 2>&1
FLS_ES_DEBUG=1 timeout 200 python tools/gpu_vg_large.py 3 > $OUT/vg_default_stamps.log 2>&1
timeout 300 python tools/gpu_perf_voxelgrid.py ndt > $OUT/perf_vg_ndt.log 2>&1
timeout 300 python tools/gpu_perf_voxelgrid.py icp > $OUT/perf_vg_icp.log 2>&1
timeout 600 python tools/gpu_kd_mapping.py > $OUT/kd_mapping.json 2> $OUT/kd_mapping.err
timeout 300 python tools/gpu_fuzz_replay.py deg > $OUT/fuzz_deg.log 2>&1
timeout 600 python tools/gpu_fuzz_replay.py loc 0 40 > $OUT/fuzz_loc.log 2>&1
timeout 900 python -m pytest tests/test_gpu_batch_ranks.py -x -q -m gpu -s > $OUT/pytest_ranks.log 2>&1; tail -5 $OUT/pytest_ranks.log
tail -n 1 $OUT/fuzz_deg.log $OUT/fuzz_loc.log
cat $OUT/vg_*.json; tail -n 3 $OUT/perf_vg_*.log
