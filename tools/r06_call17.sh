set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_q; mkdir -p $OUT
cd $R
timeout 1500 python tools/gpu_fuzz_replay.py long 0 60 > $OUT/fuzz_long.log 2>&1; tail -1 $OUT/fuzz_long.log
timeout 1500 python tools/gpu_fuzz_replay.py mapping 204 200 > $OUT/fuzz_mapping_b.log 2>&1; tail -1 $OUT/fuzz_mapping_b.log
timeout 600 python tools/gpu_fuzz_replay.py loc 40 40 > $OUT/fuzz_loc_b.log 2>&1; tail -1 $OUT/fuzz_loc_b.log
timeout 300 python tools/gpu_vg_large.py 8 > $OUT/vg_large.json 2>&1; tail -1 $OUT/vg_large.json
timeout 900 python tools/dbg_batch_stress.py 60 icp > $OUT/stress_icp.log 2>&1; tail -1 $OUT/stress_icp.log
timeout 900 python tools/dbg_batch_stress.py 30 ndt > $OUT/stress_ndt.log 2>&1; tail -1 $OUT/stress_ndt.log
