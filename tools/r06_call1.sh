# round 6, GPU call 1: (a) bisect of the large-cloud exact-VoxelGrid regression, (b) the HIP path through the pinned random scenarios (next #3)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_a; mkdir -p $OUT
cd $R
L=$R/funny_lidar_slam_amd
timeout 300 python tools/gpu_vg_large.py 6 > $OUT/vg_default.json 2> $OUT/vg_default.err
FLS_REG_LIB=$L/libfls_reg_lds4096.so timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_lds4096.json 2>&1
FLS_REG_LIB=$L/libfls_reg_lds8192.so timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_lds8192.json 2>&1
FLS_ES_LOOKBACK=0 timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_lookback0.json 2>&1
FLS_ES_LOOKBACK=0 FLS_REG_LIB=$L/libfls_reg_lds8192.so timeout 200 python tools/gpu_vg_large.py 6 > $OUT/vg_lds8192_lookback0.json 2>&1
FLS_ES_DEBUG=1 timeout 200 python tools/gpu_vg_large.py 3 loam_planar,icp > $OUT/vg_default_stamps.log 2>&1
FLS_ES_DEBUG=1 FLS_REG_LIB=$L/libfls_reg_lds8192.so timeout 200 python tools/gpu_vg_large.py 3 loam_planar,icp > $OUT/vg_lds8192_stamps.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_vgl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_vgl -- python $R/tools/gpu_vg_large.py 6 loam_planar > $OUT/vg_trace.log 2>&1
cp $(find /tmp/p_vgl -name "*kernel_stats.csv" | head -1) $OUT/vg_loam_planar_kernel_stats.csv 2>/dev/null
rm -rf /tmp/p_vgl; FLS_REG_LIB=$L/libfls_reg_lds8192.so timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_vgl -- python $R/tools/gpu_vg_large.py 6 loam_planar > $OUT/vg_trace_lds8192.log 2>&1
cp $(find /tmp/p_vgl -name "*kernel_stats.csv" | head -1) $OUT/vg_loam_planar_lds8192_kernel_stats.csv 2>/dev/null
cd $R
timeout 300 python tools/gpu_fuzz_replay.py deg > $OUT/fuzz_deg.log 2>&1
timeout 600 python tools/gpu_fuzz_replay.py loc 0 40 > $OUT/fuzz_loc.log 2>&1
timeout 1500 python tools/gpu_fuzz_replay.py mapping 4 200 > $OUT/fuzz_mapping.log 2>&1
tail -2 $OUT/fuzz_deg.log $OUT/fuzz_loc.log $OUT/fuzz_mapping.log
cat $OUT/vg_*.json
