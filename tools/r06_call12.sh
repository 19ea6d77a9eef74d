set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_l; mkdir -p $OUT
cd $R
timeout 900 python tools/dbg_batch_stress.py 150 icp > $OUT/stress_icp.log 2>&1; tail -15 $OUT/stress_icp.log
FLS_DEVICE_VOXELGRID=0 timeout 900 python tools/dbg_batch_stress.py 100 icp > $OUT/stress_icp_hostfilter.log 2>&1; tail -5 $OUT/stress_icp_hostfilter.log
timeout 900 python tools/dbg_batch_stress.py 60 ndt > $OUT/stress_ndt.log 2>&1; tail -5 $OUT/stress_ndt.log
