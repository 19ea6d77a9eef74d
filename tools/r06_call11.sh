set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_k; mkdir -p $OUT
cd $R
timeout 600 python tools/dbg_determinism.py vg 400 > $OUT/det_vg.log 2>&1; cat $OUT/det_vg.log
timeout 600 python tools/dbg_determinism.py icp 200 > $OUT/det_icp.log 2>&1; cat $OUT/det_icp.log
FLS_DEVICE_VOXELGRID=0 timeout 600 python tools/dbg_determinism.py icp 100 > $OUT/det_icp_hostfilter.log 2>&1; cat $OUT/det_icp_hostfilter.log
