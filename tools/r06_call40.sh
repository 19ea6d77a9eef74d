set -u
R=${GRAFT_REPO_ROOT:-$PWD}; TAG=${1:-r06_aw}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
L=funny_lidar_slam_amd
for cid in 0 1 2 3; do timeout 600 python tools/gpu_ab_libs.py $cid $L/libfls_reg_head.so $L/libfls_reg.so >> $OUT/ab_reduce.log 2>&1; done; cat $OUT/ab_reduce.log
FLS_REG_LIB=$R/$L/libfls_reg_timing.so timeout 300 python tools/gpu_icp_stamps.py icp > $OUT/stamps.log 2>&1
FLS_REG_LIB=$R/$L/libfls_reg_timing.so timeout 300 python tools/gpu_fanin_stamps.py "FLS_X=0" >> $OUT/stamps.log 2>&1
cat $OUT/stamps.log
bash tools/final_round6.sh $TAG
bash tools/prof_round6.sh $TAG > $OUT/prof.log 2>&1; tail -3 $OUT/prof.log
