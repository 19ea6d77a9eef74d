set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_h; mkdir -p $OUT
cd $R
for h in 32768 65536 131072; do for big in 2048 4096; do
  FLS_ES_HANDOVER=$h FLS_ES_LDS_BIG=$big timeout 200 python tools/gpu_vg_large.py 8 icp,loam_planar,loam_corner > $OUT/vg_h${h}_b${big}.json 2>&1
done; done
for b in 16384 32768 65536 131072; do
  FLS_ES_BIG=$b timeout 200 python tools/gpu_vg_large.py 8 scan > $OUT/vg_scan_big${b}.json 2>&1
done
FLS_ES_BIG=0 timeout 200 python tools/gpu_vg_large.py 8 scan > $OUT/vg_scan_big0.json 2>&1
