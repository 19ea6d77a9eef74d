# round 6, GPU call 3: large-cloud filter after (grid, hand-over threshold, minmax blocks, wave-summed long leaves): correctness, A/B, the bench legs
set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_c; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -x -q -m gpu -s > $OUT/pytest_sort_vg.log 2>&1; tail -3 $OUT/pytest_sort_vg.log
timeout 300 python tools/gpu_vg_large.py 6 > $OUT/vg_default.json 2> $OUT/vg_default.err
FLS_ES_HANDOVER=131072 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_handover131072.json 2>&1
FLS_ES_HANDOVER=65536 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_handover65536.json 2>&1
FLS_ES_HANDOVER=16384 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_handover16384.json 2>&1
FLS_ES_LDS_BIG=4096 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_big4096.json 2>&1
FLS_ES_LDS_BIG=2048 timeout 200 python tools/gpu_vg_large.py 6 icp,loam_planar,loam_corner > $OUT/vg_big2048.json 2>&1
FLS_ES_DEBUG=1 timeout 200 python tools/gpu_vg_large.py 3 > $OUT/vg_default_stamps.log 2>&1
FLS_ES_DEBUG=1 FLS_ES_LDS_BIG=2048 timeout 200 python tools/gpu_vg_large.py 3 loam_planar > $OUT/vg_big2048_stamps.log 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_vgl; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/p_vgl -- python $R/tools/gpu_vg_large.py 6 loam_planar > $OUT/vg_trace.log 2>&1
cp $(find /tmp/p_vgl -name "*kernel_stats.csv" | head -1) $OUT/vg_loam_planar_kernel_stats.csv 2>/dev/null
cd $R
timeout 600 python tools/gpu_kd_mapping.py > $OUT/kd_mapping.json 2> $OUT/kd_mapping.err
timeout 900 python -m pytest tests/test_gpu_fuzz_replay.py tests/test_gpu_mapping_replay.py -x -q -m gpu > $OUT/pytest_replays.log 2>&1; tail -3 $OUT/pytest_replays.log
cat $OUT/vg_*.json
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_c/kd_mapping.json"))
for k in ("icp_optimized","loam_full"):
    for s,v in d[k].items():
        if isinstance(v,dict): print(k,s,"kf_update_only %.3f ms" % v["ms_keyframe_update_only"])
PY
