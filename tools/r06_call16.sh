set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/r06_p; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_exact_sort.py tests/test_gpu_voxelgrid.py -x -q -m gpu > $OUT/pytest_sort_vg.log 2>&1; tail -2 $OUT/pytest_sort_vg.log
timeout 300 python tools/gpu_vg_large.py 8 > $OUT/vg_fused_all.json 2>&1; tail -1 $OUT/vg_fused_all.json
FLS_VG_FUSED_MAX=131072 timeout 300 python tools/gpu_vg_large.py 8 > $OUT/vg_fused_small_only.json 2>&1; tail -1 $OUT/vg_fused_small_only.json
FLS_ES_TOP_EXTRA=-3 timeout 300 python tools/gpu_vg_large.py 8 icp,loam_planar,loam_corner > $OUT/vg_fused_all_top_m3.json 2>&1; tail -1 $OUT/vg_fused_all_top_m3.json
FLS_ES_DEBUG=1 timeout 300 python tools/gpu_vg_large.py 2 loam_planar > $OUT/vg_stamps.log 2>&1; grep "fls exact sort" $OUT/vg_stamps.log | grep -v "global partition " | head -6
timeout 600 python tools/gpu_kd_mapping.py > $OUT/kd_mapping.json 2> $OUT/kd_mapping.err
python - <<'PY'
import json
d=json.load(open("gpurun_out/r06_p/kd_mapping.json"))
for k in ("icp_optimized","loam_full"):
    for s,v in d[k].items():
        if isinstance(v,dict): print(k,s,"kf_update_only %.3f ms" % v["ms_keyframe_update_only"])
PY
timeout 900 python -m pytest tests/test_gpu_mapping_replay.py tests/test_gpu_fuzz_replay.py -x -q -m gpu > $OUT/pytest_replays.log 2>&1; tail -2 $OUT/pytest_replays.log
timeout 300 python tools/gpu_perf_voxelgrid.py > $OUT/perf_vg.log 2>&1; tail -6 $OUT/perf_vg.log
