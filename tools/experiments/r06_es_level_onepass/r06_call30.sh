set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_ae}; mkdir -p $OUT
cd $R
FLS_ES_DEBUG=1 timeout 300 python tools/es_level_stamps.py 115200 1555200 > $OUT/level_stamps.log 2>&1
grep -B16 "rep 1" $OUT/level_stamps.log | cut -c1-400 | head -60
timeout 900 python -m pytest tests/test_gpu_exact_sort.py -q -x 2>&1 | tail -15 > $OUT/pytest_sort.log; tail -8 $OUT/pytest_sort.log
