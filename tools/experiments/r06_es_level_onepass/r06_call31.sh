set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_af}; mkdir -p $OUT
cd $R
FLS_FUZZ_VERBOSE=1 timeout 600 python tools/es_fuzz.py 200 777 2 700000 > $OUT/fuzz_mode2.log 2>&1
tail -5 $OUT/fuzz_mode2.log | cut -c1-300
FLS_FUZZ_VERBOSE=1 timeout 600 python tools/es_fuzz.py 200 778 2 700000 > $OUT/fuzz_mode2b.log 2>&1
tail -5 $OUT/fuzz_mode2b.log | cut -c1-300
