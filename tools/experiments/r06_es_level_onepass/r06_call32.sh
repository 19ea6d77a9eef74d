set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_ag}; mkdir -p $OUT
cd $R
FLS_ES_DEBUG=1 timeout 300 python tools/es_level_stamps.py 115200 > $OUT/level_stamps.log 2>&1
grep "level 1," $OUT/level_stamps.log | cut -c1-300

cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/p_trace; timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/p_trace -- python $R/tools/gpu_vg_large.py 8 scan > $OUT/under_trace.log 2> $OUT/trace.err
python $R/tools/trace_sequence.py $(find /tmp/p_trace -name "*kernel_trace.csv" | head -1) --skip 0.6 --n 30 > $OUT/sequence.txt 2>&1
cat $OUT/sequence.txt
