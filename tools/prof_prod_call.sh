# usage: bash tools/prof_prod_call.sh <tag> ["ENV=1 ENV2=0"]  -- kernel + copy trace of the adapter's production call (tools/gpu_prod_ivox.py), read launch by launch
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-prod}; shift || true
for kv in ${1:-}; do export $kv; done
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/gpu_prod_ivox.py 24 2>&1 | tail -1 > $OUT/prod_untraced.log
rm -rf /tmp/p_prod; rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d /tmp/p_prod -- python $R/tools/gpu_prod_ivox.py 24 > $OUT/prod_under_trace.log 2>&1
K=$(find /tmp/p_prod -name "*kernel_trace.csv" | head -1); M=$(find /tmp/p_prod -name "*memory_copy_trace.csv" | head -1)
python $R/tools/trace_sequence.py $K $M --skip 0.6 --n 60 > $OUT/prod_sequence.txt 2>&1
python $R/tools/trace_timeline.py $K > $OUT/prod_timeline.txt 2>&1
cat $OUT/prod_untraced.log
