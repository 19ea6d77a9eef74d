#!/bin/bash
# Evidence of a round-6 state in ONE gpurun call: the full GPU test suite, smoke, the full bench line, the regression guard against the previous round's
# record (tools/bench_diff.py), the large-cloud VoxelGrid table.  usage: bash tools/final_round6.sh <tag>  -> gpurun_out/<tag>/ (copied to profiles/<tag>_*)
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
TAG=${1:-r06_z}
OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/gpu_pytest.log
timeout 200 python __graft_entry__.py smoke > $OUT/smoke.log 2>&1
timeout 900 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
python tools/bench_diff.py $OUT/bench_full.json > $OUT/bench_diff.txt 2>&1
timeout 200 python tools/gpu_vg_large.py 8 > $OUT/vg_large.json 2> $OUT/vg_large.err
tail -3 $OUT/gpu_pytest.log; tail -1 $OUT/smoke.log; tail -c 300 $OUT/bench_full.err; cat $OUT/bench_diff.txt; cat $OUT/vg_large.json
