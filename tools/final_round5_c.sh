#!/bin/bash
# Last gpurun call of round 5, on the tree with the late per-point stores (kernels_ivox_coop.hpp / kernels_grid_coop.hpp): the bit-identity
# check against the signatures recorded on the tree that passed the full GPU suite, the full bench line, a slice of the parity tests.
# Fills gpurun_out/r05_late/.
set -u
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/r05_late; mkdir -p $OUT
cd $R
timeout 60 python tools/gpu_ab_fanin.py check reps=120 2>&1 | tee $OUT/signature_check.log | cut -c1-240
timeout 125 python bench.py --steps 20 --warmup 5 > $OUT/bench_full.json 2> $OUT/bench_full.err
head -c 260 $OUT/bench_full.json; echo; tail -c 200 $OUT/bench_full.err
timeout 60 python -m pytest tests/test_gpu_parity.py -x -q -k "config2_p2plane_ivox or config1_icp or config4_loam_full or determinism or empty_and_tiny" 2>&1 | tail -4 | tee $OUT/gpu_pytest_slice.log
