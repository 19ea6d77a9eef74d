set -u
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/${1:-r06_ai}; mkdir -p $OUT
cd $R
timeout 300 python tools/gpu_ab_ndt.py "FLS_FUSED_TAIL=0" "FLS_FUSED_TAIL=1" "FLS_FUSED_TAIL=0" "FLS_FUSED_TAIL=1" > $OUT/ab_ndt_fused.log 2>&1; cat $OUT/ab_ndt_fused.log
timeout 1200 python tools/es_fuzz.py 1500 31337 2 1600000 > $OUT/es_fuzz_mode2_1500.log 2>&1; tail -1 $OUT/es_fuzz_mode2_1500.log
timeout 1500 python tools/gpu_vs_ref_fuzz.py mapping 80 320 > $OUT/vs_ref_mapping_80_399.log 2>&1; tail -1 $OUT/vs_ref_mapping_80_399.log
timeout 900 python tools/gpu_vs_ref_fuzz.py loc 40 40 > $OUT/vs_ref_loc_40_79.log 2>&1; tail -1 $OUT/vs_ref_loc_40_79.log
timeout 900 python tools/gpu_vs_ref_fuzz.py long 12 48 > $OUT/vs_ref_long_12_59.log 2>&1; tail -1 $OUT/vs_ref_long_12_59.log
