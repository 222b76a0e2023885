#!/bin/bash
# Copies what is worth keeping of a tools/profile_gpu.sh run (gpurun_out/prof/TAG, scratch) into profiles/ (tracked):
#   tools/keep_profile.sh TAG NAME PMC_NAME   ->  profiles/NAME_summary.txt, NAME_kernel_stats*.csv, NAME_bench.json,
#                                                 profiles/pmc_PMC_NAME.json (what bench.py reads)
set -e
SRC=gpurun_out/prof/$1; NAME=$2; PMC=$3
cp "$SRC/summary.txt" "profiles/${NAME}_summary.txt"
cp "$(find "$SRC/kt" -name '*kernel_stats.csv' | head -1)" "profiles/${NAME}_kernel_stats.csv"
cp "$(find "$SRC/kt_alone" -name '*kernel_stats.csv' | head -1)" "profiles/${NAME}_kernel_stats_one_batch_at_a_time.csv"
cp "$SRC/kt.json" "profiles/${NAME}_bench_under_rocprof.json"
cp "$SRC/pmc.json" "profiles/pmc_${PMC}.json"
ls -la profiles | grep -E "${NAME}|pmc_${PMC}"
