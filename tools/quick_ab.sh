#!/bin/bash
# tools/quick_ab.sh OUTDIR lib1.so lib2.so ...   ("default" = the product build): Box and Veggie at 1080p, two rounds each
OUT=$1; shift; mkdir -p gpurun_out/$OUT
timeout -k 10 600 tools/ab_variants.sh -- "$@" > gpurun_out/$OUT/ab_box.log 2>&1; cat gpurun_out/$OUT/ab_box.log
timeout -k 10 600 tools/ab_variants.sh --scene Veggie -- "$@" > gpurun_out/$OUT/ab_veggie.log 2>&1; cat gpurun_out/$OUT/ab_veggie.log
