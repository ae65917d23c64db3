#!/bin/bash
# Dev tool (GPU box): timing experiments on the progressive staged kernel (debug bits: 24 = no bucket boundaries / one segment,
# 25 = no staging, 30 = LDS reads in the unsorted order; bit 22 = the progressive form, else the barrier form).  Sums are wrong with bits 25 / 30.
OUT=gpurun_out/k1q_prog; mkdir -p $OUT
timeout 200 python tools/k1_probe.py --variants ${VARIANTS:-4194304,37748736,1111490560,54525952,1128267776,33554432} --reps 5 > $OUT/debug.txt 2>&1
grep -v phases $OUT/debug.txt | cut -c1-300
