#!/bin/bash
# Dev tool (GPU box): the progressive staged kernel (round 6 experiment, tuning bit 22) against the shipped barrier form: the probe first (short, under
# its own timeout: a wedged kernel shows here, not in a 15-minute test run), then parity / property tests, then phase clocks.
OUT=gpurun_out/k1q_prog; mkdir -p $OUT
timeout 120 python tools/k1_probe.py --variants 4194304,0,4194304,0 --reps 7 --out $OUT/probe.json > $OUT/probe.txt 2>&1
echo "probe rc $?"; cat $OUT/probe.txt | cut -c1-600
timeout 400 python -m pytest tests/test_properties_gpu.py tests/test_kernel_parity.py -q -m gpu -x --timeout 60 -k "staged or properties or kernel or index or block or sets or pair" 2>&1 | tail -8 > $OUT/tests.txt
cat $OUT/tests.txt
timeout 120 python tools/k1_probe.py --variants 71303168,67108864 --reps 5 > $OUT/phases.txt 2>&1
cat $OUT/phases.txt | cut -c1-1300
