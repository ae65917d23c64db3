#!/bin/bash
# Dev tool (GPU box): block costs (COOLPUPPY_AMD_WIDE_COST / _K1Q_COST) and K1w quad shares (COOLPUPPY_AMD_WIDE_SHARES), tools/k1_probe.py timings.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/wide_cost.txt; : > $OUT
for pad in 100 40 25 16; do
  echo "== pad $pad default: $(timeout 300 python tools/k1_probe.py --variants 0 --reps 5 --pad $pad 2>/dev/null | grep k1_ms | cut -c1-110)" >> $OUT
done
for sh in "92,166,222" "96,170,224"; do
  for pad in 100 40; do
  echo "== pad $pad shares $sh: $(COOLPUPPY_AMD_WIDE_SHARES=$sh timeout 300 python tools/k1_probe.py --variants 0 --reps 5 --pad $pad 2>/dev/null | grep k1_ms | cut -c1-110)" >> $OUT
  done
done
for cost in 250 400 600 900; do
  echo "== k1q pad 10 cost $cost: $(COOLPUPPY_AMD_K1Q_COST=$cost timeout 300 python tools/k1_probe.py --variants 0 --reps 7 2>/dev/null | grep k1_ms | cut -c1-110)" >> $OUT
  echo "== k1q pad 5 cost $cost: $(COOLPUPPY_AMD_K1Q_COST=$cost timeout 300 python tools/k1_probe.py --variants 0 --reps 7 --pad 5 2>/dev/null | grep k1_ms | cut -c1-110)" >> $OUT
  echo "== k1q pad 15 cost $cost: $(COOLPUPPY_AMD_K1Q_COST=$cost timeout 300 python tools/k1_probe.py --variants 0 --reps 7 --pad 15 2>/dev/null | grep k1_ms | cut -c1-110)" >> $OUT
  echo "== k1q 8 groups cost $cost: $(COOLPUPPY_AMD_K1Q_COST=$cost timeout 300 python tools/k1_probe.py --variants 0 --reps 7 --tiles 8 2>/dev/null | grep k1_ms | cut -c1-110)" >> $OUT
done
cat $OUT
