#!/bin/bash
# Dev tool (GPU box, through gpurun): the round's profile sets — rocprofv3 stats + PMC passes of bench.py for the headline workload
# and for --pad 25 (K1w), condensed by tools/summarize_profiles.py; results are copied under gpurun_out/prof_out/ (what gpurun
# brings back).  Usage: bash tools/profile_round.sh <tag>   e.g. r04_v1
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r04_v1}"
mkdir -p "$REPO/gpurun_out/prof_out"
for W in "pad10:" "pad25:--pad 25"; do
  NAME="${W%%:*}"; EXTRA="${W#*:}"
  rm -rf "$REPO/gpurun_out/prof"
  BENCH_ARGS="--steps 10 --warmup 2 --cpu-sample 0 --no-end-to-end $EXTRA" timeout 900 bash "$REPO/tools/profile_bench.sh" > /dev/null 2>&1
  T="$TAG"; [ "$NAME" = "pad25" ] && T="${TAG}_k1w_pad25"
  (cd "$REPO" && timeout 300 python tools/summarize_profiles.py "$T" > "gpurun_out/prof_out/${T}_summary.txt" 2>&1)
  cp "$REPO"/profiles/${T}_* "$REPO/profiles/traffic.json" "$REPO/gpurun_out/prof_out/" 2>/dev/null
done
rm -rf "$REPO/gpurun_out/prof"
ls -la "$REPO/gpurun_out/prof_out"
