#!/bin/bash
# Dev tool (GPU box, through gpurun): the round's profile sets — rocprofv3 stats + PMC passes for EVERY kernel family on the current
# sources: the headline (K1q), --pad 25 and --pad 100 (K1w), --config 3 (grouped K1q), --config 4 (K1s), --variant 16 (K1r on the
# headline windows) and the by-window pile-up (K1r + the many-tiles reduction), condensed by tools/summarize_profiles.py into
# profiles/<tag>_* and profiles/traffic.json (what bench.py quotes as roofline.traffic while csrc/ is unchanged); results are also
# copied under gpurun_out/prof_out/ (what gpurun brings back).  Usage: bash tools/profile_round.sh <tag> [set ...]   e.g. r05 k1q k1w_pad25
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${1:-r06}"; shift || true
SETS="${*:-k1q k1w_pad25 k1w_pad100 k1q_grouped k1s k1r bywindow}"
mkdir -p "$REPO/gpurun_out/prof_out"
for NAME in $SETS; do
  case "$NAME" in
    k1q)         EXTRA="" ;;
    k1w_pad25)   EXTRA="--pad 25" ;;
    k1w_pad100)  EXTRA="--pad 100" ;;
    k1q_grouped) EXTRA="--config 3" ;;
    k1s)         EXTRA="--config 4" ;;
    k1r)         EXTRA="--variant 16" ;;
    bywindow)    EXTRA="" ;;
    *) echo "unknown set $NAME"; continue ;;
  esac
  rm -rf "$REPO/gpurun_out/prof"
  T="${TAG}_${NAME}"
  if [ "$NAME" = "bywindow" ]; then
    PROFILE_CMD="python $REPO/tools/probe_bywindow.py --reps 2 --serial" timeout 900 bash "$REPO/tools/profile_bench.sh" > /dev/null 2>&1
    (cd "$REPO" && timeout 300 python tools/summarize_profiles.py --generic "$T" > "gpurun_out/prof_out/${T}_summary.txt" 2>&1)
    tail -c 2000 "$REPO/gpurun_out/prof/stats.err" > "$REPO/gpurun_out/prof_out/${T}_stats_err.txt" 2>/dev/null
  else
    BENCH_ARGS="--steps 10 --warmup 2 --cpu-sample 0 --no-end-to-end $EXTRA" timeout 900 bash "$REPO/tools/profile_bench.sh" > /dev/null 2>&1
    (cd "$REPO" && timeout 300 python tools/summarize_profiles.py "$T" > "gpurun_out/prof_out/${T}_summary.txt" 2>&1)
  fi
  cp "$REPO"/profiles/${T}_* "$REPO/profiles/traffic.json" "$REPO/gpurun_out/prof_out/" 2>/dev/null
done
rm -rf "$REPO/gpurun_out/prof"
ls -la "$REPO/gpurun_out/prof_out"
