#!/bin/bash
# Dev tool (run on the GPU box through gpurun): rocprofv3 passes over one command (default: bench.py $BENCH_ARGS; PROFILE_CMD overrides).
#   pass 1: --kernel-trace --stats   -> per-kernel time summary
#   pass 2..: --pmc <counters>       -> HBM traffic / L2 hit counters (separate passes, as the guide prescribes; never with a trace domain
#                                       other than the kernel trace)
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof"
ARGS="${BENCH_ARGS:---steps 10 --warmup 2 --cpu-sample 0}"
CMD="${PROFILE_CMD:-python $REPO/bench.py $ARGS}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
[ -z "${PROFILE_CMD:-}" ] && python "$REPO/bench.py" $ARGS --steps 1 --warmup 0 > /dev/null 2>&1   # build the workload cache once
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- $CMD > "$OUT/stats_bench.json" 2> "$OUT/stats.err"
for C in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT"; do
  tag=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -- $CMD > "$OUT/pmc_${tag}_bench.json" 2> "$OUT/pmc_$tag.err"
done
du -sh "$OUT"
