#!/bin/bash
# Dev tool: PMC passes over tools/probe_perf.py (K1 instruction mix / stalls). Usage: pmc_probe.sh "<probe args>"
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/pmc_probe"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ARGS="${1:---chroms 23 --lam 4200 --pairs 1000000 --reps 2}"
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
         "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python "$REPO/tools/probe_perf.py" $ARGS > "$OUT/p$i.log" 2>&1
done
python - <<PY
import pandas as pd, glob
for f in sorted(glob.glob("$OUT/p*/*/*_counter_collection.csv")):
    d=pd.read_csv(f); k=d[d.Kernel_Name.str.contains("pileup_")]
    print(k.groupby("Counter_Name").Counter_Value.mean().to_string())
PY
