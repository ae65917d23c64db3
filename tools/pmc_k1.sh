#!/bin/bash
# Dev tool (GPU box): PMC passes over tools/k1_probe.py (instruction mix / stalls / LDS conflicts / HBM bytes of the
# pile-up kernels).  Usage: pmc_k1.sh "<k1_probe args>" [tag]
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
TAG="${2:-pmc_k1}"
OUT="$REPO/gpurun_out/$TAG"
mkdir -p "$OUT"; cd /tmp; export TMPDIR=/tmp
ARGS="${1:---variants 0 --reps 2}"
python "$REPO/tools/k1_probe.py" --variants 16 --reps 1 > /dev/null 2>&1     # build the workload cache once
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_WAVES SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT" \
         "SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_VALU_INT32 SQ_INSTS_LDS" \
         "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/p$i" -- python "$REPO/tools/k1_probe.py" $ARGS > "$OUT/p$i.log" 2>&1
done
python - <<PY
import pandas as pd, glob, json
res = {}
for f in sorted(glob.glob("$OUT/p*/*/*_counter_collection.csv")):
    d = pd.read_csv(f)
    k = d[d.Kernel_Name.str.contains("pileup_|radix|block_key|permute|count_changes|balance_pixels")]
    for (kn, cn), g in k.groupby(["Kernel_Name", "Counter_Name"]):
        short = kn.split("(")[0].replace("void ", "")[:70]
        res.setdefault(short, {})[cn] = float(g.Counter_Value.mean())
        res[short]["VGPR"] = int(g.VGPR_Count.iloc[0]); res[short]["LDS"] = int(g.LDS_Block_Size.iloc[0])
json.dump(res, open("$OUT/summary.json", "w"), indent=1)
for kn, v in res.items():
    if "pileup_" in kn: print(kn, json.dumps(v))
PY
rm -rf "$OUT"/p*/
