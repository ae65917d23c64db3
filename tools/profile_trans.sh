#!/bin/bash
# Dev tool (GPU box): rocprofv3 passes over BASELINE configs[4] (trans pile-up, sparse kernel K1s) through tools/run_configs.py.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_trans"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$REPO/tools/run_configs.py" --only 4 --out "$OUT/c4.json" > "$OUT/stats.log" 2>&1
for C in FETCH_SIZE "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD"; do
  tag=$(echo "$C" | tr ' ' '_')
  rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -- python "$REPO/tools/run_configs.py" --only 4 --out "$OUT/c4_$tag.json" > "$OUT/pmc_$tag.log" 2>&1
done
python - <<PY
import glob, json, pandas as pd
out = {}
st = glob.glob("$OUT/stats/*/*_kernel_stats.csv")
if st:
    d = pd.read_csv(st[0]); d = d[d["Name"].str.contains("pileup_sparse_kernel")]
    out["kernel_stats"] = d[["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]].to_dict("records")
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    d = pd.read_csv(f); d = d[d["Kernel_Name"].str.contains("pileup_sparse_kernel")]
    for name, g in d.groupby("Counter_Name"):
        out.setdefault("counters_mean_per_launch", {})[name] = float(g.groupby("Dispatch_Id")["Counter_Value"].sum().mean())
json.dump(out, open("$OUT/trans_summary.json", "w"), indent=1)
print(json.dumps(out)[:1500])
PY
