#!/bin/bash
# Dev tool (GPU box): rocprofv3 passes over BASELINE configs[4] (trans pile-up, sparse kernel K1s) through tools/probe_trans.py.
# The synthetic table is built ONCE outside the profiler (its builder forks workers; rocprofv3 hung waiting for them in round 3)
# and every pass runs under its own timeout.
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"
OUT="$REPO/gpurun_out/prof_trans"
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
export TRANS_CACHE=/tmp/trans_table.npz
TRANS_CACHE_ONLY=1 python "$REPO/tools/probe_trans.py" 100 > "$OUT/build.log" 2>&1
timeout 240 rocprofv3 -L > "$OUT/counters_available.txt" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats" -- python "$REPO/tools/probe_trans.py" 100 > "$OUT/stats.log" 2>&1
for C in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "SQ_WAVES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" \
         "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_ATOMIC_WITHOUT_RET_REQ_sum" "TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum TD_TD_BUSY_sum"; do
  tag=$(echo "$C" | tr ' ' '_' | cut -c1-60)
  timeout 300 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/pmc_$tag" -- python "$REPO/tools/probe_trans.py" 100 > "$OUT/pmc_$tag.log" 2>&1
done
python - <<PY
import glob, json, pandas as pd
out = {}
st = glob.glob("$OUT/stats/*/*_kernel_stats.csv")
if st:
    d = pd.read_csv(st[0]); d = d[d["Name"].str.contains("pileup_sparse_kernel|reduce_partials")]
    out["kernel_stats"] = d[["Name", "Calls", "AverageNs", "MinNs", "MaxNs"]].to_dict("records")
for f in glob.glob("$OUT/pmc_*/*/*_counter_collection.csv"):
    d = pd.read_csv(f); d = d[d["Kernel_Name"].str.contains("pileup_sparse_kernel")]
    for name, g in d.groupby("Counter_Name"):
        out.setdefault("counters_mean_per_launch", {})[name] = float(g.groupby("Dispatch_Id")["Counter_Value"].sum().mean())
    if len(d):
        out["vgpr"] = int(d.VGPR_Count.iloc[0]); out["lds_bytes"] = int(d.LDS_Block_Size.iloc[0]); out["grid"] = int(d.Grid_Size.iloc[0])
json.dump(out, open("$OUT/trans_summary.json", "w"), indent=1)
print(json.dumps(out)[:3000])
PY
