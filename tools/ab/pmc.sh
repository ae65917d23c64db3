#!/bin/bash
# instruction mix / activity of the prepass kernels (counters in passes of their own, kernel trace only)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
export TMPDIR=/tmp
i=0
for C in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
         "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAVES SQ_LDS_BANK_CONFLICT"; do
  i=$((i+1)); rm -rf /tmp/pm$i
  (cd /tmp && timeout 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pm$i -- python $R/tools/k1_probe.py --variants 0 --reps 3 > /tmp/pm$i.log 2>&1)
done
python - <<PY
import pandas as pd, glob
for f in sorted(glob.glob("/tmp/pm*/*/*_counter_collection.csv")):
    d = pd.read_csv(f)
    d = d[d.Kernel_Name.str.contains("bin_|key_kernel|table_kernel")]
    d["k"] = d.Kernel_Name.str.replace("pup::","").str.replace("void ","").str[:16]
    print(d.groupby(["k","Counter_Name"]).Counter_Value.mean().unstack().round(0).to_string())
PY
