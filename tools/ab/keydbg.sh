#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
export TMPDIR=/tmp
run() {
  rm -rf /tmp/prof1
  (cd /tmp && env $1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $R/tools/k1_probe.py --variants 0 --reps 6 > /tmp/k1.log 2>&1)
  tail -1 /tmp/k1.log | cut -c1-200
  python - <<PY
import pandas as pd, glob
f = glob.glob("/tmp/prof1/*/*_kernel_stats.csv")[0]
d = pd.read_csv(f)
d = d[d.Name.str.contains("bin_|key_kernel|table_kernel|publish")]
print("$1", [(n.replace("pup::","").replace("void ","")[:14], int(c), round(a/1e3,1)) for n,c,a in zip(d.Name, d.Calls, d.AverageNs)])
PY
}
run COOLPUPPY_AMD_KEY_DEBUG=0
run COOLPUPPY_AMD_KEY_DEBUG=27
run COOLPUPPY_AMD_BIN_DH=11
run COOLPUPPY_AMD_BIN_DH=9
