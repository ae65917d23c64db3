#!/bin/bash
# dev: per-kernel averages of the prepass under rocprofv3 for a few settings (k1_probe workload)
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
export TMPDIR=/tmp
run() {
  rm -rf /tmp/prof1
  (cd /tmp && env $1 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $R/tools/k1_probe.py --variants 0 --reps 6 $2 > /tmp/k1.log 2>&1)
  python - <<PY
import pandas as pd, glob
f = glob.glob("/tmp/prof1/*/*_kernel_stats.csv")[0]
d = pd.read_csv(f)
d = d[d.Name.str.contains("bin_|key_kernel")]
print("$1 $2", [(n.replace("pup::","").replace("void ","")[:14], int(c), round(a/1e3,1)) for n,c,a in zip(d.Name, d.Calls, d.AverageNs)], [l for l in open("/tmp/k1.log").read().strip().splitlines() if l.startswith("{")][-1][-75:])
PY
}
run A=0 "--pad 25"
run COOLPUPPY_AMD_BUCKET_WAVES=8 "--pad 25"
run COOLPUPPY_AMD_BUCKET_WAVES=2 "--pad 25"
run A=0 "--pad 100 --pairs 300000"
run A=0 ""
