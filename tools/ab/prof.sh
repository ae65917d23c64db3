#!/bin/bash
# per-kernel averages of the prepass for the new and the old build on the same box
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
cp coolpuppy_amd/libpup_hip.so /tmp/new.so
export TMPDIR=/tmp
for which in new old new old; do
  if [ $which = new ]; then cp /tmp/new.so coolpuppy_amd/libpup_hip.so; else cp tools/ab/libpup_hip_old.so coolpuppy_amd/libpup_hip.so; fi
  rm -rf /tmp/prof1
  (cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $R/tools/k1_probe.py --variants 0 --reps 8 > /dev/null 2>&1)
  python - <<PY
import pandas as pd, glob
f = glob.glob("/tmp/prof1/*/*_kernel_stats.csv")[0]
d = pd.read_csv(f)
d = d[d.Name.str.contains("bin_|key_kernel|table|publish|fillBuffer|reduce_staged|add_counts|pileup_staged")]
print("$which", [(n.replace("pup::","").replace("void ","")[:16], int(c), round(a/1e3,1)) for n,c,a in zip(d.Name, d.Calls, d.AverageNs)])
PY
done
cp /tmp/new.so coolpuppy_amd/libpup_hip.so
