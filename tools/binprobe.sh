cd /tmp && export TMPDIR=/tmp
for dbg in 0; do
  rm -rf /tmp/prof1
  COOLPUPPY_AMD_BIN_DEBUG=$dbg timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $GRAFT_REPO_ROOT/tools/k1_probe.py --variants 0 --reps 6 > /dev/null 2>&1
  python - <<PY
import pandas as pd, glob
f = glob.glob("/tmp/prof1/*/*_kernel_stats.csv")[0]
d = pd.read_csv(f)
d = d[d.Name.str.contains("bin_|key_kernel|table|publish|fillBuffer")]
print("dbg $dbg", [(n[5:25], round(a/1e3,1)) for n,a in zip(d.Name, d.AverageNs)])
PY
done
