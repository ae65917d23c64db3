#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"; R=$PWD
export TMPDIR=/tmp
run() {
  rm -rf /tmp/prof1
  (cd /tmp && env $2 timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof1 -- python $R/tools/k1_probe.py --variants 0 --reps 6 $3 > /tmp/k1.log 2>&1)
  python - <<PY
import pandas as pd, glob
f = glob.glob("/tmp/prof1/*/*_kernel_stats.csv")[0]
d = pd.read_csv(f)
d = d[d.Name.str.contains("bin_|key_kernel")]
print("$1 $2 $3", [(n.replace("pup::","").replace("void ","")[:14], int(c), round(a/1e3,1)) for n,c,a in zip(d.Name, d.Calls, d.AverageNs)], [l for l in open("/tmp/k1.log").read().strip().splitlines() if l.startswith("{")][-1][-75:])
PY
}
run new A=0
run new A=0 "--pad 25"
timeout 600 python -m pytest tests/test_kernel_parity.py tests/test_properties_gpu.py tests/test_scale_gpu.py -q -m gpu -x 2>&1 | tail -2
timeout 300 python bench.py --steps 60 --warmup 5 --no-end-to-end --cpu-sample 0 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('bench', d['ms_per_step'], r.get('kernel_ms_per_launch'), r.get('prepass_ms_per_launch'))"
