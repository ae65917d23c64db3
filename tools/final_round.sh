#!/bin/bash
# Dev tool (GPU box, through gpurun): the round's closing run — GPU test suite, smoke, profile sets, the un-profiled bench lines,
# the per-config table.  Everything lands under gpurun_out/prof_out/.  Usage: bash tools/final_round.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; TAG="${1:-r04_v2}"; OUT="$REPO/gpurun_out/prof_out"
mkdir -p "$OUT"; cd "$REPO"
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > "$OUT/${TAG}_gputests.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> "$OUT/${TAG}_gputests.txt"
timeout 1500 bash tools/profile_round.sh "$TAG" > /dev/null 2>&1
timeout 400 python bench.py > "$OUT/r04_bench.json" 2> "$OUT/r04_bench.err"
timeout 400 python bench.py --pad 25 --steps 50 > "$OUT/r04_bench_pad25.json" 2> "$OUT/r04_bench_pad25.err"
timeout 400 python bench.py --pad 100 --steps 10 --warmup 2 --cpu-sample 0 --no-end-to-end > "$OUT/r04_bench_pad100.json" 2> "$OUT/r04_bench_pad100.err"
timeout 400 python bench.py --config 3 --steps 50 > "$OUT/r04_bench_config3.json" 2> "$OUT/r04_bench_config3.err"
timeout 400 python bench.py --config 4 --steps 50 > "$OUT/r04_bench_config4.json" 2> "$OUT/r04_bench_config4.err"
timeout 900 python tools/run_configs.py --out "$OUT/r04_configs.json" > "$OUT/r04_configs.log" 2>&1
timeout 600 python tools/probe_trans_bins.py "$OUT/r04_trans_bins.json" > "$OUT/r04_trans_bins.log" 2>&1
cat "$OUT/${TAG}_gputests.txt"
python - <<PY
import json
for f in ("r04_bench", "r04_bench_pad25", "r04_bench_pad100", "r04_bench_config3", "r04_bench_config4"):
    try:
        d = json.loads(open("$OUT/" + f + ".json").read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r.get("kernel_ms_per_launch", r.get("kernel_ms_per_step")), r.get("prepass_ms_per_launch", r.get("prepass_ms_per_step")),
              r["frac"], r["frac_is"], r.get("lds_frac"), r.get("traffic"), (d.get("cpu_baseline") or {}).get("gpu_matches_oracle_on_sample"))
    except Exception as e:
        print(f, "ERR", e)
PY
