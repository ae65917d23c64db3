#!/bin/bash
# Dev tool (GPU box, through gpurun): the round's closing run on the FINAL sources — GPU test suite, smoke, the profile sets of every
# kernel family (tools/profile_round.sh -> profiles/traffic.json), the un-profiled bench lines (which then quote the measured traffic),
# the per-config table, by-window / coverage / host-layer / phase-clock probes.  Everything lands under gpurun_out/prof_out/.
# Usage: bash tools/final_round.sh <tag>
set -u
REPO="${GRAFT_REPO_ROOT:-/root/repo}"; TAG="${1:-r06}"; OUT="$REPO/gpurun_out/prof_out"
mkdir -p "$OUT"; cd "$REPO"
timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -E "passed|failed|error" | tail -3 > "$OUT/${TAG}_gputests.txt"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 >> "$OUT/${TAG}_gputests.txt"
timeout 2400 bash tools/profile_round.sh "$TAG" > "$OUT/${TAG}_profile_round.log" 2>&1
timeout 400 python bench.py > "$OUT/${TAG}_bench.json" 2> "$OUT/${TAG}_bench.err"
timeout 400 python bench.py --pad 25 --steps 50 > "$OUT/${TAG}_bench_pad25.json" 2> "$OUT/${TAG}_bench_pad25.err"
timeout 400 python bench.py --pad 100 --steps 10 --warmup 2 --cpu-sample 0 --no-end-to-end > "$OUT/${TAG}_bench_pad100.json" 2> "$OUT/${TAG}_bench_pad100.err"
timeout 400 python bench.py --config 3 --steps 50 > "$OUT/${TAG}_bench_config3.json" 2> "$OUT/${TAG}_bench_config3.err"
timeout 400 python bench.py --config 4 --steps 50 > "$OUT/${TAG}_bench_config4.json" 2> "$OUT/${TAG}_bench_config4.err"
timeout 400 python bench.py --variant 16 --steps 20 --cpu-sample 0 --no-end-to-end > "$OUT/${TAG}_bench_k1r.json" 2> "$OUT/${TAG}_bench_k1r.err"
timeout 900 python tools/run_configs.py --out "$OUT/${TAG}_configs.json" > "$OUT/${TAG}_configs.log" 2>&1
timeout 400 python tools/probe_bywindow.py --out "$OUT/${TAG}_bywindow.json" > "$OUT/${TAG}_bywindow.log" 2>&1
timeout 400 python tools/probe_coverage.py > "$OUT/${TAG}_coverage.txt" 2>&1
timeout 300 python tools/host_profile.py --plain --top 30 > "$OUT/${TAG}_host_plain.txt" 2>&1
timeout 300 python tools/host_profile.py --top 30 > "$OUT/${TAG}_host_grouped.txt" 2>&1
timeout 300 python tools/k1_probe.py --variants 0,128,67108864,67108992 --reps 7 --out "$OUT/${TAG}_k1q_probe.json" > "$OUT/${TAG}_k1q_phases.txt" 2>&1
timeout 300 python tools/probe_rescale.py > "$OUT/${TAG}_rescale.txt" 2>&1
cat "$OUT/${TAG}_gputests.txt"
python - <<PY
import json
for f in ("bench", "bench_pad25", "bench_pad100", "bench_config3", "bench_config4", "bench_k1r"):
    try:
        d = json.loads(open("$OUT/${TAG}_" + f + ".json").read().strip().splitlines()[-1]); r = d["roofline"]
        print(f, d["value"], d["ms_per_step"], r.get("kernel_ms_per_launch", r.get("kernel_ms_per_step")), r.get("prepass_ms_per_launch", r.get("prepass_ms_per_step")),
              r["frac"], r["frac_is"], r.get("lds_frac"), r.get("traffic"), (d.get("cpu_baseline") or {}).get("gpu_matches_oracle_on_sample"), (d.get("end_to_end") or {}).get("pileup_wall_s"))
    except Exception as e:
        print(f, "ERR", e)
PY
