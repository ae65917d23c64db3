"""Dev tool: condense gpurun_out/prof (rocprofv3 passes of bench.py, see tools/profile_bench.sh) into profiles/.

Writes profiles/<tag>_kernel_stats.csv (the --stats summary), profiles/<tag>_pmc_summary.json (per-kernel
counter means) and profiles/traffic.json (HBM bytes per K1 launch, used by bench.py's roofline.traffic).
FETCH_SIZE calibration: balance_pixels_kernel streams a KNOWN byte count (8 B/pixel read, 8 B/pixel written)
in the same runs, so its counters give the bytes-per-count factor for this access width on this machine.
"""
import glob
import json
import os
import shutil
import sys

import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GENERIC = "--generic" in sys.argv          # any command under the profiler (no bench line): kernel stats + counter means only
sys.argv = [x for x in sys.argv if x != "--generic"]
tag = sys.argv[1] if len(sys.argv) > 1 else "r02"
prof = os.path.join(ROOT, "gpurun_out", "prof")
out = os.path.join(ROOT, "profiles")
os.makedirs(out, exist_ok=True)

stats = glob.glob(os.path.join(prof, "stats", "*", "*_kernel_stats.csv"))[0]
shutil.copy(stats, os.path.join(out, f"{tag}_kernel_stats.csv"))
bench = None
if not GENERIC:
    bench = json.loads(open(os.path.join(prof, "stats_bench.json")).read().strip().splitlines()[-1])
    nnz = bench["config"]["nnz"]

summary = {}
for d in sorted(glob.glob(os.path.join(prof, "pmc_*"))):
    if not os.path.isdir(d):
        continue
    f = glob.glob(os.path.join(d, "*", "*_counter_collection.csv"))
    if not f:
        continue
    df = pd.read_csv(f[0])
    tr = pd.read_csv(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv"))[0])
    tr["ms"] = (tr.End_Timestamp - tr.Start_Timestamp) / 1e6
    for kname, g in df.groupby("Kernel_Name"):
        if "pup::" not in kname:
            continue
        short = kname.split("(")[0].replace("void ", "")
        e = summary.setdefault(short, {"counters": {}})
        for cname, gg in g.groupby("Counter_Name"):
            e["counters"][cname] = {"mean_per_launch": float(gg.Counter_Value.mean()), "launches": int(len(gg))}
        e["ms_per_launch_under_pmc"] = float(tr[tr.Kernel_Name == kname].ms.mean())
        e["vgpr"] = int(g.VGPR_Count.iloc[0]); e["agpr"] = int(g.Accum_VGPR_Count.iloc[0])
        e["lds_bytes"] = int(g.LDS_Block_Size.iloc[0])

if GENERIC:
    def _mix(cs):
        g = lambda n: cs.get(n, {}).get("mean_per_launch")      # noqa: E731
        out = {}
        if g("SQ_INSTS_VALU"):
            out["salu_over_valu"] = round(g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU"), 3) if g("SQ_INSTS_SALU") else None
            out["wait_any_over_wave_cycles"] = round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3) if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES") else None
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum"):
            out["l2_hit"] = round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 3)
        if g("FETCH_SIZE") is not None and g("WRITE_SIZE") is not None:
            out["hbm_bytes_per_launch_x2_rule"] = g("FETCH_SIZE") * 1024 * 2.0 + g("WRITE_SIZE") * 1024      # (guide: gfx950 FETCH_SIZE counts wide reads at half)
        return out
    for k in summary:
        summary[k]["derived"] = _mix(summary[k]["counters"])
    note = open(os.path.join(prof, "stats_bench.json")).read()[-4000:] if os.path.exists(os.path.join(prof, "stats_bench.json")) else ""
    json.dump({"command_output_tail": note, "kernels": summary, "units": "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB"},
              open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps({k: v["derived"] for k, v in summary.items() if "pileup" in k or "reduce" in k}, indent=1))
    sys.exit(0)

# the pile-up step may be served by two kernels (block-staged for dense tiles + plain register tile for sparse ones):
# traffic and time of "K1" are the sums over the pile-up kernels launched once per step
k1s = [k for k in summary if "pileup_" in k]
# ... of the timed loop: bench.py also runs ONE statistics-gathering step through another instantiation of the kernel
k1s = [k for k in k1s if "FETCH_SIZE" in summary[k]["counters"] and "WRITE_SIZE" in summary[k]["counters"]]
most = max((summary[k]["counters"]["FETCH_SIZE"]["launches"] for k in k1s), default=0)
k1s = [k for k in k1s if summary[k]["counters"]["FETCH_SIZE"]["launches"] * 2 > most]
k1 = " + ".join(k1s)
cal = next((k for k in summary if "balance_pixels" in k), None)
doc = {"bench_line_under_rocprof": bench, "kernels": summary,
       "units": "FETCH_SIZE / WRITE_SIZE are reported by rocprofv3 in KiB"}
fetch_kib = sum(summary[k]["counters"]["FETCH_SIZE"]["mean_per_launch"] for k in k1s)
write_kib = sum(summary[k]["counters"]["WRITE_SIZE"]["mean_per_launch"] for k in k1s)
factor = 2.0      # MI355X_MICROARCH.md: gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads
if cal:
    known_read = nnz * 8.0
    cal_fetch = summary[cal]["counters"]["FETCH_SIZE"]["mean_per_launch"] * 1024
    cal_write = summary[cal]["counters"]["WRITE_SIZE"]["mean_per_launch"] * 1024
    doc["calibration"] = {"kernel": cal, "known_read_bytes": known_read, "known_write_bytes": nnz * 8.0,
                          "FETCH_SIZE_bytes": cal_fetch, "WRITE_SIZE_bytes": cal_write,
                          "read_bytes_per_counted_byte": known_read / cal_fetch,
                          "write_bytes_per_counted_byte": nnz * 8.0 / cal_write}
    factor = known_read / cal_fetch
hbm = fetch_kib * 1024 * factor + write_kib * 1024
rl = bench.get("roofline", {})
doc["k1"] = {"kernel": k1, "FETCH_SIZE_KiB": fetch_kib, "WRITE_SIZE_KiB": write_kib, "fetch_correction_factor": factor,
             "hbm_bytes_per_launch": hbm,
             "algorithmic_bytes_per_launch": rl.get("algorithmic_bytes_per_launch", rl.get("algorithmic_bytes_per_step"))}
# instruction mix and waiting of the pile-up kernel(s): what the verdicts quote (SALU / VALU, waves waiting, LDS conflicts)
mix = {}
for k in k1s:
    cs = summary[k]["counters"]
    g = lambda n: cs.get(n, {}).get("mean_per_launch")      # noqa: E731
    if g("SQ_INSTS_VALU"):
        mix[k] = {"salu_over_valu": round(g("SQ_INSTS_SALU") / g("SQ_INSTS_VALU"), 3) if g("SQ_INSTS_SALU") else None,
                  "wait_any_over_wave_cycles": round(g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES"), 3) if g("SQ_WAIT_ANY") and g("SQ_WAVE_CYCLES") else None,
                  "lds_bank_conflict_over_lds_active": round(g("SQ_LDS_BANK_CONFLICT") / g("SQ_ACTIVE_INST_LDS"), 4) if g("SQ_LDS_BANK_CONFLICT") is not None and g("SQ_ACTIVE_INST_LDS") else None,
                  "l2_hit": round(g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")), 3) if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") else None}
doc["k1"]["instruction_mix"] = mix
json.dump(doc, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
sys.path.insert(0, ROOT)
import bench as bench_mod                                     # workload / kernel-source keys exactly as bench.py computes them
c = bench["config"]
cfg = 3 if "configs[3]" in c["workload"] else (4 if "configs[4]" in c["workload"] else 2)
args = bench_mod.parse(["--pairs", str(c["pairs"]), "--nshifts", str(c["nshifts"]), "--pad", str(c["pad"]), "--config", str(cfg),
                        "--variant", str(c.get("variant", 0))] if cfg == 2 else ["--config", str(cfg), "--variant", str(c.get("variant", 0))])
entry = {"workload_key": bench_mod.traffic_key(args), "workload": c["workload"], "variant": c.get("variant", 0), "source_key": bench_mod.source_key(),
         "hbm_bytes_per_launch": hbm, "kernel": k1, "source": f"profiles/{tag}_pmc_summary.json",
         "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes), KiB->bytes, FETCH_SIZE x calibration "
                   "factor measured on balance_pixels_kernel (known 8 B/pixel stream) in the same run; quoted by "
                   "bench.py only while coolpuppy_amd/csrc is unchanged (source_key)"}
# one entry per measured workload (the default bench, --pad 25, ...): bench.py looks its own workload key up
tpath = os.path.join(out, "traffic.json")
book = {"entries": {}}
if os.path.exists(tpath):
    try:
        old = json.load(open(tpath))
        book = old if "entries" in old else {"entries": {old["workload_key"]: old}}
    except Exception:
        pass
book["entries"] = {k: v for k, v in book["entries"].items() if v.get("source_key") == entry["source_key"]}      # (other sources: void)
book["entries"][entry["workload_key"]] = entry
json.dump(book, open(tpath, "w"), indent=1)
print(json.dumps(doc["k1"], indent=1)); print(json.dumps(doc.get("calibration"), indent=1))
print(open(os.path.join(out, f"{tag}_kernel_stats.csv")).read()[:1500])
