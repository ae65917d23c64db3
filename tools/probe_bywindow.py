"""By-window pile-up at scale: one tile per feature — all 37 331 Bonev CTCF+ sites, pairs of sites 0.3 - 1 Mb
apart (each pair emitted once per side, coolpuppy/coolpup.py:1696-1755) — through pileup(), with the engine's share timed on its own.
Reports tiles, windows, wall, engine kernel times, kernel family, tiles/s and the size of the all-reduce message a multi-GPU run
would exchange.  Run on the GPU box:  python tools/probe_bywindow.py [--every 1] [--out profiles/r04_bywindow.json]"""
import argparse, gzip, json, os, sys, time, warnings
import numpy as np, pandas as pd
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coolpuppy_amd import coolpup
import synth
ap = argparse.ArgumentParser()
ap.add_argument("--every", type=int, default=1, help="use every k-th site")
ap.add_argument("--maxdist", type=int, default=1_000_000)
ap.add_argument("--out", default="")
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--profile", type=int, default=0, help="cProfile of one more call: print the top N functions")
ap.add_argument("--serial", action="store_true", help="build the table in this process (under rocprofv3: forked workers hang the profiler)")
a = ap.parse_args()
warnings.simplefilter("ignore")
clr = synth.make_cooler(synth.MM9, binsize=10_000, lam=120, seed=1000, name="mm9_like", parallel=not a.serial)
with gzip.open(os.path.join(ROOT, "tests", "golden", "ref_data", "Bonev_CTCF+.bed.gz"), "rt") as f:
    bed = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
bed = bed.iloc[::a.every].reset_index(drop=True)
kw = dict(features_format="bed", flank=100_000, by_window=True, mindist=300_000, maxdist=a.maxdist)
coolpup.pileup(clr, bed, **kw)
eng = next(iter(coolpup._ENGINES.values()))[1]
eng.set_profiling(3)
res = []
for rep in range(a.reps):
    eng.clear_stats()
    t = time.time(); df = coolpup.pileup(clr, bed, **kw); wall = time.time() - t
    st = eng.stats()
    nf, ni = eng.packed_sizes()
    rows = df[df["group"].astype(str) != "all"] if "group" in df else df
    res.append({"features": int(len(bed)), "tiles": int(eng.n_tiles), "rows_out": int(len(df)), "windows": int(st["snippets"]),
                "pileup_wall_s": round(wall, 3), "k1_ms": round(st["k1_ms"], 3), "reduce_ms": round(st["reduce_ms"], 3),
                "prepass_ms": round(st["prepare_ms"], 3), "launches": int(st["k1_launches"]), "kernel_family": eng.last_kernel(),
                "tiles_per_s_engine": round(eng.n_tiles / max((st["k1_ms"] + st["reduce_ms"]) * 1e-3, 1e-9)),
                "windows_per_s_engine": round(st["snippets"] / max((st["k1_ms"] + st["reduce_ms"]) * 1e-3, 1e-9)),
                # what a multi-GPU run exchanges: a flat all-reduce of every accumulator (round 4), or — dist.exchange_tiles, round 5 —
                # every tile once from the rank that piled it up (a rank of N sends about 1/N of this and receives the rest)
                "allreduce_message_bytes": int(8 * (nf + ni)),
                "exchange_tiles_total_bytes": int(8 * sum(eng.tile_block_sizes(int(rows.shape[0])))),
                "exchange_bytes_per_rank_of_8": int(8 * sum(eng.tile_block_sizes((int(rows.shape[0]) + 7) // 8)))})
    print(json.dumps(res[-1]), flush=True)
if a.profile:
    import cProfile, pstats
    cProfile.runctx("coolpup.pileup(clr, bed, **kw)", globals(), locals(), "/tmp/bywindow.prof")
    st = pstats.Stats("/tmp/bywindow.prof")
    st.sort_stats("cumulative").print_stats(a.profile)
    st.sort_stats("tottime").print_stats(a.profile)
if a.out:
    json.dump({"what": "by-window pile-up, Bonev CTCF+ sites on the mm9-like synthetic 10 kb table (tools/probe_bywindow.py)", "runs": res},
              open(a.out, "w"), indent=1)
