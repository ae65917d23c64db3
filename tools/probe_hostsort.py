"""Dev probe: how fast are the pieces of a 1e6-row pair sort on this host?"""
import os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth
hg = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=2, seed=1000, name="s", parallel=True)
f = synth.random_cis_pairs(hg, 1_000_000, seed=42, strands=True)
n = len(f)
def T(label, fn, reps=3):
    best = 1e9
    for _ in range(reps):
        t = time.perf_counter(); r = fn(); best = min(best, time.perf_counter() - t)
    print(f"{label:34s} {1e3*best:8.1f} ms", flush=True); return r
s1, s2 = f["start1"].to_numpy(), f["start2"].to_numpy()
cat = T("concat chroms", lambda: np.concatenate([f["chrom1"].to_numpy(), f["chrom2"].to_numpy()]))
codes, uniq = T("factorize 2e6", lambda: pd.factorize(cat))
rank = np.empty(len(uniq), np.int64); rank[np.argsort(np.asarray(uniq, dtype=object), kind="stable")] = np.arange(len(uniq))
hi = T("compose key", lambda: ((rank[codes[:n]] * len(uniq) + rank[codes[n:]]) << 32) | s1.astype(np.int64))
T("argsort quick int64", lambda: np.argsort(hi))
T("argsort stable int64", lambda: np.argsort(hi, kind="stable"))
o = T("lexsort (s2, hi)", lambda: np.lexsort((s2, hi)))
T("take all columns", lambda: f.take(o))
T("sort_values 4 keys", lambda: f.sort_values(["chrom1", "chrom2", "start1", "start2"]))
fc = f.copy(); fc["chrom1"] = fc["chrom1"].astype("category"); fc["chrom2"] = fc["chrom2"].astype("category")
T("sort_values, categorical chroms", lambda: fc.sort_values(["chrom1", "chrom2", "start1", "start2"]))
T("take, categorical chroms", lambda: fc.take(o))
print(np.__version__, pd.__version__, np.show_config is not None)
