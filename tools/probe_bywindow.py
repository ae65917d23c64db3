"""Time a by-window pile-up (one tile per feature) end to end with a host profile.  Run on the GPU box."""
import gzip, os, sys, time, warnings
import numpy as np, pandas as pd
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from coolpuppy_amd import coolpup
import synth
warnings.simplefilter("ignore")
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
clr = synth.make_cooler(synth.MM9, binsize=10_000, lam=120, seed=1000, name="mm9_like", parallel=True)
with gzip.open(os.path.join(ROOT, "tests", "golden", "ref_data", "Bonev_CTCF+.bed.gz"), "rt") as f:
    bed = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
bed = bed.iloc[::4].reset_index(drop=True)
kw = dict(features_format="bed", flank=100_000, by_window=True, mindist=300_000, maxdist=1_000_000)
coolpup.pileup(clr, bed, **kw)
import cProfile, pstats
t = time.time(); df = coolpup.pileup(clr, bed, **kw); print("by-window wall", round(time.time() - t, 3), "rows", len(df), "n_all", int(df["n"].iloc[-1]))
cProfile.runctx("coolpup.pileup(clr, bed, **kw)", globals(), locals(), "/tmp/bw.prof")
pstats.Stats("/tmp/bw.prof").sort_stats("cumulative").print_stats(18)
