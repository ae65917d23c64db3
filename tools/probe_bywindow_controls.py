import gzip, os, sys, time, warnings
import numpy as np, pandas as pd
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
from coolpuppy_amd import coolpup
import synth
warnings.simplefilter("ignore")
clr = synth.make_cooler(synth.MM9, binsize=10_000, lam=120, seed=1000, name="mm9_like", parallel=True)
with gzip.open(os.path.join(ROOT, "tests", "golden", "ref_data", "Bonev_CTCF+.bed.gz"), "rt") as f:
    bed = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
kw = dict(features_format="bed", flank=100_000, by_window=True, mindist=300_000, maxdist=1_000_000, nshifts=3, seed=0)
for rep in range(2):
    t = time.time(); df = coolpup.pileup(clr, bed, **kw); print("by-window nshifts=3 wall", round(time.time() - t, 3), len(df), flush=True)
import cProfile, pstats
cProfile.runctx("coolpup.pileup(clr, bed, **kw)", globals(), locals(), "/tmp/p.prof")
pstats.Stats("/tmp/p.prof").sort_stats("cumulative").print_stats(22)
