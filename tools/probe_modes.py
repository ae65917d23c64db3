"""Dev tool (GPU box): wall time of pileup() over the mode combinations users actually type, on one synthetic table — to find host paths
that are out of line with the others (time per snippet).  python tools/probe_modes.py [--profile NAME]"""
import argparse, gzip, os, sys, time, warnings
import numpy as np, pandas as pd
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from coolpuppy_amd import coolpup
import synth
ap = argparse.ArgumentParser()
ap.add_argument("--profile", default="")
ap.add_argument("--only", default="")
a = ap.parse_args()
warnings.simplefilter("ignore")
clr = synth.make_cooler(synth.MM9, binsize=10_000, lam=60, seed=1000, name="mm9_like60", parallel=True)
with gzip.open(os.path.join(ROOT, "tests", "golden", "ref_data", "Bonev_CTCF+.bed.gz"), "rt") as f:
    bed = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
bed["strand"] = np.where(np.arange(len(bed)) % 2 == 0, "+", "-")
pairs = synth.random_cis_pairs(clr, 200_000, seed=42, strands=True)
exp = synth.cis_expected(clr)
cases = {
    "bedpe plain": (pairs, dict(features_format="bedpe", flank=100_000)),
    "bedpe ooe": (pairs, dict(features_format="bedpe", flank=100_000, expected_df=exp)),
    "bedpe expected not ooe": (pairs, dict(features_format="bedpe", flank=100_000, expected_df=exp, ooe=False)),
    "bedpe nshifts 10": (pairs, dict(features_format="bedpe", flank=100_000, nshifts=10, seed=0)),
    "bedpe ooe by_distance": (pairs, dict(features_format="bedpe", flank=100_000, expected_df=exp, by_distance=True)),
    "bedpe by_strand flip": (pairs, dict(features_format="bedpe", flank=100_000, by_strand=True, flip_negative_strand=True)),
    "bedpe coverage_norm": (pairs, dict(features_format="bedpe", flank=100_000, coverage_norm=True, clr_weight_name=None)),
    "bedpe stripes": (pairs.iloc[:20_000], dict(features_format="bedpe", flank=100_000, store_stripes=True)),
    "bedpe rescale": (pairs.iloc[:20_000], dict(features_format="bedpe", flank=100_000, rescale=True, rescale_flank=1, rescale_size=51)),
    "bed combinations": (bed, dict(features_format="bed", flank=100_000, mindist=300_000, maxdist=1_000_000)),
    "bed combinations ooe": (bed, dict(features_format="bed", flank=100_000, mindist=300_000, maxdist=1_000_000, expected_df=exp)),
    "bed combinations nshifts 3": (bed, dict(features_format="bed", flank=100_000, mindist=300_000, maxdist=1_000_000, nshifts=3, seed=0)),
    "bed combinations by_strand": (bed, dict(features_format="bed", flank=100_000, mindist=300_000, maxdist=1_000_000, by_strand=True)),
    "bed local ooe": (bed, dict(features_format="bed", flank=100_000, local=True, expected_df=exp)),
    "bed local rescale": (bed.iloc[::4], dict(features_format="bed", flank=100_000, local=True, rescale=True, rescale_flank=3, rescale_size=51)),
    "bed by_window ooe": (bed, dict(features_format="bed", flank=100_000, by_window=True, mindist=300_000, maxdist=1_000_000, expected_df=exp)),
}
for name, (feats, kw) in cases.items():
    if a.only and a.only not in name:
        continue
    try:
        coolpup.pileup(clr, feats, **kw)
        t = time.time(); df = coolpup.pileup(clr, feats, **kw); wall = time.time() - t
        n = int(df["n"].iloc[-1]) if "n" in df else -1
        nall = int(df.loc[df["group"].astype(str) == "all", "n"].iloc[0]) if "group" in df.columns and (df["group"].astype(str) == "all").any() else n
        print(f"{name:32s} wall {wall:7.3f} s  rows {len(df):6d}  n(all) {nall:9d}  us/snippet {wall / max(nall, 1) * 1e6:8.3f}", flush=True)
        if a.profile and a.profile in name:
            import cProfile, pstats
            cProfile.runctx("coolpup.pileup(clr, feats, **kw)", globals(), locals(), "/tmp/m.prof")
            pstats.Stats("/tmp/m.prof").sort_stats("cumulative").print_stats(28)
    except Exception as e:      # noqa: BLE001
        print(f"{name:32s} ERROR {type(e).__name__}: {e}", flush=True)
