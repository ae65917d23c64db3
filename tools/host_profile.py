"""cProfile of one end-to-end pileup() on a config-3-shaped workload (1e6 cis pairs, nshifts=10, by-distance + by-strand)
over a SPARSE synthetic table: the host side does not depend on nnz, so this isolates coordinates / plan / finalise cost.
Run on the GPU box:  python tools/host_profile.py [--plain]"""
import argparse
import cProfile
import os
import pstats
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coolpuppy_amd import coolpup
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plain", action="store_true", help="no by-distance / by-strand grouping")
    ap.add_argument("--trans", action="store_true", help="config-4 shape: inter-chromosomal pairs, pad=25, no controls")
    ap.add_argument("--local", action="store_true", help="config-1 shape: Bonev CTCF+ local pile-up with expected")
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--top", type=int, default=40)
    a = ap.parse_args()
    warnings.simplefilter("ignore")
    hg = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=20, seed=1000, name="sparse_hg38",
                           parallel=True)
    if a.local:
        import gzip
        import pandas as pd
        hg = synth.make_cooler(synth.MM9, binsize=10_000, lam=20, seed=1000, name="sparse_mm9", parallel=True)
        with gzip.open(os.path.join(ROOT, "tests", "golden", "ref_data", "Bonev_CTCF+.bed.gz"), "rt") as f:
            feats = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
        kw = dict(features_format="bed", flank=100_000, local=True, expected_df=synth.cis_expected(hg))
    elif a.trans:
        feats = synth.random_trans_pairs(hg, a.pairs // 2, seed=43)
        kw = dict(features_format="bedpe", flank=250_000, trans=True)
    else:
        feats = synth.random_cis_pairs(hg, a.pairs, seed=42, strands=True)
        kw = dict(features_format="bedpe", flank=100_000, nshifts=10, seed=0)
        if not a.plain:
            kw.update(by_distance=True, by_strand=True)
    coolpup.pileup(hg, feats, **kw)
    for _ in range(2):
        t = time.time()
        df = coolpup.pileup(hg, feats, **kw)
        print(f"pileup wall {time.time() - t:.3f}s rows {len(df)}", flush=True)
    cProfile.runctx("coolpup.pileup(hg, feats, **kw)", globals(), locals(), "/tmp/host.prof")
    st = pstats.Stats("/tmp/host.prof")
    st.sort_stats("cumulative").print_stats(a.top)
    st.sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
