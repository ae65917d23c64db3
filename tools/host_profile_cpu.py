"""Host-layer profile WITHOUT a GPU: pileup() with the engine replaced by a stub that accepts every call and returns zero tiles
(build container only; the numbers it prints are host time alone).  python tools/host_profile_cpu.py [--plain|--local|--trans|--bywindow|--loops]"""
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
from coolpuppy_amd import coolpup  # noqa: E402
import synth  # noqa: E402


class NullEngine:
    device_id = 0

    def __init__(self, nbins):
        self.nbins = nbins
        self.n_tiles = self.pad = 0

    def load_bins(self, *a, **k): pass
    def set_expected(self, *a, **k): pass
    def set_expected_table(self, *a, **k): pass
    def set_tuning(self, *a, **k): pass

    def reset(self, T, pad):
        self.n_tiles, self.pad = int(T), int(pad)

    def accumulate(self, r0, c0, tile_ptr, **k):
        self.last = (len(r0), np.diff(tile_ptr))

    accumulate_rescaled = accumulate

    def fetch(self):
        T, W = self.n_tiles, 2 * self.pad + 1
        n = np.zeros(T, np.int64)
        if hasattr(self, "last") and len(self.last[1]) == T:
            n[:] = self.last[1]
        return {"sum": np.ones((T, W, W)), "num": np.ones((T, W, W), np.int64), "n": n,
                "cov_start": np.zeros((T, W)), "cov_end": np.zeros((T, W))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--plain", action="store_true")
    ap.add_argument("--trans", action="store_true")
    ap.add_argument("--local", action="store_true")
    ap.add_argument("--bywindow", action="store_true")
    ap.add_argument("--loops", action="store_true", help="configs[0] shape: 3 331 loops, no controls")
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--top", type=int, default=35)
    ap.add_argument("--reps", type=int, default=3)
    a = ap.parse_args()
    warnings.simplefilter("ignore")
    import gzip
    import pandas as pd
    if a.local or a.bywindow or a.loops:
        hg = synth.make_cooler(synth.MM9, binsize=10_000, lam=2, seed=1000, name="sparse_mm9", parallel=True)
    else:
        hg = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=2, seed=1000, name="sparse_hg38", parallel=True)
    gold = os.path.join(ROOT, "tests", "golden", "ref_data")
    if a.local or a.bywindow:
        with gzip.open(os.path.join(gold, "Bonev_CTCF+.bed.gz"), "rt") as f:
            feats = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
        if a.local:
            kw = dict(features_format="bed", flank=100_000, local=True, expected_df=synth.cis_expected(hg))
        else:
            kw = dict(features_format="bed", flank=100_000, by_window=True, mindist=300_000, maxdist=1_000_000)
    elif a.loops:
        feats = synth.random_cis_pairs(hg, 3331, seed=42)
        kw = dict(features_format="bedpe", flank=100_000, nshifts=0)
    elif a.trans:
        feats = synth.random_trans_pairs(hg, a.pairs // 2, seed=43)
        kw = dict(features_format="bedpe", flank=250_000, trans=True)
    else:
        feats = synth.random_cis_pairs(hg, a.pairs, seed=42, strands=True)
        kw = dict(features_format="bedpe", flank=100_000, nshifts=10, seed=0)
        if not a.plain:
            kw.update(by_distance=True, by_strand=True)
    null = NullEngine(hg.bins().shape[0] if hasattr(hg.bins(), "shape") else 0)
    coolpup._engine_for = lambda clr, dev, rows=None: null
    coolpup._prefetch_engine = lambda *x, **k: None
    from coolpuppy_amd import dist
    dist.allreduce_engine = lambda eng: None
    coolpup.pileup(hg, feats, **kw)
    best = 1e9
    for _ in range(a.reps):
        t = time.time()
        df = coolpup.pileup(hg, feats, **kw)
        best = min(best, time.time() - t)
        print(f"pileup wall {time.time() - t:.4f}s rows {len(df)}", flush=True)
    cProfile.runctx("coolpup.pileup(hg, feats, **kw)", globals(), locals(), "/tmp/host.prof")
    st = pstats.Stats("/tmp/host.prof")
    st.sort_stats("cumulative").print_stats(a.top)
    st.sort_stats("tottime").print_stats(a.top)


if __name__ == "__main__":
    main()
