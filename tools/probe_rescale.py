"""Time a rescaled local pile-up (TAD-style: features x (1 + 2*rescale_flank), zoomed to rescale_size^2) end to end and
in the engine.  Run on the GPU box:  python tools/probe_rescale.py [--n 5000] [--size 99]"""
import argparse
import os
import sys
import time
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coolpuppy_amd import coolpup
import synth  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=5000)
    ap.add_argument("--size", type=int, default=99)
    ap.add_argument("--lam", type=float, default=120.0)
    a = ap.parse_args()
    warnings.simplefilter("ignore")
    clr = synth.make_cooler(synth.MM9, binsize=10_000, lam=a.lam, seed=1000, name="mm9_like", parallel=True)
    rng = np.random.default_rng(3)
    chroms = rng.choice(clr.chromnames, a.n)
    length = rng.integers(200_000, 2_000_000, a.n)
    start = np.array([rng.integers(3_000_000, int(clr.chromsizes[c]) - 6_000_000) for c in chroms])
    tads = pd.DataFrame({"chrom": chroms, "start": start, "end": start + length})
    exp = synth.cis_expected(clr)
    kw = dict(features_format="bed", local=True, rescale=True, rescale_flank=1, rescale_size=a.size, expected_df=exp)
    coolpup.pileup(clr, tads, **kw)
    from coolpuppy_amd.coolpup import _ENGINES
    eng = next(iter(_ENGINES.values()))[1]
    eng.set_profiling(True)
    for _ in range(2):
        eng.clear_stats()
        t = time.time()
        df = coolpup.pileup(clr, tads, **kw)
        st = eng.stats()
        print(f"n={int(df['n'].iloc[0])} size={a.size} pileup wall {time.time() - t:.3f}s  K5 {st['k1_ms']:.2f} ms "
              f"({st['k1_ms'] * 1e3 / max(int(df['n'].iloc[0]), 1):.1f} us / window)", flush=True)


if __name__ == "__main__":
    main()
