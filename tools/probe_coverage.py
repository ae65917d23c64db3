"""Time K3 (pup_coverage) on the bench-sized synthetic table and check it against the numpy restatement on a sample.
Run on the GPU box:  python tools/probe_coverage.py [--lam 4200] [--trans 0]"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth  # noqa: E402
from coolpuppy_amd.engine import PileupEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--lam", type=float, default=4200.0)
    ap.add_argument("--chroms", type=int, default=23)
    ap.add_argument("--trans", type=int, default=0)
    ap.add_argument("--reps", type=int, default=5)
    a = ap.parse_args()
    names = list(synth.HG38)[: a.chroms]
    t = time.time()
    clr = synth.make_cooler({c: synth.HG38[c] for c in names}, binsize=10_000, lam=a.lam, seed=1000, name="cov",
                            parallel=True, trans_nnz=a.trans)
    print(f"table: {clr.nbins} bins, {clr.nnz} nnz ({time.time() - t:.1f}s)", flush=True)
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    for igd in (2, 0):
        best = 1e9
        for _ in range(a.reps):
            cis, tot = eng.coverage(clr.chrom_offset, ignore_diags=igd)
            best = min(best, eng.stats()["coverage_ms"])
        print(f"ignore_diags={igd}: K3 {best:.3f} ms  -> {clr.nnz * 8 / best / 1e6:.0f} GB/s of pixel bytes", flush=True)
    # parity on the whole table (numpy bincount restatement; exact integers)
    from oracle.pileup_oracle import coverage_numpy
    indptr, col, cnt = clr.pixel_table()
    wc, wt = coverage_numpy(indptr, col, cnt, clr.chrom_offset, 0)
    print("parity (ignore_diags=0):", bool(np.array_equal(cis, wc) and np.array_equal(tot, wt)))


if __name__ == "__main__":
    main()
