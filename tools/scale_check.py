"""Size-independent parity at scale (run on the GPU box): millions of overlapping cis windows piled up twice — automatic
kernel choice (block-staged where it pays) vs the plain register-tile kernel — must give identical integers and sums
equal up to addition order, for several window widths, with flips, several tiles and observed/expected."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth  # noqa: E402
from coolpuppy_amd.engine import MODE_OOE, PileupEngine  # noqa: E402


def main():
    clr = synth.make_cooler({c: synth.HG38[c] for c in ("chr18", "chr19", "chr20", "chr21", "chr22")}, lam=1500, seed=1000,
                            parallel=True)
    e = synth.cis_expected(clr)
    nb = clr.nbins
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    eng.load_bins(clr.bins()["weight"][:].values, None)
    # expected table: one by-diagonal vector per chromosome
    starts = clr.chrom_offset[:-1]; ends = clr.chrom_offset[1:]
    vectors = [e[e.region1 == c]["balanced.avg"].values.astype(float) for c in clr.chromnames]
    rng = np.random.default_rng(7)
    out = []
    for pad in (3, 7, 10, 15):
        W = 2 * pad + 1
        for T, n, ooe in ((2, 3_000_000, False), (3, 2_400_000, True)):
            r0l, c0l = [], []
            for k in range(len(starts)):
                m = n // len(starts)
                r = rng.integers(starts[k], ends[k] - W - 300, m)
                r0l.append(r); c0l.append(r + rng.integers(2, 300, m))
            r0 = np.concatenate(r0l); c0 = np.concatenate(c0l)
            tile = rng.integers(0, T, len(r0)); flip = rng.random(len(r0)) < 0.2
            o = np.lexsort((c0, r0, flip, tile))
            r0, c0, tile, flip = r0[o].astype(np.int32), c0[o].astype(np.int32), tile[o], flip[o]
            tp = np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=T))]).astype(np.int64)
            ff = tp[1:] - np.bincount(tile[flip], minlength=T)
            mode = MODE_OOE if ooe else 0
            if ooe:
                eng.set_expected_table(starts, ends, vectors=vectors)
            res = {}
            for name, variant in (("plain", 16), ("auto", 0)):
                eng.set_tuning(0, variant)
                eng.reset(T, pad)
                t = time.time()
                eng.accumulate(r0, c0, tp, flip_from=ff, ignore_diags=2, mode=mode)
                eng.sync()
                res[name] = (eng.fetch(), time.time() - t, eng.stats()["staged_regions"])
            a, b = res["auto"][0], res["plain"][0]
            ok_int = bool(np.array_equal(a["num"], b["num"]) and np.array_equal(a["n"], b["n"]))
            with np.errstate(invalid="ignore", divide="ignore"):
                rel = float(np.nanmax(np.abs(a["sum"] - b["sum"]) / np.maximum(np.abs(b["sum"]), 1e-300)))
            rec = {"pad": pad, "tiles": T, "windows": int(len(r0)), "ooe": ooe, "staged_regions": int(res["auto"][2]),
                   "integers_equal": ok_int, "max_rel_diff_sum": rel,
                   "wall_plain_ms": round(res["plain"][1] * 1e3, 2), "wall_auto_ms": round(res["auto"][1] * 1e3, 2)}
            print(json.dumps(rec), flush=True)
            out.append(rec)
            assert ok_int and rel < 1e-10, rec
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scale_check.json"), "w"), indent=1)
    print("scale check passed:", len(out), "cases", nb, "bins")


if __name__ == "__main__":
    main()
