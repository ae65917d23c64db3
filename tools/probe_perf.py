"""Dev tool: time the pile-up kernel on a synthetic genome slice (not part of the product or the tests)."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import synth  # noqa: E402
from coolpuppy_amd.engine import PileupEngine  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--chroms", type=int, default=3)
    ap.add_argument("--lam", type=float, default=1000.0)
    ap.add_argument("--pairs", type=int, default=200_000)
    ap.add_argument("--nshifts", type=int, default=10)
    ap.add_argument("--pad", type=int, default=10)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--sort", type=int, default=1)
    ap.add_argument("--tiles", type=int, default=1, help="split ROI and control snippets over this many groups each")
    ap.add_argument("--variant", type=int, default=0, help="1 = ignore the rank-bitmap index")
    ap.add_argument("--group", type=str, default="0", help="comma list of waves-per-group values to sweep")
    a = ap.parse_args()
    names = list(synth.HG38)[: a.chroms]
    t = time.time()
    clr = synth.make_cooler({c: synth.HG38[c] for c in names}, lam=a.lam, seed=1000)
    print(f"cooler: {clr.nbins} bins, {clr.nnz} nnz, {time.time()-t:.1f}s", flush=True)
    pairs = synth.random_cis_pairs(clr, a.pairs, seed=42)
    res = clr.binsize
    W = 2 * a.pad + 1
    off = {c: clr.offset(c) for c in clr.chromnames}
    choff = pairs["chrom1"].map(off).values
    r0 = (pairs["start1"].values // res - a.pad + choff).astype(np.int64)
    c0 = (pairs["start2"].values // res - a.pad + choff).astype(np.int64)
    rng = np.random.RandomState(0)
    rr, cc, kind = [r0], [c0], [np.zeros(len(r0), np.int8)]
    for _ in range(a.nshifts):
        sh = np.round(rng.randint(100_000, 1_000_000, len(r0)) * rng.choice([-1, 1], len(r0)) / res).astype(np.int64)
        rr.append(r0 + sh); cc.append(c0 + sh); kind.append(np.ones(len(r0), np.int8))
    r = np.concatenate(rr); c = np.concatenate(cc); k = np.concatenate(kind)
    lo = np.repeat(choff, 1)  # region bounds per pair
    clo = np.tile(choff, a.nshifts + 1)
    chi = np.tile(pairs["chrom1"].map({cn: clr.extent(cn)[1] for cn in clr.chromnames}).values, a.nshifts + 1)
    ok = (r >= clo) & (c >= clo) & (r + W <= chi) & (c + W <= chi)
    r, c, k = r[ok], c[ok], k[ok]
    order = np.lexsort((c, r, k)) if a.sort else np.argsort(k, kind="stable")
    r, c, k = r[order].astype(np.int32), c[order].astype(np.int32), k[order]
    tile_ptr = np.array([0, int((k == 0).sum()), len(k)], np.int64)
    n_tiles = 2
    if a.tiles > 1:
        # by-distance x by-strand shape: every snippet gets one of a.tiles groups per kind (position order kept)
        g = np.random.RandomState(1).randint(0, a.tiles, len(k)) + a.tiles * k.astype(np.int64)
        o = np.argsort(g, kind="stable")
        r, c, k = r[o], c[o], k[o]
        n_tiles = 2 * a.tiles
        tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(g, minlength=n_tiles))]).astype(np.int64)
    print(f"snippets: {len(r)}", flush=True)
    eng = PileupEngine(0)
    t = time.time(); eng.load_pixels(*clr.pixel_table()); print(f"H2D pixels {time.time()-t:.2f}s")
    t = time.time(); ok = eng.build_index(clr.chrom_offset); print(f"index built={ok} {time.time()-t:.3f}s")
    eng.load_bins(clr.bins()["weight"][:].values, None)
    eng.set_profiling(True)
    for grp in [int(x) for x in a.group.split(",")]:
      eng.set_tuning(a.chunk, a.variant | (grp << 8))
      eng.reset(n_tiles, a.pad)
      print("group_waves", grp, flush=True)
      for rep in range(a.reps):
          eng.clear_stats()
          t = time.time()
          eng.accumulate(r, c, tile_ptr, ignore_diags=2, mode=0)
          eng.sync()
          wall = time.time() - t
          st = eng.stats()
          n = len(r)
          alg_bytes = n * (8 * (W + 1) + 16 * W + 12) + 8 * st["pixels_in_windows"]
          print(json.dumps({"rep": rep, "wall_s": round(wall, 4), "k1_ms": round(st["k1_ms"], 3),
                            "reduce_ms": round(st["reduce_ms"], 3),
                            "snips_per_s_k1": round(n / (st["k1_ms"] * 1e-3)),
                            "nnz_win_mean": round(st["pixels_in_windows"] / n, 1),
                            "probes_per_snip": round(st["probe_loads"] / n, 1),
                            "staged_regions": st["staged_regions"], "prepare_ms": round(st["prepare_ms"], 3),
                            "alg_GBps": round(alg_bytes / (st["k1_ms"] * 1e-3) / 1e9, 1)}), flush=True)
    out = eng.fetch()
    print("n", out["n"][:4], "center", (out["sum"][:, a.pad, a.pad] / np.maximum(out["num"][:, a.pad, a.pad], 1))[:4])


if __name__ == "__main__":
    main()
