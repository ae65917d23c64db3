"""Dev probe (GPU box): the inter-chromosomal kernel K1s and its presence filter as the genome grows.

  A. the configs[4] geometry (3.1e5 bins, 5e7 inter-chromosomal pixels, 5e5 windows of 51 x 51) with the exact bitmap and with the
     filter forced to 2, 4, 8, 64 columns per bit (COOLPUPPY_AMD_TBITS_SHIFT) and switched off (variant 1: bisection per row);
  B. a table of 1.2e6 bins at the SAME pixel density per window (7.6e8 pixels): the exact bitmap would take 180 GB, the engine
     picks the finest filter that fits a quarter of the free memory.  Parity of every window against the C oracle.

The tables are made on the GPU with torch (sorting 7.6e8 keys on one host core would take the whole call) and handed to the
engine through the host, as a caller would.  python tools/probe_trans_bins.py [out.json]
"""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import synth  # noqa: E402


def make_table(nbins_target, trans_nnz, seed):
    import torch
    sizes = np.array([synth.HG38[c] for c in synth.HG38], np.float64)
    nb = np.maximum((sizes / sizes.sum() * nbins_target).astype(np.int64), 100)
    co = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
    nbins = int(co[-1])
    g = torch.Generator(device="cuda").manual_seed(seed)
    keys = []
    chunk = 1 << 27
    off = torch.from_numpy(co).cuda()
    for s in range(0, trans_nnz, chunk):
        m = min(chunk, trans_nnz - s)
        a = torch.randint(0, nbins, (m,), generator=g, device="cuda")
        b = torch.randint(0, nbins, (m,), generator=g, device="cuda")
        lo, hi = torch.minimum(a, b), torch.maximum(a, b)
        keep = torch.searchsorted(off, lo, right=True) != torch.searchsorted(off, hi, right=True)
        keys.append((lo * nbins + hi)[keep])
    key = torch.unique(torch.cat(keys))
    del keys
    row = key // nbins
    col = (key % nbins).to(torch.int32).cpu().numpy()
    indptr = torch.searchsorted(row, torch.arange(nbins + 1, device="cuda")).cpu().numpy().astype(np.int64)
    del key, row
    torch.cuda.empty_cache()
    rng = np.random.default_rng(seed)
    w = rng.uniform(0.5, 1.5, nbins)
    w[rng.random(nbins) < 0.02] = np.nan
    return co, indptr, col, np.ones(col.shape[0], np.int32), w


def windows(co, n, pad, seed):
    rng = np.random.default_rng(seed)
    W = 2 * pad + 1
    nb = np.diff(co)
    p = nb / nb.sum()
    c1 = rng.choice(len(nb), n, p=p)
    c2 = rng.choice(len(nb), n, p=p)
    same = c1 == c2
    c2[same] = (c2[same] + 1) % len(nb)
    lo, hi = np.minimum(c1, c2), np.maximum(c1, c2)
    r0 = co[lo] + (rng.random(n) * (nb[lo] - W)).astype(np.int64)
    c0 = co[hi] + (rng.random(n) * (nb[hi] - W)).astype(np.int64)
    return r0.astype(np.int32), c0.astype(np.int32)


def run(co, indptr, col, cnt, w, r0, c0, pad, variant, shift, reps=4):
    from coolpuppy_amd.engine import PileupEngine
    if shift is None:
        os.environ.pop("COOLPUPPY_AMD_TBITS_SHIFT", None)
    else:
        os.environ["COOLPUPPY_AMD_TBITS_SHIFT"] = str(shift)
    eng = PileupEngine(0)
    eng.load_pixels(indptr, col, cnt)
    eng.build_index(co)
    eng.load_bins(w, None)
    eng.set_tuning(0, variant)
    eng.set_profiling(3)
    tp = np.array([0, len(r0)], np.int64)
    ms, wall = [], []
    for _ in range(reps):
        eng.clear_stats()
        eng.reset(1, pad)
        t = time.time()
        eng.accumulate(r0, c0, tp, ignore_diags=-1, mode=0)
        got = eng.fetch()
        wall.append(round((time.time() - t) * 1e3, 2))
        ms.append(round(eng.stats()["k1_ms"], 3))
    kern = eng.last_kernel()
    eng.close()
    os.environ.pop("COOLPUPPY_AMD_TBITS_SHIFT", None)
    return got, {"kernel": kern, "k1_ms": ms, "call_wall_ms": wall}


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else None
    from oracle import pileup_oracle as po          # (the checker of this probe, as in tests/: never the thing measured)
    po.build()
    res = []
    pad, n = 25, 500_000
    for label, nbins, nnz, cases in (
        ("A: configs[4] geometry", 308_840, 50_000_000, [("exact bitmap", 0, 0), ("2 columns per bit", 0, 1), ("4 columns per bit", 0, 2), ("8 columns per bit", 0, 3),
                                                          ("64 columns per bit", 0, 6), ("no filter (bisection per row)", 1, None)]),
        ("B: 1.2e6 bins, same pixel density", 1_200_000, int(50_000_000 * (1_200_000 / 308_840) ** 2), [("engine's choice", 0, None)]),
    ):
        t = time.time()
        co, indptr, col, cnt, w = make_table(nbins, nnz, 7)
        r0, c0 = windows(co, n, pad, 11)
        t_make = time.time() - t
        tile = np.zeros(n, np.int32)
        t = time.time()
        want = po.pileup_c_mt(indptr, col, cnt, w, None, None, r0, c0, None, tile, 1, pad, -1, 0, max(1, min(os.cpu_count() or 1, 64)))
        t_oracle = time.time() - t
        for name, variant, shift in cases:
            got, r = run(co, indptr, col, cnt, w, r0, c0, pad, variant, shift)
            ok = bool(np.array_equal(got["n"], want["n"]) and np.array_equal(got["num"], want["num"])
                      and np.allclose(got["sum"], want["sum"], rtol=1e-9, atol=0))
            r.update({"table": label, "filter": name, "nbins": int(co[-1]), "pixels": int(indptr[-1]), "windows": n, "pad": pad,
                      "pixels_per_window": round(float(want["sum"].size and (indptr[-1] / (co[-1] ** 2 / 2)) * (2 * pad + 1) ** 2), 3),
                      "gpu_equals_oracle": ok, "table_made_s": round(t_make, 1), "oracle_s": round(t_oracle, 1)})
            print(json.dumps(r), flush=True)
            res.append(r)
        del co, indptr, col, cnt, w
    if out:
        with open(out, "w") as f:
            json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
