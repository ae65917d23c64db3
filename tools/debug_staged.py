"""Dev tool: localise differences between the staged kernels and the plain one (GPU box)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coolpuppy_amd import synth
from coolpuppy_amd.engine import PileupEngine, MODE_COV, MODE_OOE

pad = int(sys.argv[1]) if len(sys.argv) > 1 else 10
clr = synth.make_cooler({"chrA": 12_000_000, "chrB": 7_000_000, "chrC": 3_000_000}, lam=60, seed=21)
W, T, n = 2 * pad + 1, 3, 6000
rng = np.random.default_rng(1000 + pad)
r0l, c0l = [], []
for ch in clr.chromnames:
    lo, hi = clr.extent(ch)
    m = n // 3
    r = rng.integers(lo, hi - W + 1, m)
    c = np.clip(r + rng.integers(-6, 120, m), lo, hi - W)
    r[:8] = lo; c[:8] = lo + np.arange(8)
    r[8:16] = hi - W - np.arange(8); c[8:16] = hi - W
    r0l.append(r); c0l.append(c)
r0 = np.concatenate(r0l).astype(np.int32); c0 = np.concatenate(c0l).astype(np.int32)
w = clr.bins()["weight"][:].values
e = synth.cis_expected(clr)
expv = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
print("len expv", len(expv), clr.chrom_offset)
expv[3] = 0.0; expv[7] = np.nan
eng = PileupEngine(0)
eng.load_pixels(*clr.pixel_table())
eng.build_index(clr.chrom_offset)
eng.load_bins(w, None)
eng.set_expected(expv)

def run(rr, cc, variant):
    eng.set_tuning(0, variant); eng.reset(1, pad)
    eng.accumulate(rr, cc, np.array([0, len(rr)], np.int64), ignore_diags=2, mode=MODE_OOE)
    return eng.fetch()

def cmp(label, rr, cc):
    a = run(rr, cc, 16); b = run(rr, cc, 8)
    dn = b["num"][0] - a["num"][0]
    print(label, "n", len(rr), "num diff cells", int((dn != 0).sum()), "max", int(np.abs(dn).max()))
    return dn

cmp("all", r0, c0)
m = n // 3
for k, ch in enumerate(clr.chromnames):
    sl = slice(k * m, (k + 1) * m)
    cmp(ch, r0[sl], c0[sl])
    cmp(ch + " start", r0[sl][:8], c0[sl][:8])
    cmp(ch + " end", r0[sl][8:16], c0[sl][8:16])
    cmp(ch + " rest", r0[sl][16:], c0[sl][16:])
bad = 0
for i in list(range(0, 16)) + list(range(m, m + 16)) + list(range(16, 400)):
    a = run(r0[i:i+1], c0[i:i+1], 16); b = run(r0[i:i+1], c0[i:i+1], 8)
    dn = b["num"][0] - a["num"][0]
    if (dn != 0).any():
        bad += 1
        if bad <= 4:
            print("window", i, "r0", int(r0[i]), "c0", int(c0[i]), "d", int(c0[i] - r0[i]))
            print("plain num:\n", a["num"][0]); print("staged num:\n", b["num"][0])
print("single-window mismatches:", bad)
