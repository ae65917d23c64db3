"""Dev tool: time the other BASELINE configs end to end through the public API (host + GPU) and the engine alone."""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coolpuppy_amd import coolpup, synth  # noqa: E402
from coolpuppy_amd.coolpup import _engine_for  # noqa: E402


def timed_plan(pu, plan, reps=3):
    eng = _engine_for(pu._aclr, 0)
    bins = pu.clr.bins()
    eng.load_bins(bins[plan["weight_name"]][:].values if plan["weight_name"] else None,
                  bins[plan["cov_name"]][:].values if plan["cov_name"] else None)
    eng.set_profiling(True)
    out = []
    for _ in range(reps):
        eng.clear_stats()
        t = time.time()
        eng.reset(plan["T"], plan["pad"])
        et = plan.get("expected_table")
        for c in plan["calls"]:
            if isinstance(c["expected"], str):
                eng.set_expected_table(et["start"], et["end"], vectors=et["vectors"], pair=et["pair"])
            else:
                eng.set_expected(c["expected"])
            eng.accumulate(c["r0"], c["c0"], c["tile_ptr"], flip_from=c["flip_from"], ignore_diags=c["ignore_diags"],
                           mode=c["mode"])
        eng.sync()
        wall = time.time() - t
        st = eng.stats()
        out.append((wall, st["k1_ms"], st["reduce_ms"], st["snippets"]))
    eng.set_profiling(False)
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", default="grouped")
    ap.add_argument("--chroms", type=int, default=23)
    ap.add_argument("--lam", type=float, default=1000)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    a = ap.parse_args()
    names = list(synth.HG38)[: a.chroms]
    t = time.time()
    clr = synth.make_cooler({c: synth.HG38[c] for c in names}, lam=a.lam, seed=1000, parallel=True,
                            trans_nnz=50_000_000 if a.config == "trans" else 0)
    print(f"cooler {clr.nbins} bins {clr.nnz} nnz in {time.time()-t:.1f}s", flush=True)
    if a.config == "grouped":
        feats = synth.random_cis_pairs(clr, a.pairs, seed=42, strands=True)
        kw = dict(features_format="bedpe", flank=100_000, nshifts=10, seed=0, by_distance=True, by_strand=True)
    elif a.config == "trans":
        feats = synth.random_trans_pairs(clr, a.pairs // 2, seed=43)
        kw = dict(features_format="bedpe", flank=250_000, trans=True)
    elif a.config == "local":
        feats = None
    t = time.time()
    df = coolpup.pileup(clr, feats, **kw)
    print(f"pileup() end to end: {time.time()-t:.2f}s rows={len(df)} n_all={int(df['n'].iloc[-1])}", flush=True)
    t = time.time()
    df = coolpup.pileup(clr, feats, **kw)
    print(f"pileup() second call (table resident): {time.time()-t:.2f}s", flush=True)
    # engine-only timing of the same plan
    np.random.seed(kw.get("seed"))
    cc = coolpup.CoordCreator(feats, clr.binsize, features_format="bedpe", flank=kw["flank"], nshifts=kw.get("nshifts", 0),
                              trans=kw.get("trans", False), chroms=list(clr.chromnames))
    pu = coolpup.PileUpper(clr, cc, control=kw.get("nshifts", 0) > 0)
    pu.ignore_group_order = False
    t = time.time()
    if a.config == "grouped":
        from functools import partial
        modify = partial(coolpup.bin_distance_intervals, band_edges="default")
        groupby = ["strand1", "strand2", "distance_band"]
        batches = [(r1, r2, pu.region_snippets(r1, r2, groupby=groupby, modify_2Dintervals_func=modify, columns=["distance"]))
                   for r1, r2 in pu._region_pairs()]
    else:
        groupby = []
        batches = [(r1, r2, pu.region_snippets(r1, r2)) for r1, r2 in pu._region_pairs()]
    plan = pu.make_plan(batches, groupby)
    print(f"host coordinates+plan: {time.time()-t:.2f}s calls={len(plan['calls'])} T={plan['T']}", flush=True)
    for wall, k1, red, n in timed_plan(pu, plan):
        print(json.dumps({"wall_s": round(wall, 4), "k1_ms": round(k1, 3), "reduce_ms": round(red, 3), "snippets": n,
                          "snips_per_s_wall": round(n / wall)}), flush=True)


if __name__ == "__main__":
    main()
