import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from coolpuppy_amd import coolpup
import synth
hg = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=4200.0, seed=1000, name="synthetic_hg38_10kb", parallel=True, trans_nnz=50_000_000)
feats = synth.random_trans_pairs(hg, 500_000, seed=43)
cc = coolpup.CoordCreator(feats, hg.binsize, features_format="bedpe", flank=250_000, trans=True, chroms=list(hg.chromnames))
pu = coolpup.PileUpper(hg, cc, ignore_diags=2)
pu.ignore_group_order = False
batches = [(r1, r2, pu.region_snippets(r1, r2)) for r1, r2 in pu._region_pairs()]
plan = pu.make_plan(batches, [])
eng = coolpup._engine_for(pu._aclr, 0)
eng.load_bins(pu._aclr.bins()["weight"][:].values, None)
eng.sync()
for rep in range(3):
    t = time.perf_counter()
    eng.reset(plan["T"], plan["pad"])
    for c in plan["calls"]:
        eng.accumulate(c["r0"], c["c0"], c["tile_ptr"], flip_from=c["flip_from"], ignore_diags=c["ignore_diags"], mode=c["mode"])
    eng.sync()
    print("call", rep, round((time.perf_counter() - t) * 1e3, 2), "ms", flush=True)
