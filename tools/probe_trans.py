"""Dev probe (GPU box): BASELINE configs[4] (trans, pad 25) engine time against the windows-per-chunk setting of the sparse kernel."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coolpuppy_amd import coolpup
import synth

CACHE = os.environ.get("TRANS_CACHE")                # profiling runs: build the table once (multiprocessing) OUTSIDE rocprofv3
if CACHE and os.path.exists(CACHE):
    from coolpuppy_amd.cooler_lite import ArrayCooler
    z = np.load(CACHE)
    hg = ArrayCooler(dict(synth.HG38), 10_000, z["bin1_offset"], z["bin2_id"], z["count"], bins={"weight": z["weight"]},
                     filename="synthetic_hg38_10kb.cool")
else:
    hg = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=float(os.environ.get("LAM", "4200")), seed=1000,
                           name="synthetic_hg38_10kb", parallel=True, trans_nnz=50_000_000)
    if CACHE:
        i, c, v = hg.pixel_table()
        np.savez(CACHE, bin1_offset=i, bin2_id=c, count=v, weight=hg.bins()["weight"][:].values)
        if os.environ.get("TRANS_CACHE_ONLY"):
            sys.exit(0)
feats = synth.random_trans_pairs(hg, 500_000, seed=43)
cc = coolpup.CoordCreator(feats, hg.binsize, features_format="bedpe", flank=250_000, trans=True, chroms=list(hg.chromnames))
pu = coolpup.PileUpper(hg, cc, ignore_diags=2)
pu.ignore_group_order = False
batches = [(r1, r2, pu.region_snippets(r1, r2)) for r1, r2 in pu._region_pairs()]
plan = pu.make_plan(batches, [])
eng = coolpup._engine_for(pu._aclr, 0)
eng.load_bins(pu._aclr.bins()["weight"][:].values, None)
if os.environ.get("SORT_ROWS"):                     # experiment: windows of a tile in matrix-row order (locality of the row lookups)
    for c in plan["calls"]:
        tp = c["tile_ptr"]; r0 = np.asarray(c["r0"]).copy(); c0 = np.asarray(c["c0"]).copy()
        assert c["flip_from"] is None
        for t in range(len(tp) - 1):
            a, b = int(tp[t]), int(tp[t + 1])
            key = r0[a:b].astype(np.int64) * (1 << 20) + c0[a:b] if os.environ["SORT_ROWS"] == "rc" else r0[a:b]
            o = np.argsort(key, kind="stable")
            r0[a:b] = r0[a:b][o]; c0[a:b] = c0[a:b][o]
        c["r0"], c["c0"] = r0, c0
ref = None
for C in [int(x) for x in (sys.argv[1] if len(sys.argv) > 1 else "0,100,70,50,35").split(",")]:
    eng.set_tuning(C, int(os.environ.get('VARIANT', '0')))
    eng.set_profiling(3)
    best = None
    for _ in range(4):
        eng.clear_stats(); eng.reset(plan["T"], plan["pad"])
        for c in plan["calls"]:
            eng.accumulate(c["r0"], c["c0"], c["tile_ptr"], flip_from=c["flip_from"], ignore_diags=c["ignore_diags"], mode=c["mode"])
        eng.sync(); st = eng.stats()
        rec = (round(st["k1_ms"], 3), round(st["reduce_ms"], 3))
        best = rec if best is None or sum(rec) < sum(best) else best
    out = eng.fetch()
    ref = out if ref is None else ref
    if os.environ.get("PHASES"):
        eng.set_profiling(1); eng.clear_stats(); eng.reset(plan["T"], plan["pad"])
        for c in plan["calls"]:
            eng.accumulate(c["r0"], c["c0"], c["tile_ptr"], flip_from=c["flip_from"], ignore_diags=c["ignore_diags"], mode=c["mode"])
        eng.sync(); st = eng.stats()
        # (a library built with COOLPUPPY_AMD_EXTRA_CXXFLAGS=-DPUP_K1S_CLOCKS=1: the sparse kernel's phase clocks, packed — see the kernel)
        c0_, c1_ = int(st["pixels_in_windows"]), int(st["probe_loads"])
        nw = -(-sum(len(c["r0"]) for c in plan["calls"]) // max(C, 1)) if C else None
        print("  clocks summed over waves: total", (c0_ & 0xffffffff) << 4, "phase A", (c0_ >> 32) << 4, "phase B", (c1_ & 0xffffffff) << 4,
              "batch set-up", (c1_ >> 32) << 4, "chunks (if C given)", nw, flush=True)
    print("chunk", C, "k1_ms, reduce_ms =", best, "same", bool(np.array_equal(out["num"], ref["num"]) and np.allclose(out["sum"], ref["sum"], rtol=1e-12)), flush=True)
