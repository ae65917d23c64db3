"""Dev probe (GPU box): the staged kernels on a pixel table beyond 2^30 pixels (COPIES x the human-scale synthetic table side by
side), band staging against index staging, with phase clocks.  python tools/probe_big_table.py [copies] [pairs_per_copy]"""
import os, sys, json
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import synth
import test_scale_gpu as t
from coolpuppy_amd.engine import PileupEngine

copies = int(sys.argv[1]) if len(sys.argv) > 1 else 4
per = int(sys.argv[2]) if len(sys.argv) > 2 else 250_000
one = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=4200, seed=1000, name="hg38_10kb", parallel=True)
indptr, col, cnt = one.pixel_table()
w = one.bins()["weight"][:].values
nb, nnz = one.nbins, int(indptr[-1])
big_indptr = np.concatenate([indptr[:-1] + k * nnz for k in range(copies)] + [[copies * nnz]]).astype(np.int64)
big_col = np.concatenate([col.astype(np.int32) + np.int32(k * nb) for k in range(copies)])
big_cnt = np.tile(cnt.astype(np.int32), copies)
co = np.concatenate([one.chrom_offset[:-1] + k * nb for k in range(copies)] + [[copies * nb]]).astype(np.int64)
eng = PileupEngine(0)
eng.load_pixels(big_indptr, big_col, big_cnt)
eng.build_index(co)
eng.load_bins(np.tile(w, copies), None)
rng = np.random.default_rng(1)
for pad in (10, 25):
    r0, c0, tp = t._windows(co, nb, per, pad, rng, copies=copies)
    for name, variant in (("band", 0), ("index", 1 << 27), ("band+phases", 1 << 26)):
        if pad == 25 and name == "index":
            continue
        eng.set_tuning(0, variant); eng.set_profiling(3)
        rows = []
        for rep in range(3):
            eng.clear_stats(); eng.reset(2, pad)
            eng.accumulate(r0, c0, tp, ignore_diags=2, mode=0)
            st = eng.stats()
            rows.append((round(st["k1_ms"], 3), round(st["prepare_ms"], 3), int(st["staged_regions"])))
        print(f"copies {copies} pad {pad} n {len(r0)} {name}: {eng.last_kernel()} (k1_ms, prepass_ms, regions) {rows}", flush=True)
        if variant & (1 << 26):
            tm = eng.debug_timing()
            if tm is not None:
                live = tm[:, :, 6].max(axis=1) > 0
                pw = tm[live][:, :, :6].astype(float)
                print("  phases mean clk/wave:", dict(zip(["issue", "windows", "barrier1", "store", "barrier2", "rows"], [round(float(pw[:, :, i].mean())) for i in range(6)])),
                      "wg total min/mean/max", [round(float(x)) for x in (pw.sum(axis=2).max(axis=1).min(), pw.sum(axis=2).max(axis=1).mean(), pw.sum(axis=2).max(axis=1).max())],
                      "blocks/wg", int(tm[live][:, 0, 6].min()), float(tm[live][:, 0, 6].mean()), int(tm[live][:, 0, 6].max()), flush=True)
