"""Dev tool (GPU box): time the pile-up kernel variants on the bench workload (BASELINE configs[2]) and check that
they agree.  Input order = the reference's stream order (what pileup() hands the engine), so the device-side block
sort is part of every timed call.  Not part of the product or the tests.

    python tools/k1_probe.py --variants 0,128,64,16 --reps 5 [--chroms 23] [--pairs 1000000]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default="0,128,64,16")
    ap.add_argument("--reps", type=int, default=5)
    ap.add_argument("--chroms", type=int, default=23)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--nshifts", type=int, default=10)
    ap.add_argument("--pad", type=int, default=10)
    ap.add_argument("--lam", type=float, default=4200.0)
    ap.add_argument("--chunk", type=int, default=0)
    ap.add_argument("--tiles", type=int, default=0, help="split every kind over this many random groups (many-tile shape)")
    ap.add_argument("--ooe", action="store_true", help="observed over expected: a synthetic by-diagonal expected per chromosome")
    ap.add_argument("--out", default="")
    a = ap.parse_args()
    a.no_cache = False
    a.scaling = "weak"
    wl = bench.load_workload(a, 0, 1)
    from coolpuppy_amd.engine import PileupEngine
    r0, c0, n_roi = wl["r0_stream"], wl["c0_stream"], int(wl["n_roi"])
    n = len(r0)
    tile_ptr = np.array([0, n_roi, n], np.int64)
    T = 2
    if a.tiles > 1:
        kind = (np.arange(n) >= n_roi).astype(np.int64)
        g = np.random.RandomState(1).randint(0, a.tiles, n) + a.tiles * kind
        o = np.argsort(g, kind="stable")
        r0, c0 = r0[o], c0[o]
        T = 2 * a.tiles
        tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(g, minlength=T))]).astype(np.int64)
    import torch
    eng = PileupEngine(0)
    eng.load_pixels(wl["bin1_offset"], wl["bin2_id"], wl["count"])
    eng.load_bins(wl["weight"], None)
    eng.build_index(wl["chrom_offset"])
    mode = 0
    if a.ooe:
        co = np.asarray(wl["chrom_offset"], np.int64)
        eng.set_expected_table(co[:-1], co[1:], vectors=[1.0 / (1.0 + np.arange(int(e - s), dtype=np.float64)) for s, e in zip(co[:-1], co[1:])])
        mode = 1
    d_r0 = torch.from_numpy(np.ascontiguousarray(r0)).cuda()
    d_c0 = torch.from_numpy(np.ascontiguousarray(c0)).cuda()
    torch.cuda.synchronize()
    res = {}
    ref = None
    for v in [int(x) for x in a.variants.split(",")]:
        eng.set_tuning(a.chunk, v)
        eng.set_profiling(3)          # HIP events only: no pixel statistics inside the kernels
        rows = []
        for rep in range(a.reps):
            eng.clear_stats()
            eng.reset(T, a.pad)
            eng.sync()
            t = time.perf_counter()
            eng.accumulate_device(d_r0.data_ptr(), d_c0.data_ptr(), n, tile_ptr, ignore_diags=2, mode=mode)
            eng.sync()
            wall = (time.perf_counter() - t) * 1e3
            st = eng.stats()
            rows.append({"wall_ms": round(wall, 3), "k1_ms": round(st["k1_ms"], 3), "prep_ms": round(st["prepare_ms"], 3),
                         "reduce_ms": round(st["reduce_ms"], 3), "staged": int(st["staged_regions"])})
        out = eng.fetch()
        if ref is None:
            ref = out
            agree = True
        else:
            agree = bool(np.array_equal(out["n"], ref["n"]) and np.array_equal(out["num"], ref["num"]) and
                         np.allclose(out["sum"], ref["sum"], rtol=1e-12, atol=0))
        if v & (4 << 24):            # phase clocks of the staged kernel (variant bit 26)
            tm = eng.debug_timing()
            if tm is not None:
                names = ["issue", "windows", "barrier1", "store", "barrier2", "rows"]
                live = tm[:, :, 6].max(axis=1) > 0
                per_wave = tm[live][:, :, :6].astype(float)
                tot = per_wave.sum(axis=2)                       # clocks per wave
                k = per_wave.shape[1] if (per_wave[:, 8:, :].sum() > 0) else 8
                per_wave = per_wave[:, :k]
                print(json.dumps({"phases_mean_clk_per_wave": {n: round(float(per_wave[:, :, i].mean())) for i, n in enumerate(names)},
                                  "waves": int(k), "workgroups": int(live.sum()),
                                  "wave_total_min_mean_max": [round(float(x)) for x in (per_wave.sum(axis=2).min(), per_wave.sum(axis=2).mean(), per_wave.sum(axis=2).max())],
                                  "wg_total_min_mean_max": [round(float(x)) for x in (per_wave.sum(axis=2).max(axis=1).min(), per_wave.sum(axis=2).max(axis=1).mean(), per_wave.sum(axis=2).max(axis=1).max())],
                                  "lookahead_inside_windows_mean": round(float(tm[live][:, :k, 7].mean())),
                                  "lookahead_by_wave_mean": [round(float(x)) for x in tm[live][:, :k, 7].mean(axis=0)],
                                  "windows_by_wave_mean": [round(float(x)) for x in per_wave[:, :, 1].mean(axis=0)],
                                  "blocks_per_wg_min_mean_max": [int(tm[live][:, 0, 6].min()), float(tm[live][:, 0, 6].mean()), int(tm[live][:, 0, 6].max())]}), flush=True)
        best = min(rows, key=lambda x: x["wall_ms"])
        res[str(v)] = {"best": best, "all": rows, "agrees_with_first": agree, "n": out["n"].tolist()[:4]}
        print(json.dumps({"variant": v, **best, "agree": agree}), flush=True)
    if a.out:
        json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
