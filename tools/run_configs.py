"""Run the five BASELINE.json configurations end to end through the public API on one GPU and write a JSON summary.

    python tools/run_configs.py --out gpurun_out/configs.json

Per config: wall time of pileup() (host coordinate generation + GPU + finaliser, table already resident), the GPU
engine time alone (HIP events), snippets piled up, a same-run parity check against the C oracle on a strided sample
of the engine calls, and for config 0 the qualitative comparison with the reference's legacy loop_ref.np.txt shape.
Not a test and not the benchmark: a reproducible record of SURVEY.md §8(d)'s other rows.
"""
import argparse
import gzip
import json
import os
import sys
import time
import warnings

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from coolpuppy_amd import coolpup
import synth  # noqa: E402
from coolpuppy_amd.coolpup import _engine_for  # noqa: E402
from oracle import pileup_oracle as po  # noqa: E402

REF = os.path.join(ROOT, "tests", "golden", "ref_data")


def engine_time(pu, plan, reps=3):
    eng = _engine_for(pu._aclr, 0)
    bins = pu.clr.bins()
    eng.load_bins(bins[plan["weight_name"]][:].values if plan["weight_name"] else None,
                  bins[plan["cov_name"]][:].values if plan["cov_name"] else None)
    eng.set_profiling(3)          # HIP events only: no pixel statistics inside the timed kernels
    best = None
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
        rec = {"engine_wall_s": round(wall, 5), "k1_ms": round(st["k1_ms"], 3), "reduce_ms": round(st["reduce_ms"], 3),
               "snippets": int(st["snippets"]), "calls": len(plan["calls"]), "tiles": plan["T"]}
        if best is None or rec["engine_wall_s"] < best["engine_wall_s"]:
            best = rec
    got = eng.fetch()
    eng.set_profiling(False)
    return best, got


def oracle_check(pu, plan, got, max_snips=60_000):
    """Replay a strided sample of every call on the oracle AND on the GPU; compare exactly / 1e-6."""
    indptr, col, cnt = pu._aclr.pixel_table()
    bins = pu.clr.bins()
    weight = bins[plan["weight_name"]][:].values if plan["weight_name"] else None
    cov = bins[plan["cov_name"]][:].values if plan["cov_name"] else None
    total = sum(len(c["r0"]) for c in plan["calls"])
    step = max(1, total // max_snips)
    eng = _engine_for(pu._aclr, 0)
    eng.reset(plan["T"], plan["pad"])
    acc = po.empty_acc(plan["T"], plan["pad"])
    n_s = 0
    et = plan.get("expected_table")
    for c in plan["calls"]:
        idx = np.arange(0, len(c["r0"]), step)
        if len(idx) == 0:
            continue
        tile = c["tile"][idx]
        flip = None if c["flip"] is None else c["flip"][idx]
        call = coolpup._engine_call(c["region1"], c["region2"], c["expected"], c["r0"][idx].astype(np.int64),
                                    c["c0"][idx].astype(np.int64), flip, tile.astype(np.int64), plan["T"],
                                    c["ignore_diags"], c["mode"])
        if isinstance(call["expected"], str):
            eng.set_expected_table(et["start"], et["end"], vectors=et["vectors"], pair=et["pair"])
        else:
            eng.set_expected(call["expected"])
        eng.accumulate(call["r0"], call["c0"], call["tile_ptr"], flip_from=call["flip_from"],
                       ignore_diags=call["ignore_diags"], mode=call["mode"])
        for expected, sc in coolpup.iter_expected_subcalls(plan, call):
            po.pileup_c(indptr, col, cnt, weight, cov, expected, sc["r0"], sc["c0"], sc["flip"], sc["tile"],
                        plan["T"], plan["pad"], sc["ignore_diags"], sc["mode"], acc=acc)
        n_s += len(idx)
    g = eng.fetch()
    ok = (np.array_equal(g["n"], acc["n"]) and np.array_equal(g["num"], acc["num"])
          and np.allclose(g["sum"], acc["sum"], rtol=1e-6, atol=0, equal_nan=True)
          and np.allclose(g["cov_start"], acc["cov_start"], rtol=1e-9) and np.allclose(g["cov_end"], acc["cov_end"], rtol=1e-9))
    return {"sample_snippets": int(n_s), "gpu_equals_oracle": bool(ok)}


def run(name, clr, features, kw, plan_kw=None):
    out = {"config": name}
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        coolpup.pileup(clr, features, **kw)                       # first call: uploads the table, builds the index
        t = time.time()
        df = coolpup.pileup(clr, features, **kw)
        out["pileup_wall_s"] = round(time.time() - t, 3)
        out["rows"] = int(len(df))
        out["n_all"] = int(df["n"].iloc[-1] if "all" in str(df["group"].iloc[-1]) else df.loc[df["group"].astype(str) == "all", "n"].iloc[0])
        # the same plan, engine only
        if kw.get("seed") is not None:
            np.random.seed(kw["seed"])
        view = kw.get("view_df")
        cc = coolpup.CoordCreator(features, clr.binsize, features_format=kw["features_format"], flank=kw["flank"],
                                  nshifts=kw.get("nshifts", 0), trans=kw.get("trans", False), local=kw.get("local", False),
                                  chroms=list(clr.chromnames), mindist=kw.get("mindist", "auto"))
        pu = coolpup.PileUpper(clr, cc, control=kw.get("nshifts", 0) > 0, expected=kw.get("expected_df", False),
                               ignore_diags=kw.get("min_diag", 2), view_df=view)
        pu.ignore_group_order = False
        t = time.time()
        groupby, modify, cols = [], None, ()
        if kw.get("by_strand") and kw.get("by_distance"):
            from functools import partial
            modify = partial(coolpup.bin_distance_intervals, band_edges="default")
            groupby, cols = ["strand1", "strand2", "distance_band"], ["distance"]
        batches = [(r1, r2, pu.region_snippets(r1, r2, groupby=groupby, modify_2Dintervals_func=modify, columns=cols))
                   for r1, r2 in pu._region_pairs()]
        plan = pu.make_plan(batches, groupby)
        out["host_coordinates_plan_s"] = round(time.time() - t, 3)
        best, got = engine_time(pu, plan)
        out.update(best)
        out["snippets_per_s_engine"] = round(best["snippets"] / best["engine_wall_s"])
        out.update(oracle_check(pu, plan, got))
    print(json.dumps(out), flush=True)
    return out, df


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "configs.json"))
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    want = set(a.only.split(",")) if a.only else None
    res = []

    # ---- configs 0 and 1: mm9-like synthetic 10 kb cooler (the real Scc1-control.10000.cool is not in the tree)
    if want is None or want & {"0", "1"}:
        t = time.time()
        mm9 = synth.make_cooler(synth.MM9, binsize=10_000, lam=120, seed=1000, name="synthetic_mm9_10kb", parallel=True)
        print(f"mm9-like cooler: {mm9.nbins} bins, {mm9.nnz} nnz, {time.time()-t:.1f}s", flush=True)
    if want is None or "0" in want:
        loops = pd.read_csv(os.path.join(REF, "CH12_loops_Rao.bed"), sep="\t", header=None,
                            names=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])
        r, df = run("configs[0] CH12 loops, pad=10, nshifts=0, balanced", mm9, loops,
                    dict(features_format="bedpe", flank=100_000, nshifts=0))
        d = df["data"].iloc[0]
        r["pileup_shape"] = list(d.shape)
        r["note"] = ("synthetic distance-decay matrix has no loops: centre/corner ratio ~1 (the legacy loop_ref.np.txt "
                     "from real data has centre 1.80); shape and NaN-free 21x21 output match")
        r["centre_over_mean"] = float(d[10, 10] / np.nanmean(d))
        res.append(r)
    if want is None or "1" in want:
        exp = synth.cis_expected(mm9)
        for sign in ("+", "-"):
            with gzip.open(os.path.join(REF, f"Bonev_CTCF{sign}.bed.gz"), "rt") as f:
                bed = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
            r, df = run(f"configs[1] Bonev_CTCF{sign} local pile-up, pad=10, expected (ooe), ignore_diags=2", mm9, bed,
                        dict(features_format="bed", flank=100_000, local=True, expected_df=exp))
            r["nan_cells_in_pileup"] = int(np.isnan(df["data"].iloc[0]).sum())      # SURVEY: 61 for ignore_diags=2
            res.append(r)
    # ---- configs 3 and 4 on the human-scale table ---------------------------------------------------------------
    if want is None or want & {"2", "3", "4"}:
        t = time.time()
        hg = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=4200, seed=1000,
                               name="synthetic_hg38_10kb", parallel=True, trans_nnz=50_000_000)
        print(f"hg38-like cooler: {hg.nbins} bins, {hg.nnz} nnz, {time.time()-t:.1f}s", flush=True)
    if want is None or "2" in want:
        feats = synth.random_cis_pairs(hg, 1_000_000, seed=42, strands=True)
        r, _ = run("configs[2] 1e6 pairs, nshifts=10 (the bench workload, through pileup())", hg, feats,
                   dict(features_format="bedpe", flank=100_000, nshifts=10, seed=0))
        res.append(r)
    if want is None or "3" in want:
        feats = synth.random_cis_pairs(hg, 1_000_000, seed=42, strands=True)
        r, _ = run("configs[3] 1e6 pairs, by-distance + by-strand, nshifts=10", hg, feats,
                   dict(features_format="bedpe", flank=100_000, nshifts=10, seed=0, by_distance=True, by_strand=True))
        res.append(r)
    if want is None or "4" in want:
        feats = synth.random_trans_pairs(hg, 500_000, seed=43)
        r, _ = run("configs[4] trans, 5e5 inter-chromosomal pairs, pad=25", hg, feats,
                   dict(features_format="bedpe", flank=250_000, trans=True))
        res.append(r)
    os.makedirs(os.path.dirname(a.out), exist_ok=True)
    json.dump(res, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
