"""Helpers shared by the golden-vector tests: rebuild the inputs stored in tests/golden/*.npz, run this
build's pileup(), compare with what the reference produced (oracle/make_golden.py)."""
import io
import json
import os
import warnings

import numpy as np
import pandas as pd

from coolpuppy_amd.cooler_lite import ArrayCooler

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SCENARIOS = json.load(open(os.path.join(GOLD, "index.json")))

_coolers = {}


def cooler(name):
    if name not in _coolers:
        z = np.load(os.path.join(GOLD, "coolers.npz"))
        sizes = pd.Series(z[f"{name}__chromsizes"], index=[str(c) for c in z[f"{name}__chromnames"]])
        _coolers[name] = ArrayCooler(
            sizes, int(z[f"{name}__binsize"]), z[f"{name}__bin1_offset"], z[f"{name}__bin2_id"], z[f"{name}__count"],
            bins={c: z[f"{name}__{c}"] for c in ("weight", "cov_tot_raw", "cov_cis_raw")},
            filename=str(z[f"{name}__filename"]))
    return _coolers[name]


def scenario_cooler(meta):
    """The scenario's cooler: a stored one, with the bins-column patch of the scenario (if any) applied."""
    base = cooler(meta["cooler"])
    if not meta.get("patch"):
        return base
    import synth
    if meta["patch"].get("drop"):          # the run ADDS the dropped columns to the object: a fresh one every time
        return synth.patched_cooler(base, meta["patch"])
    key = meta["cooler"] + "|" + json.dumps(meta["patch"], sort_keys=True)
    if key not in _coolers:
        _coolers[key] = synth.patched_cooler(base, meta["patch"])
    return _coolers[key]


def load(name):
    z = np.load(os.path.join(GOLD, name + ".npz"))
    meta = json.loads(str(z["meta"]))
    rd = lambda txt: None if txt is None else pd.read_csv(io.StringIO(txt))   # noqa: E731
    kw = dict(meta["kw"])
    if isinstance(kw.get("by_distance"), list):
        kw["by_distance"] = np.array(kw["by_distance"])
    return z, meta, rd(meta["features"]), rd(meta["view"]), rd(meta["expected"]), kw


def key_repr(k):
    if isinstance(k, str):
        return k
    out = []
    for v in k:
        if isinstance(v, tuple):
            out.append([int(x) for x in v])
        elif isinstance(v, (np.integer, int)):
            out.append(int(v))
        else:
            out.append(str(v))
    return out


def run(name, pileup_func):
    z, meta, features, view, expected, kw = load(name)
    clr = scenario_cooler(meta)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = pileup_func(clr, features, view_df=view, expected_df=expected, **kw)
    for f in z.files:                           # columns the reference computed and stored in its cooler
        if f.startswith("stored__"):
            np.testing.assert_array_equal(np.asarray(clr.bins()[f[8:]][:].values, float), z[f])
    return z, df


def compare(z, df, rtol):
    want_groups = json.loads(str(z["group"]))
    if "group" in df.columns:
        got_groups = [key_repr(g) for g in df["group"]]
    else:   # by-window output spells the group out as chrom / start / end
        got_groups = [("all" if c == "all" else [str(c), int(s), int(e)])
                      for c, s, e in zip(df["chrom"], df["start"], df["end"])]
    assert got_groups == want_groups, f"group rows/order differ: {got_groups} vs {want_groups}"
    W = z["data"].shape[1]
    got = np.stack([np.asarray(x, float).reshape(W, W) if np.ndim(x) == 2 else np.full((W, W), np.nan)
                    for x in df["data"]])
    np.testing.assert_allclose(got, z["data"], rtol=rtol, atol=0, equal_nan=True)
    np.testing.assert_array_equal(df["n"].values.astype(float), z["n"])
    np.testing.assert_array_equal(np.stack([np.asarray(x) if np.ndim(x) == 2 else np.full((W, W), -1)
                                            for x in df["num"]]), z["num"])
    if "control_n" in z.files:
        np.testing.assert_array_equal(df["control_n"].values.astype(float), z["control_n"])
        got_cn = np.stack([np.asarray(x) if np.ndim(x) == 2 else np.full((W, W), -1) for x in df["control_num"]])
        np.testing.assert_array_equal(got_cn, z["control_num"])
    if "stripe_ptr" in z.files:
        ptr = z["stripe_ptr"]
        for i in range(len(df)):
            a, b = int(ptr[i]), int(ptr[i + 1])
            if b == a:
                continue
            np.testing.assert_array_equal(np.asarray(df["coordinates"].iloc[i]).astype(str), z["coords"][a:b])
            np.testing.assert_allclose(np.asarray(df["horizontal_stripe"].iloc[i], float), z["hstripe"][a:b],
                                       rtol=rtol, atol=0, equal_nan=True)
            np.testing.assert_allclose(np.asarray(df["vertical_stripe"].iloc[i], float), z["vstripe"][a:b],
                                       rtol=rtol, atol=0, equal_nan=True)
    for c in ("orientation", "separation"):
        if c in z.files:
            assert [str(x) for x in df[c]] == json.loads(str(z[c])), c
    assert list(df.columns) == json.loads(str(z["columns"])), "output columns differ from the reference's"
    scal = json.loads(str(z["scalars"]))
    for c, v in scal.items():
        if c in ("clr",):
            continue
        g = df[c].iloc[0]
        g = g.item() if isinstance(g, (np.integer, np.floating, np.bool_)) else g
        if isinstance(v, float) and np.isnan(v):
            assert g is None or (isinstance(g, float) and np.isnan(g)), (c, g, v)
        else:
            assert str(g) == str(v), f"column {c}: {g!r} != {v!r}"


def oracle_run_plan(pu, plan, calls=None, reduce=True):
    """Replay a plan (list of engine calls) on the CPU oracle — CPU stand-in for PileUpper.run_plan in tests."""
    from oracle import pileup_oracle as po
    indptr, col, cnt = pu._aclr.pixel_table()
    bins = pu.clr.bins()
    weight = bins[plan["weight_name"]][:].values if plan["weight_name"] else None
    cov = bins[plan["cov_name"]][:].values if plan["cov_name"] else None
    acc = po.empty_acc(plan["T"], plan["pad"])
    from coolpuppy_amd.coolpup import iter_expected_subcalls
    big = None
    for call in (plan["calls"] if calls is None else calls):
        for expected, c in iter_expected_subcalls(plan, call):
            if plan.get("rescale"):
                if big is None:
                    nb = indptr.shape[0] - 1
                    big = po.symmetric_csr(indptr, col, cnt, weight, 0, nb, 0, nb)
                po.pileup_rescaled(big, 0, 0, weight, cov, expected, c["r0"], c["c0"], c["h"], c["w"], c["flip"],
                                   c["tile"], plan["T"], 2 * plan["pad"] + 1, c["ignore_diags"], c["mode"], acc=acc)
                continue
            po.pileup_c(indptr, col, cnt, weight, cov, expected, c["r0"], c["c0"], c["flip"], c["tile"],
                        plan["T"], plan["pad"], c["ignore_diags"], c["mode"], acc=acc)
    from coolpuppy_amd import dist as pdist
    world = pdist.world()[1]
    if world > 1 and reduce and calls is None and plan["grouped"] and plan["T"] >= pdist.sparse_exchange_min_tiles():
        # many groups: the ranks swap the tiles they hold (dist.exchange_tiles; here its host-array form)
        from coolpuppy_amd.coolpup import plan_tiles_with_windows
        pdist.check_same_plan(plan)
        pdist.exchange_tile_arrays(acc, plan_tiles_with_windows(plan))
    elif world > 1 and reduce and calls is None:
        # what run_plan does between its ranks: same-plan check, then the tiles are summed (here from host arrays)
        pdist.check_same_plan(plan)
        T, W = plan["T"], 2 * plan["pad"] + 1
        f64 = np.concatenate([acc["sum"].ravel(), acc["cov_start"].ravel(), acc["cov_end"].ravel()])
        i64 = np.concatenate([acc["num"].ravel(), acc["n"].ravel()])
        f64, i64 = pdist.allreduce_arrays(f64, i64)
        acc["sum"] = f64[:T * W * W].reshape(T, W, W)
        acc["cov_start"] = f64[T * W * W:T * W * W + T * W].reshape(T, W)
        acc["cov_end"] = f64[T * W * W + T * W:].reshape(T, W)
        acc["num"] = i64[:T * W * W].reshape(T, W, W)
        acc["n"] = i64[T * W * W:]
    if (plan.get("stripe_jobs") or (plan.get("store_stripes") and world > 1)) and calls is None:
        acc["stripes"] = []
        for job in plan["stripe_jobs"]:
            fake = {"expected": job["expected"], "r0": job["r0"], "c0": job["c0"], "mode": job["mode"],
                    "tile": np.zeros(len(job["r0"]), np.int32), "flip": None, "tile_ptr": np.array([0, len(job["r0"])])}
            W = 2 * plan["pad"] + 1
            h = np.empty((len(job["r0"]), W)); v = np.empty((len(job["r0"]), W))
            if "h" in job:      # rescaled: stripes of the zoomed windows
                if big is None:
                    nb = indptr.shape[0] - 1
                    big = po.symmetric_csr(indptr, col, cnt, weight, 0, nb, 0, nb)
                fake["h"], fake["w"] = job["h"], job["w"]
                for expected, sc in iter_expected_subcalls(plan, fake):
                    win = po.windows_scipy(big, 0, 0, weight, None, expected, sc["r0"], sc["c0"], plan["pad"],
                                           job["ignore_diags"], sc["mode"], h=sc["h"], w=sc["w"])[0]
                    for k in range(len(sc["r0"])):
                        sel = np.flatnonzero((job["r0"] == sc["r0"][k]) & (job["c0"] == sc["c0"][k])
                                             & (job["h"] == sc["h"][k]) & (job["w"] == sc["w"][k]))
                        h[sel] = win[k][plan["pad"], :]; v[sel] = win[k][::-1, plan["pad"]]
                acc["stripes"].append((h, v))
                continue
            pos = {int(a) * (1 << 32) + int(b): i for i, (a, b) in enumerate(zip(job["r0"], job["c0"]))}
            for expected, sc in iter_expected_subcalls(plan, fake):
                hh, vv = po.stripes_c(indptr, col, cnt, weight, expected, sc["r0"], sc["c0"], plan["pad"],
                                      job["ignore_diags"], sc["mode"])
                for k in range(len(sc["r0"])):      # sub-calls permute the snippets: put them back by position
                    sel = np.flatnonzero((job["r0"] == sc["r0"][k]) & (job["c0"] == sc["c0"][k]))
                    h[sel] = hh[k]; v[sel] = vv[k]
            del pos
            acc["stripes"].append((h, v))
        from coolpuppy_amd.coolpup import _gather_stripes
        _gather_stripes(plan, acc)
    return acc


def oracle_windows(pu, expected, r0, c0, pad, height=None, width=None, ignore_diags=2, mode=0, coverage=False):
    """CPU stand-in for PileupEngine.extract (PileUpper._window_source hook) built on the numpy/scipy oracle."""
    from oracle import pileup_oracle as po
    indptr, col, cnt = pu._aclr.pixel_table()
    bins = pu.clr.bins()
    weight = bins[pu.clr_weight_name][:].values if pu.clr_weight_name else None
    cov = bins[pu.coverage_norm][:].values if pu.coverage_norm else None
    if not hasattr(pu, "_oracle_big"):
        nb = indptr.shape[0] - 1
        pu._oracle_big = po.symmetric_csr(indptr, col, cnt, weight, 0, nb, 0, nb)
    data, cs, ce = po.windows_scipy(pu._oracle_big, 0, 0, weight, cov, expected, r0, c0, pad, ignore_diags, mode,
                                    h=height, w=width)
    return (data, cs, ce) if coverage else (data, None, None)
