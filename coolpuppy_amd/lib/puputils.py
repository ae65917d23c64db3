"""Host-side merge / normalisation of fetched pile-up tiles.

Restates, on whole (kind, group) tiles instead of per-snippet dicts, what the reference does after the
per-region loops: ``sum_pups`` (coolpuppy/lib/puputils.py:88-113), ``norm_coverage`` (:168-190) and the
tail of ``PileUpper.pileupsWithControl`` (coolpuppy/coolpup.py:1511-1654).  All of it is O(groups * W^2).
"""
import os

import numpy as np
import pandas as pd

KIND_ROI, KIND_CONTROL = 0, 1

# scalar attributes copied into every output row, in the reference's column order
# (PileUpper.__dict__ iteration minus its exclude list, coolpup.py:1628-1653)
_ANNOTATION_ATTRS = [
    "clr", "resolution", "flank", "rescale_flank", "chroms", "minshift", "maxshift", "nshifts", "trans", "mindist",
    "maxdist", "local", "subset", "seed", "clr_weight_name", "expected", "expected_value_col", "ooe", "control",
    "pad_bins", "coverage_norm", "rescale", "rescale_size", "flip_negative_strand", "ignore_diags", "store_stripes",
    "nproc", "ignore_group_order",
]


def _collapse(items):
    """Flatten nested iterables down to dicts (more_itertools.collapse(..., base_type=dict))."""
    if isinstance(items, (dict, str, bytes)):
        yield items
        return
    try:
        it = iter(items)
    except TypeError:
        yield items
        return
    for x in it:
        yield from _collapse(x)


_PUP_LISTS = ("horizontal_stripe", "vertical_stripe", "coordinates")


class SnipAccumulator:
    """Per-(kind) pile-ups of a snippet stream for the CALLBACK path (the GPU produced the windows, user Python sees
    every snippet): ``entries[group]`` is the pile-up dict the reference's API exposes — data (sum), num, n,
    cov_start, cov_end and the per-snippet lists — kept as running arrays updated in place.

    Observable result = the reference's ``_add_snip`` (coolpuppy/lib/puputils.py:12-41) applied snippet by snippet:
    the first snippet of a group is kept as is (its NaN cells stay NaN until a second snippet arrives), afterwards
    data / coverage are nan-sums, ``num`` counts finite cells, ``n`` snippets.  Unlike the reference nothing is
    re-allocated per snippet: NaN is folded to 0 when the second snippet arrives and lists grow by append (the
    reference rebuilds three Python lists per snippet, its quadratic term)."""

    def __init__(self, extra_funcs=None):
        self.entries = {}
        self._extra = list(extra_funcs.values()) if extra_funcs else []

    @staticmethod
    def _nan_as_zero(a):
        return np.where(np.isnan(a), 0.0, a)

    def add(self, key, snip):
        e = self.entries.get(key)
        if e is None:
            e = {"data": snip["data"], "cov_start": snip["cov_start"], "cov_end": snip["cov_end"],
                 "coordinates": [snip["coordinates"]], "horizontal_stripe": [snip["horizontal_stripe"]],
                 "vertical_stripe": [snip["vertical_stripe"]],
                 "num": np.isfinite(snip["data"]).astype(int), "n": 1}
            self.entries[key] = e
        else:
            if e["n"] == 1:          # the seed may be the caller's array and may hold NaN: own, NaN-free copies from now on
                for f in ("data", "cov_start", "cov_end"):
                    e[f] = self._nan_as_zero(np.asarray(e[f], dtype=float))
            for f in ("data", "cov_start", "cov_end"):
                e[f] = e[f] + self._nan_as_zero(np.asarray(snip[f], dtype=float))
            e["num"] += np.isfinite(snip["data"])
            e["n"] += 1
            for f in _PUP_LISTS:
                e[f].append(snip[f])
        for func in self._extra:     # user hooks see (running pile-up, snippet) and return the pile-up to keep
            self.entries[key] = e = func(e, snip)


def _add_snip(outdict, key, snip, extra_funcs=None):
    """Function form of :meth:`SnipAccumulator.add` on a plain {group: pile-up} dict (the name the reference's
    callers know, coolpuppy/lib/puputils.py:12)."""
    acc = SnipAccumulator(extra_funcs)
    acc.entries = outdict
    acc.add(key, snip)


def sum_pups(pup1, pup2, extra_funcs={}):
    """Merge two pile-up dicts of one group (two regions): additive fields add, per-snippet lists concatenate.  Both
    inputs have NaN / inf in ``data`` replaced first, IN PLACE (np.nan_to_num: +inf -> 1.8e308) — the reference does
    (coolpuppy/lib/puputils.py:97-98) and its callers see it; a missing ``n`` counts as 1, a missing ``num`` as the
    finite cells.  With extra_funcs the merged pile-up is whatever the last hook returns for (pup1, pup2) — the
    reference's behaviour (:110-112), kept because hooks written for it rely on it."""
    for pup in (pup1, pup2):
        pup["data"] = np.nan_to_num(pup["data"])
    finite = [np.isfinite(pup["data"]).astype(int) for pup in (pup1, pup2)]
    merged = {f: pup1[f] + pup2[f] for f in ("data", "cov_start", "cov_end")}
    merged["n"] = pup1.get("n", 1) + pup2.get("n", 1)
    merged["num"] = pup1.get("num", finite[0]) + pup2.get("num", finite[1])
    for f in _PUP_LISTS:
        merged[f] = pup1[f] + pup2[f]
    for func in (extra_funcs or {}).values():
        merged = func(pup1, pup2)
    return pd.Series(merged)


def accumulate_values(dict1, dict2, key):
    """extra_sum_funcs hook: gather dict2[key] into the flat list dict1[key] (reference lib/puputils.py:244-253)."""
    if key not in dict2:
        raise AssertionError(f"{key} not in dict2")
    gathered = list(_collapse([dict1[key]])) if key in dict1 else []
    gathered.extend(_collapse([dict2[key]]))
    dict1[key] = gathered
    return dict1


def bin_distance(snip, band_edges="default"):
    """postprocess_func: label one snippet with the distance band its 'distance' falls in (reference
    lib/puputils.py:193-215; the per-snippet twin of coolpup.bin_distance_intervals)."""
    edges = np.append([0], 50000 * 2 ** np.arange(30)) if isinstance(band_edges, str) and band_edges == "default" \
        else band_edges
    hi = int(np.searchsorted(edges, snip["distance"]))
    snip["distance_band"] = tuple(edges[hi - 1: hi + 1])
    return snip


def group_by_region(snip):
    """postprocess_func: the snippet once per side, grouped under that side's feature (reference
    lib/puputils.py:218-223)."""
    for side in "12":
        twin = dict(snip)
        twin["group"] = tuple(twin[f + side] for f in ("chrom", "start", "end"))
        yield twin


def norm_coverage(snip):
    """Divide a summed pile-up by the outer product of its coverage vectors, scaled to mean 1; cells that end up NaN
    become 0, inf stays (reference lib/puputils.py:168-190)."""
    with np.errstate(divide="ignore", invalid="ignore"):
        expected_cov = np.multiply.outer(snip["cov_start"], snip["cov_end"])
        q = snip["data"] / (expected_cov / np.nanmean(expected_cov))
    snip["data"] = np.where(np.isnan(q), 0.0, q)
    return snip


def _tile_frame(acc, kind, order, gid, G):
    """DataFrame indexed by group key (first-appearance order) with the summed tile of each group.  (What sum_pups'
    nan_to_num does to +inf cells when pile-ups are merged has been applied to acc["sum"] already,
    PileUpper._merge_inf_cells.)"""
    rows = {}
    for key in order[kind]:
        t = kind * G + gid[key]
        rows[key] = {"data": acc["sum"][t].copy(), "num": acc["num"][t].copy(), "n": int(acc["n"][t]),
                     "cov_start": acc["cov_start"][t].copy(), "cov_end": acc["cov_end"][t].copy()}
    df = pd.DataFrame(list(rows.values()), columns=["data", "num", "n", "cov_start", "cov_end"])
    df["n"] = df["n"].astype(object)     # the reference's frames hold Python ints in object columns
    df.index = pd.Index(list(rows.keys()), tupleize_cols=False)
    return df


def _copy_array_halves(x):
    """Mirror the right half of each stripe onto the left (reference lib/numutils.py:6-9)."""
    cntr = int(np.floor(x.shape[1] / 2))
    x[:, : (cntr + 1)] = np.fliplr(x[:, cntr:])
    return x


def finalize_pileups(pu, acc, order, gid, G, groupby, want_control, grouped=None, stripes=None, any_order=False):
    """Tail of pileupsWithControl (coolpup.py:1533-1654) on summed tiles -> annotated DataFrame.  any_order: the caller sorts the rows
    itself (by-window), so when ROI and control hold the SAME groups in different orders of first appearance — what random-shift controls
    do to a per-feature pile-up — the rows come in the ROI order, control tiles matched by group, instead of in the order pandas' index
    alignment would give them; the per-row frame arithmetic took 0.3 s for 37 k features."""
    if grouped is None:
        grouped = bool(groupby)
    same = not want_control or list(order[KIND_CONTROL]) == list(order[KIND_ROI])
    if stripes is None and not os.environ.get("COOLPUPPY_AMD_FRAME_FINALISER"):
        if same:
            return _finalize_tiles(pu, acc, order[KIND_ROI], gid, G, groupby, want_control)
        if any_order and len(order[KIND_ROI]) > 256:
            # the groups both kinds hold on the arrays at once, in the ROI order; the few that only one kind holds (a feature at a
            # chromosome's end whose windows — or whose shifted copies — all leave the region) through the frame arithmetic itself,
            # so that what pandas' index alignment makes of a missing side (NaN cells, NaN counts) is what they get
            in_roi, in_ctrl = set(order[KIND_ROI]), set(order[KIND_CONTROL])
            both = [k for k in order[KIND_ROI] if k in in_ctrl]
            odd = {KIND_ROI: [k for k in order[KIND_ROI] if k not in in_ctrl] + ["all"],
                   KIND_CONTROL: [k for k in order[KIND_CONTROL] if k not in in_roi] + ["all"]}
            if "all" in in_roi and "all" in in_ctrl and len(odd[KIND_ROI]) + len(odd[KIND_CONTROL]) <= max(64, len(both) // 8):
                fast = _finalize_tiles(pu, acc, both, gid, G, groupby, want_control)
                if len(odd[KIND_ROI]) + len(odd[KIND_CONTROL]) == 2:
                    return fast
                slow = _finalize_frames(pu, _tile_frame(acc, KIND_ROI, odd, gid, G), _tile_frame(acc, KIND_CONTROL, odd, gid, G),
                                        groupby, want_control, None)
                slow = slow[[not (isinstance(g, str) and g == "all") for g in slow["group"]]]
                return pd.concat([fast, slow], ignore_index=True)
    roi = _tile_frame(acc, KIND_ROI, order, gid, G)
    ctrl = _tile_frame(acc, KIND_CONTROL, order, gid, G) if want_control else None
    return _finalize_frames(pu, roi, ctrl, groupby, want_control, stripes)


def _object_column(arr):
    """An object array whose entry i is arr[i] (a view): what a frame column of per-row arrays holds."""
    return np.fromiter(arr, dtype=object, count=len(arr))


def _annotation(pu):
    """(name, value) of the scalar attributes every output row carries (coolpup.py:1628-1653), lists as their str()."""
    out = []
    for name in _ANNOTATION_ATTRS:
        if not hasattr(pu, name):
            continue
        attr = getattr(pu, name)
        if isinstance(attr, list):
            attr = str(attr)
        if name == "clr":
            attr = os.path.abspath(attr.filename)
        out.append((name, attr))
    return out


def _normalise_tiles(S, N, Sc, Nc):
    """data = S / N [/ (Sc / Nc)], +inf -> NaN (reference coolpup.py:1533-1545), in place of S (and Sc): big tile arrays (by-window: a
    tile per feature) go through the library's multi-threaded pass (pup_host_normalise_tiles: same operations, same order),
    small or unusual ones through numpy."""
    arrs = [S, N] + ([Sc, Nc] if Sc is not None else [])
    if S.size >= 200_000 and all(isinstance(a, np.ndarray) and a.flags.c_contiguous and a.flags.writeable for a in arrs) \
            and S.dtype == np.float64 and N.dtype == np.int64 and N.shape == S.shape \
            and (Sc is None or (Sc.dtype == np.float64 and Nc.dtype == np.int64 and Sc.shape == S.shape and Nc.shape == S.shape)):
        from .. import _ffi
        ptr = lambda a: None if a is None else a.ctypes.data      # noqa: E731
        if _ffi.lib().pup_host_normalise_tiles(ptr(S), ptr(N), ptr(Sc), ptr(Nc), S.size) == 0:
            return S
    with np.errstate(divide="ignore", invalid="ignore"):
        data = np.divide(S, N, out=S)
        if Sc is not None:
            data = np.divide(data, np.divide(Sc, Nc, out=Sc), out=data)
    np.putmask(data, data == np.inf, np.nan)
    return data


def _finalize_tiles(pu, acc, keys, gid, G, groupby, want_control):
    """_finalize_frames for the usual case — no stripes, every group piled up for both kinds — on the [T][W][W] arrays at
    once instead of one pandas operation per step and row: the same frame (columns, order, dtypes, values; tests compare the
    two), assembled in one construction.  A by-window pile-up has a row per feature (tens of thousands): the per-row form
    spent a second there."""
    import warnings
    keys = list(keys)
    n_rows = len(keys)
    t = np.array([gid[k] for k in keys], np.int64)

    run = n_rows > 0 and bool(np.all(np.diff(t) == 1))       # tiles in output order (by-window: one per feature): views, no gather

    def tiles(kind):
        tt = slice(int(t[0]) + kind * G, int(t[-1]) + kind * G + 1) if run else t + kind * G
        return (acc["sum"][tt], acc["num"][tt], acc["n"][tt], acc["cov_start"][tt], acc["cov_end"][tt])

    def cov_normalised(S, cs, ce):
        with np.errstate(divide="ignore", invalid="ignore"), warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            ec = cs[:, :, None] * ce[:, None, :]
            mean = np.nanmean(ec.reshape(len(ec), -1), axis=1)
            q = S / (ec / mean[:, None, None])
        return np.where(np.isnan(q), 0.0, q)

    S, N, nn, cs, ce = tiles(KIND_ROI)
    if want_control:
        Sc, Nc, nc, csc, cec = tiles(KIND_CONTROL)
    if pu.coverage_norm:
        S = cov_normalised(S, cs, ce)
        if pu.control:
            Sc = cov_normalised(Sc, csc, cec)
        elif pu.expected:
            warnings.warn("Expected can not be normalized to coverage", stacklevel=3)
    # (in place: S / Sc are this function's own copies, or views of the fetched tiles nobody reads again — a by-window pile-up holds
    # a tile per feature, and every temporary of that size is 130 MB of fresh pages)
    data = _normalise_tiles(S, N, Sc if want_control else None, Nc if want_control else None)
    if pu.local:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            data = np.nanmean(np.stack((data, data.transpose(0, 2, 1)), axis=-1), axis=-1)
    ints = lambda v: np.asarray(v, np.int64).astype(object)      # noqa: E731  Python ints in an object column
    group = np.fromiter(keys, dtype=object, count=n_rows)       # (entry i IS keys[i]: tuples stay tuples)
    cols = {}
    if groupby:
        gdf = pd.DataFrame([("all",) * len(groupby) if (isinstance(k, str) and k == "all") else k for k in keys],
                           columns=groupby)
        for val in reversed(groupby):
            cols[val] = gdf[val].values
    cols["group"] = group
    cols["data"] = _object_column(data)
    # (DataFrame.apply(norm_coverage, axis=1) re-infers the dtypes of the frame it rebuilds: n comes back as int64 there)
    if want_control:
        cols["control_n"] = np.asarray(nc, np.int64) if (pu.coverage_norm and pu.control) else ints(nc)
        cols["control_num"] = _object_column(Nc)
    cols["n"] = np.asarray(nn, np.int64) if pu.coverage_norm else ints(nn)
    cols["num"] = _object_column(N)
    import logging
    logging.getLogger("coolpuppy").info(f"Total number of piled up windows: {int(nn[keys.index('all')])}")
    index = pd.RangeIndex(n_rows)
    for name, attr in _annotation(pu):
        cols[name] = attr
    return pd.DataFrame(cols, index=index, copy=False)


def merge_region_pups(per_region, extra_funcs=None):
    """[{group: pup}, ...] (one dict per region, in region order) -> DataFrame indexed by group, one column per pup
    field: groups in order of first appearance, each reduced over the regions holding it with sum_pups — a group
    held by a single region is passed through untouched (reference coolpup.py:1511-1531)."""
    from functools import partial, reduce
    sum_func = partial(sum_pups, extra_funcs=extra_funcs)
    keys = list(dict.fromkeys(k for d in per_region for k in d))
    merged = {}
    for k in keys:
        held = [pd.Series(d[k]) for d in per_region if k in d]
        merged[k] = reduce(sum_func, held)
    # fields as rows, then transposed: every column ends up with object dtype, as in the reference's frames
    out = pd.DataFrame(dict(enumerate(merged.values()))).T
    out.index = pd.Index(keys, tupleize_cols=False)
    return out


def finalize_callback_pileups(pu, pileups, groupby, want_control, extra_sum_funcs=None):
    """Callback mode: per-region host pile-ups ({"ROI": {...}, "control": {...}} each) -> annotated DataFrame."""
    roi = merge_region_pups([p["ROI"] for p in pileups], extra_sum_funcs)
    ctrl = merge_region_pups([p["control"] for p in pileups], extra_sum_funcs) if want_control else None
    stripes = None
    if pu.store_stripes:
        # coordinates were joined with "." per snippet and are split again here (coolpup.py:1170-1182, 1557-1560)
        stripes = {}
        for i, key in enumerate(roi.index):          # positional: tuple keys would be read as multi-axis labels
            co = np.vstack([c.split(".") for c in roi["coordinates"].iloc[i]])
            stripes[key] = (co, np.vstack(roi["horizontal_stripe"].iloc[i]), np.vstack(roi["vertical_stripe"].iloc[i]))
    return _finalize_frames(pu, roi, ctrl, groupby, want_control, stripes, extra_sum_funcs=extra_sum_funcs)


def _finalize_frames(pu, roi, ctrl, groupby, want_control, stripes=None, extra_sum_funcs=None):
    import warnings
    if pu.coverage_norm:
        roi = roi.apply(norm_coverage, axis=1)
        if pu.control:
            ctrl = ctrl.apply(norm_coverage, axis=1)
        elif pu.expected:
            warnings.warn("Expected can not be normalized to coverage", stacklevel=2)
    with np.errstate(divide="ignore", invalid="ignore"):
        normalized_roi = pd.DataFrame(roi["data"] / roi["num"], columns=["data"])
        if want_control:
            normalized_control = pd.DataFrame(ctrl["data"] / ctrl["num"], columns=["data"])
            normalized_roi = normalized_roi / normalized_control
            normalized_roi["control_n"] = ctrl["n"]
            normalized_roi["control_num"] = ctrl["num"]
    normalized_roi["data"] = normalized_roi["data"].apply(lambda x: np.where(x == np.inf, np.nan, x))
    normalized_roi["n"] = roi["n"]
    normalized_roi["num"] = roi["num"]
    if stripes is not None:
        # per-snippet centre row / column and coordinates of the ROI snippets (coolpup.py:1556-1600)
        keys = list(roi.index)
        co = pd.Series([stripes[k][0] if k in stripes else np.nan for k in keys], index=roi.index, dtype=object)
        hs = pd.Series([stripes[k][1] if k in stripes else np.nan for k in keys], index=roi.index, dtype=object)
        vs = pd.Series([stripes[k][2] if k in stripes else np.nan for k in keys], index=roi.index, dtype=object)
        normalized_roi["coordinates"] = co
        normalized_roi["horizontal_stripe"] = hs
        normalized_roi["vertical_stripe"] = vs
        if want_control:
            ctrl_all = normalized_control["data"]["all"]
            cntr = int(np.floor(ctrl_all.shape[0] / 2))
            ch = np.array(ctrl_all[cntr, :], dtype=float)
            cv = np.array(ctrl_all[:, cntr][::-1], dtype=float)
            with np.errstate(divide="ignore", invalid="ignore"):
                normalized_roi["horizontal_stripe"] = normalized_roi["horizontal_stripe"].apply(lambda x: np.divide(x, ch))
                normalized_roi["vertical_stripe"] = normalized_roi["vertical_stripe"].apply(lambda x: np.divide(x, cv))
        if pu.local:
            for c in ("vertical_stripe", "horizontal_stripe"):
                normalized_roi[c] = normalized_roi[c].apply(lambda x: _copy_array_halves(np.array(x, dtype=float)))

    if pu.local:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            normalized_roi["data"] = normalized_roi["data"].apply(lambda x: np.nanmean(np.dstack((x, x.T)), 2))
    n = normalized_roi.loc["all", "n"]
    normalized_roi = normalized_roi.reset_index().rename(columns={"index": "group"})
    if groupby:
        normalized_roi[groupby] = pd.DataFrame(
            [("all",) * len(groupby) if (isinstance(i, str) and i == "all") else i
             for i in normalized_roi["group"].to_list()],
            columns=groupby,
        )
        for val in groupby:
            normalized_roi.insert(0, val, normalized_roi.pop(val))
    if extra_sum_funcs:
        for key in extra_sum_funcs:
            normalized_roi[key] = roi[key].values
            if pu.control:
                # index-aligned against the already re-numbered frame, as in the reference (coolpup.py:1621-1622)
                normalized_roi[f"control_{key}"] = ctrl[key]
    import logging
    logging.getLogger("coolpuppy").info(f"Total number of piled up windows: {int(n)}")

    for name, attr in _annotation(pu):
        normalized_roi[name] = attr
    return normalized_roi
