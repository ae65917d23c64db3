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


def _add_snip(outdict, key, snip, extra_funcs=None):
    """Add one snippet to the pile-up of its group (reference lib/puputils.py:12-41): the first snippet of a group
    seeds the entry, later ones are nansum-ed into data / coverage, counted in num (finite cells) and n, and have
    their stripes / coordinates appended; every extra function then maps (entry, snip) -> entry."""
    if key not in outdict:
        entry = {k: snip[k] for k in ("data", "cov_start", "cov_end")}
        entry["coordinates"] = [snip["coordinates"]]
        entry["horizontal_stripe"] = [snip["horizontal_stripe"]]
        entry["vertical_stripe"] = [snip["vertical_stripe"]]
        entry["num"] = np.isfinite(snip["data"]).astype(int)
        entry["n"] = 1
        outdict[key] = entry
    else:
        entry = outdict[key]
        entry["data"] = np.nansum([entry["data"], snip["data"]], axis=0)
        entry["num"] += np.isfinite(snip["data"]).astype(int)
        entry["cov_start"] = np.nansum([entry["cov_start"], snip["cov_start"]], axis=0)
        entry["cov_end"] = np.nansum([entry["cov_end"], snip["cov_end"]], axis=0)
        entry["n"] += 1
        for k in ("horizontal_stripe", "vertical_stripe", "coordinates"):
            entry[k] = entry[k] + [snip[k]]
    if extra_funcs is not None:
        for _, func in extra_funcs.items():
            outdict[key] = func(outdict[key], snip)


def sum_pups(pup1, pup2, extra_funcs={}):
    """Sum two pile-up dicts (data, num, n, cov_start, cov_end, stripes, coordinates); NaN/inf in data are replaced
    first — in place, on both inputs — exactly like the reference's ``np.nan_to_num`` (lib/puputils.py:97-98).
    With extra_funcs the reference REPLACES the summed pile-up by the return value of func(pup1, pup2)
    (lib/puputils.py:110-112); that is kept, because callers of the callback API see it."""
    pup1["data"] = np.nan_to_num(pup1["data"])
    pup2["data"] = np.nan_to_num(pup2["data"])
    out = {
        "data": pup1["data"] + pup2["data"],
        "cov_start": pup1["cov_start"] + pup2["cov_start"],
        "cov_end": pup1["cov_end"] + pup2["cov_end"],
        "n": pup1.get("n", 1) + pup2.get("n", 1),
        "num": pup1.get("num", np.isfinite(pup1["data"]).astype(int))
        + pup2.get("num", np.isfinite(pup2["data"]).astype(int)),
        "horizontal_stripe": pup1["horizontal_stripe"] + pup2["horizontal_stripe"],
        "vertical_stripe": pup1["vertical_stripe"] + pup2["vertical_stripe"],
        "coordinates": pup1["coordinates"] + pup2["coordinates"],
    }
    if extra_funcs:
        for _, func in extra_funcs.items():
            out = func(pup1, pup2)
    return pd.Series(out)


def accumulate_values(dict1, dict2, key):
    """An extra_sum_func: collect dict2[key] into the flat list dict1[key] (reference lib/puputils.py:244-253)."""
    assert key in dict2, f"{key} not in dict2"
    if key in dict1:
        dict1[key] = list(_collapse([dict1[key], dict2[key]]))
    else:
        dict1[key] = [dict2[key]]
    return dict1


def bin_distance(snip, band_edges="default"):
    """Per-snippet form of bin_distance_intervals (reference lib/puputils.py:193-215): adds 'distance_band'."""
    if isinstance(band_edges, str) and band_edges == "default":
        band_edges = np.append([0], 50000 * 2 ** np.arange(30))
    i = np.searchsorted(band_edges, snip["distance"])
    snip["distance_band"] = tuple(band_edges[i - 1: i + 1])
    return snip


def group_by_region(snip):
    """A postprocess_func: count the snippet once for the feature on each side (reference lib/puputils.py:218-223)."""
    for side in ("1", "2"):
        s = snip.copy()
        s["group"] = (s["chrom" + side], s["start" + side], s["end" + side])
        yield s


def get_score(pup, center=3, ignore_central=3):
    """One number for any pile-up (reference lib/puputils.py:44-85): off-diagonal -> mean of the central `center`
    pixels; local -> insulation strength; local and rescaled -> domain score."""
    from .numutils import get_domain_score, get_enrichment, get_insulation_strength
    if not pup["local"]:
        return get_enrichment(pup["data"], center)
    if pup["rescale"]:
        return get_domain_score(pup["data"], pup["rescale_flank"])
    return get_insulation_strength(pup["data"], ignore_central)


_NOT_COMPARED = ["control_n", "control_num", "n", "num", "clr", "chroms", "minshift", "expected_file", "group", "maxshift",
                 "mindist", "maxdist", "subset", "seed", "data", "horizontal_stripe", "vertical_stripe", "cooler",
                 "features", "outname", "coordinates"]


def divide_pups(pup1, pup2):
    """Ratio of two single-row pile-up frames of identical geometry (reference lib/puputils.py:116-165): data1 / data2,
    n summed, annotation columns that differ are reported with a warning, stripes divided only when both hold the same
    coordinates (inf / NaN quotients -> 0)."""
    import logging
    import warnings
    if pup1.shape[0] > 1 or pup2.shape[0] > 1:
        raise ValueError("Pileups cannot contain multiple conditions")
    pup1, pup2 = pup1.reset_index(drop=True), pup2.reset_index(drop=True)
    out = pup1.drop(columns=list(set(_NOT_COMPARED) & set(pup1.columns)))
    for col in out.columns:
        if np.all(np.sort(pup1[col]) != np.sort(pup2[col])):
            warnings.warn(f"Note that {col} is different between the two pileups")
    out["data"] = pup1["data"] / pup2["data"]
    out["clrs"] = str(pup1["clr"]) + "/" + str(pup2["clr"])
    out["n"] = pup1["n"] + pup2["n"]
    if {"vertical_stripe", "horizontal_stripe"}.issubset(pup1.columns):
        if np.all(np.sort(pup1["coordinates"]) == np.sort(pup2["coordinates"])):
            out["coordinates"] = pup1["coordinates"]
            for stripe in ("vertical_stripe", "horizontal_stripe"):
                out[stripe] = (pup1[stripe] / pup2[stripe]).apply(lambda x: np.where(np.isin(x, [np.inf, np.nan]), 0, x))
        else:
            logging.info("Stripes cannot be divided, coordinates differ between pups")
    return out


def norm_coverage(snip):
    """data /= outer(cov_start, cov_end) / nanmean(...) ; NaN -> 0 (reference lib/puputils.py:168-190)."""
    coverage = np.outer(snip["cov_start"], snip["cov_end"])
    with np.errstate(divide="ignore", invalid="ignore"):
        coverage = coverage / np.nanmean(coverage)
        snip["data"] = snip["data"] / coverage
    snip["data"][np.isnan(snip["data"])] = 0
    return snip


def _tile_frame(acc, kind, order, contrib, gid, G, grouped):
    """DataFrame indexed by group key (first-appearance order) with the summed tile of each group."""
    rows = {}
    for key in order[kind]:
        t = kind * G + gid[key]
        data = acc["sum"][t].copy()
        # sum_pups() passes data through nan_to_num whenever >= 2 pile-ups are merged: +inf becomes the
        # largest double (lib/puputils.py:97-98).  A group held by a single region is never merged; the
        # "all" row of a grouped pile-up always is (coolpup.py:1271-1282).
        merged = contrib[kind].get(key, 0) >= 2 or (grouped and key == "all" and acc["n"][t] > 0)
        if merged:
            data = np.nan_to_num(data)
        rows[key] = {"data": data, "num": acc["num"][t].copy(), "n": int(acc["n"][t]),
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


def finalize_pileups(pu, acc, order, contrib, gid, G, groupby, want_control, n_regions, grouped=None, stripes=None):
    """Tail of pileupsWithControl (coolpup.py:1533-1654) on summed tiles -> annotated DataFrame."""
    if grouped is None:
        grouped = bool(groupby)
    roi = _tile_frame(acc, KIND_ROI, order, contrib, gid, G, grouped)
    ctrl = _tile_frame(acc, KIND_CONTROL, order, contrib, gid, G, grouped) if want_control else None
    return _finalize_frames(pu, roi, ctrl, groupby, want_control, stripes)


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

    for name in _ANNOTATION_ATTRS:
        if not hasattr(pu, name):
            continue
        attr = getattr(pu, name)
        if isinstance(attr, list):
            attr = str(attr)
        if name == "clr":
            attr = os.path.abspath(attr.filename)
        normalized_roi[name] = attr
    return normalized_roi
