"""Scores computed on a FINISHED pile-up (O(W^2) numpy; reference coolpuppy/lib/numutils.py:6-183).

Not on the per-snippet path — kept so that code written against the reference (plot annotations, the per-snippet
callbacks of TAD_score.ipynb: ``postprocess_func=lambda s: {**s, "score": get_domain_score(s["data"])}``) runs
unchanged against this package.  Restated from the reference's definitions; `tests/test_host_misc.py` checks them
against values the reference's own functions produced (tests/golden/numutils.npz).
"""
import numpy as np


def _copy_array_halves(x):
    """Mirror the right half of every row onto its left half, in place (reference :6-9)."""
    mid = int(np.floor(x.shape[1] / 2))
    x[:, : mid + 1] = np.fliplr(x[:, mid:])
    return x


def _fill_diag(arr, value, k):
    """A copy of arr with its k-th diagonal set to value (cooltools.numutils.fill_diag)."""
    out = np.array(arr, dtype=float, copy=True)
    n, m = out.shape
    i = np.arange(max(0, -k), min(n, m - k))
    out[i, i + k] = value
    return out


def corner_cv(amap, i=4):
    """Coefficient of variation of the i x i upper-left and lower-right corners: how noisy the pile-up is (:12-33)."""
    vals = np.concatenate((amap[:i, :i], amap[-i:, -i:]))
    vals = vals[np.isfinite(vals)]
    return np.std(vals) / np.mean(vals)


def norm_cis(amap, i=3):
    """Divide by the mean level of the two i x i diagonal corners; i = 0 leaves the pile-up alone (:36-57)."""
    if i <= 0:
        return amap
    return amap / np.nanmean(amap[:i, :i] + amap[-i:, -i:]) * 2


def get_enrichment(amap, n):
    """Mean of the central n x n pixels (:60-79)."""
    c = amap.shape[0] // 2
    if c < n:
        raise ValueError(f"Central pixel value {n} is too large, can be maximum {c}")
    h = n // 2
    return np.nanmean(amap[c - h: c + h + 1, c - h: c + h + 1])


def _central_side(amap, flank):
    c = amap.shape[0] / (flank * 2 + 1)
    assert int(c) == c
    return int(c)


def get_local_enrichment(amap, flank=1):
    """Mean of the central square of a rescaled pile-up whose features span 1/(2*flank+1) of it (:82-103)."""
    c = _central_side(amap, flank)
    return np.nanmean(amap[c:-c, c:-c])


def get_domain_score(amap, flank=1):
    """Central square over the rectangles above it and to its right, x2 (Flyamer et al. 2017; :106-132)."""
    c = _central_side(amap, flank)
    inside = np.nansum(amap[c:-c, c:-c])
    above = np.nansum(amap[:c, c:-c])
    right = np.nansum(amap[c:-c, -c:])
    return inside / (above + right) * 2


def get_insulation_strength(amap, ignore_central=0, ignore_diags=2):
    """Mean of the two on-diagonal corners over the mean of the two off-diagonal ones, after blanking the first
    ignore_diags diagonals and leaving out ignore_central middle bins (:135-161)."""
    for d in range(ignore_diags):
        amap = _fill_diag(amap, np.nan, d)
        if d != 0:
            amap = _fill_diag(amap, np.nan, -d)
    if ignore_central != 0 and ignore_central % 2 != 1:
        raise ValueError(f"ignore_central has to be odd (or 0), got {ignore_central}")
    i = (amap.shape[0] - ignore_central) // 2
    intra = np.nanmean(np.concatenate([amap[:i, :i].ravel(), amap[-i:, -i:].ravel()]))
    inter = np.nanmean(np.concatenate([amap[:i, -i:].ravel(), amap[-i:, :i].ravel()]))
    return intra / inter
