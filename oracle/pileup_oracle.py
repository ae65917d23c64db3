"""TEST INFRASTRUCTURE, NOT PRODUCT CODE — CPU restatements of coolpuppy's per-snippet hot path.

Two independent restatements of the same reference steps (reference = open2c/coolpuppy):

* :func:`pileup_c`      — ctypes call into ``oracle/liboracle.so`` (``pileup_oracle.c``), dense WxW
  window per snippet, symmetric pixel lookup, single-threaded.
* :func:`pileup_scipy`  — the reference's own operation sequence on a scipy CSR: symmetric
  ``count*w_i*w_j`` region matrix (coolpup.py:1053-1057), ``[r0:r1, c0:c1].toarray()`` (:1115-1121),
  NaN rows/cols (:1122-1123), Toeplitz expected (:1125-1133), diagonal mask (:1141-1149),
  ``data/exp`` (:1154-1156), ``rot90(flipud())`` (:128-131), then ``_add_snip``'s nansum / isfinite
  (lib/puputils.py:12-41).  Slow, used on small inputs and as the "reference-algorithm" CPU baseline.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Parity pinning: both are checked against tests/golden/*.npz, which were produced by importing the
reference's unchanged coolpup.py (see oracle/make_golden.py).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SRC = os.path.join(_HERE, "pileup_oracle.c")
_SO = os.path.join(_HERE, "liboracle.so")

MODE_OOE, MODE_EXPECTED, MODE_COV, MODE_TRANSPOSE = 0x01, 0x02, 0x04, 0x08

_lib = None


def build(force=False):
    """gcc -O2 the C restatement into oracle/liboracle.so."""
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(_SRC):
        subprocess.run(["gcc", "-O2", "-Wall", "-fopenmp", "-shared", "-fPIC", _SRC, "-o", _SO, "-lm"], check=True)
    return _SO


def _load():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.oracle_pileup.restype = C.c_int
        _lib.oracle_pileup.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int64] + \
            [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_uint32] + [C.c_void_p] * 5
        _lib.oracle_pileup_mt.restype = C.c_int
        _lib.oracle_pileup_mt.argtypes = [C.c_void_p] * 3 + [C.c_int64] + [C.c_void_p] * 3 + [C.c_int64] + \
            [C.c_void_p] * 4 + [C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_uint32, C.c_int32] + [C.c_void_p] * 5
        for name in ("oracle_pileup", "oracle_pileup_mt"):          # the same restatement over float64 pixel values
            f = getattr(_lib, name + "_f64")
            f.restype, f.argtypes = C.c_int, getattr(_lib, name).argtypes
    return _lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def empty_acc(n_tiles, pad):
    W = 2 * pad + 1
    return {
        "sum": np.zeros((n_tiles, W, W)), "num": np.zeros((n_tiles, W, W), np.int64),
        "n": np.zeros(n_tiles, np.int64), "cov_start": np.zeros((n_tiles, W)), "cov_end": np.zeros((n_tiles, W)),
    }


def _counts(cnt, f_int, f_float):
    """(contiguous pixel values, the entry point for their type): int32 counts, or float64 values for a float `pixels/count`."""
    cnt = np.asarray(cnt)
    if cnt.dtype.kind == "f":
        return np.ascontiguousarray(cnt, np.float64), f_float
    return np.ascontiguousarray(cnt, np.int32), f_int


def pileup_c(indptr, col, cnt, weight, cov, expv, r0, c0, flip, tile, n_tiles, pad, ignore_diags, mode, acc=None):
    """Accumulate into ``acc`` (created when None) with the C restatement; returns acc."""
    lib = _load()
    indptr = np.ascontiguousarray(indptr, np.int64)
    col = np.ascontiguousarray(col, np.int32)
    cnt, fn = _counts(cnt, lib.oracle_pileup, lib.oracle_pileup_f64)
    weight = None if weight is None else np.ascontiguousarray(weight, np.float64)
    cov = None if cov is None else np.ascontiguousarray(cov, np.float64)
    expv = None if expv is None else np.atleast_1d(np.ascontiguousarray(expv, np.float64))
    r0 = np.ascontiguousarray(r0, np.int32)
    c0 = np.ascontiguousarray(c0, np.int32)
    flip = None if flip is None else np.ascontiguousarray(flip, np.uint8)
    tile = np.ascontiguousarray(tile, np.int32)
    if acc is None:
        acc = empty_acc(n_tiles, pad)
    rc = fn(_p(indptr), _p(col), _p(cnt), indptr.shape[0] - 1, _p(weight), _p(cov), _p(expv),
                           0 if expv is None else expv.shape[0], _p(r0), _p(c0), _p(flip), _p(tile), r0.shape[0],
                           pad, ignore_diags, mode, _p(acc["sum"]), _p(acc["num"]), _p(acc["n"]),
                           _p(acc["cov_start"]), _p(acc["cov_end"]))
    if rc != 0:
        raise RuntimeError(f"oracle_pileup failed with {rc}")
    return acc


def pileup_c_mt(indptr, col, cnt, weight, cov, expv, r0, c0, flip, tile, n_tiles, pad, ignore_diags, mode, nthreads,
                acc=None):
    """The "best CPU" form of :func:`pileup_c` (row-sliced windows, ``nthreads`` OpenMP threads) — bench.py's
    cpu_baseline only.  Same integers as pileup_c; sums equal up to the f64 addition order across threads."""
    lib = _load()
    indptr = np.ascontiguousarray(indptr, np.int64)
    col = np.ascontiguousarray(col, np.int32)
    cnt, fn = _counts(cnt, lib.oracle_pileup_mt, lib.oracle_pileup_mt_f64)
    weight = None if weight is None else np.ascontiguousarray(weight, np.float64)
    cov = None if cov is None else np.ascontiguousarray(cov, np.float64)
    expv = None if expv is None else np.atleast_1d(np.ascontiguousarray(expv, np.float64))
    r0 = np.ascontiguousarray(r0, np.int32)
    c0 = np.ascontiguousarray(c0, np.int32)
    flip = None if flip is None else np.ascontiguousarray(flip, np.uint8)
    tile = np.ascontiguousarray(tile, np.int32)
    if acc is None:
        acc = empty_acc(n_tiles, pad)
    rc = fn(_p(indptr), _p(col), _p(cnt), indptr.shape[0] - 1, _p(weight), _p(cov), _p(expv),
                              0 if expv is None else expv.shape[0], _p(r0), _p(c0), _p(flip), _p(tile), r0.shape[0],
                              n_tiles, pad, ignore_diags, mode, int(nthreads), _p(acc["sum"]), _p(acc["num"]),
                              _p(acc["n"]), _p(acc["cov_start"]), _p(acc["cov_end"]))
    if rc != 0:
        raise RuntimeError(f"oracle_pileup_mt failed with {rc}")
    return acc


def symmetric_csr(indptr, col, cnt, weight, lo1, hi1, lo2, hi2):
    """cooler's ``matrix(sparse=True, balance=w).fetch(r1, r2).tocsr()`` for global bin ranges
    [lo1,hi1) x [lo2,hi2) of an upper-triangular pixel table: both triangles, value = count*w_i*w_j."""
    import scipy.sparse as sp
    nb = indptr.shape[0] - 1
    rows = np.repeat(np.arange(nb, dtype=np.int64), np.diff(indptr))
    cols = col.astype(np.int64)
    vals = cnt.astype(np.float64)
    if weight is not None:
        vals = vals * weight[rows] * weight[cols]
    # mirror, without doubling the main diagonal
    off = rows != cols
    R = np.concatenate([rows, cols[off]])
    Cc = np.concatenate([cols, rows[off]])
    V = np.concatenate([vals, vals[off]])
    keep = (R >= lo1) & (R < hi1) & (Cc >= lo2) & (Cc < hi2)
    return sp.coo_matrix((V[keep], (R[keep] - lo1, Cc[keep] - lo2)), shape=(hi1 - lo1, hi2 - lo2)).tocsr()


def pileup_scipy(bigdata, lo1, lo2, weight, cov, expv, r0, c0, flip, tile, n_tiles, pad, ignore_diags, mode,
                 acc=None):
    """Reference operation sequence on a region CSR ``bigdata`` (rows from global bin lo1, cols from lo2)."""
    W = 2 * pad + 1
    if acc is None:
        acc = empty_acc(n_tiles, pad)
    ar = np.arange(W)
    for s in range(len(r0)):
        rs, cs = (c0[s], r0[s]) if mode & MODE_TRANSPOSE else (r0[s], c0[s])
        t = tile[s]
        have_exp = (mode & (MODE_OOE | MODE_EXPECTED)) and expv is not None
        if have_exp:
            ev = np.atleast_1d(expv)
            if ev.shape[0] == 1:
                exp_data = np.full((W, W), ev[0])
            else:
                d = np.abs((cs + ar)[None, :] - (rs + ar)[:, None])
                exp_data = np.where(d < ev.shape[0], ev[np.minimum(d, ev.shape[0] - 1)], np.nan)
        if mode & MODE_EXPECTED:
            data = exp_data.astype(float)
        else:
            data = bigdata[rs - lo1:rs - lo1 + W, cs - lo2:cs - lo2 + W].toarray().astype(float)
            if weight is not None:
                data[np.isnan(weight[rs:rs + W]), :] = np.nan
                data[:, np.isnan(weight[cs:cs + W])] = np.nan
            if ignore_diags >= 0:
                D = ((cs + ar)[None, :] - (rs + ar)[:, None]) < ignore_diags
                data[D] = np.nan
            if (mode & MODE_COV) and cov is not None:
                acc["cov_start"][t] = np.nansum([acc["cov_start"][t], cov[rs:rs + W]], axis=0)
                acc["cov_end"][t] = np.nansum([acc["cov_end"][t], cov[cs:cs + W]], axis=0)
            if mode & MODE_OOE:
                with np.errstate(divide="ignore", invalid="ignore"):
                    data = data / (exp_data if have_exp else np.nan)
        if flip is not None and flip[s]:
            data = np.rot90(np.flipud(data))
        acc["sum"][t] = np.nansum([acc["sum"][t], data], axis=0)
        acc["num"][t] += np.isfinite(data).astype(int)
        acc["n"][t] += 1
    return acc


def coverage_numpy(indptr, col, cnt, chrom_offset, ignore_diags):
    """cooltools.api.coverage.coverage restated (its source is not in the reference tree: parity with a real
    cooltools install is unpinned): pixels with |bin1-bin2| < ignore_diags are zeroed, then every pixel's count is
    added to bin1 AND bin2 (np.bincount on each side, so a main-diagonal pixel counts twice); the cis variant
    multiplies by (chrom[bin1] == chrom[bin2]).  Returns (cov_cis, cov_tot) as float64."""
    nb = indptr.shape[0] - 1
    row = np.repeat(np.arange(nb, dtype=np.int64), np.diff(indptr))
    c = col.astype(np.int64)
    w = cnt.astype(np.float64).copy()
    if ignore_diags > 0:
        w[np.abs(row - c) < ignore_diags] = 0
    chrom = np.searchsorted(np.asarray(chrom_offset), np.arange(nb), side="right") - 1
    cis = chrom[row] == chrom[c]
    cov_cis = np.bincount(row, weights=w * cis, minlength=nb) + np.bincount(c, weights=w * cis, minlength=nb)
    cov_tot = np.bincount(row, weights=w, minlength=nb) + np.bincount(c, weights=w, minlength=nb)
    return cov_cis, cov_tot


def stripes_c(indptr, col, cnt, weight, expv, r0, c0, pad, ignore_diags, mode):
    """Centre row and reversed centre column of every snippet's masked / normalised window (the store_stripes
    branch, coolpup.py:1164-1169), obtained from the C restatement one snippet at a time: for a single snippet
    sum == data with NaN replaced by 0 and num == isfinite(data)."""
    W = 2 * pad + 1
    n = len(r0)
    h = np.empty((n, W))
    v = np.empty((n, W))
    tile = np.zeros(1, np.int32)
    for s in range(n):
        acc = pileup_c(indptr, col, cnt, weight, None, expv, r0[s:s + 1], c0[s:s + 1], None, tile, 1, pad,
                       ignore_diags, mode & ~MODE_COV)
        data = np.where(acc["num"][0] == 1, acc["sum"][0], np.where(np.isinf(acc["sum"][0]), acc["sum"][0], np.nan))
        h[s] = data[pad, :]
        v[s] = data[:, pad][::-1]
    return h, v


MODE_LOCAL = 0x20


def zoom_array(in_array, final_shape):
    """cooltools.lib.numutils.zoom_array restated (source not in the reference tree: parity with a real cooltools
    install is unpinned): blow the array up with scipy.ndimage.zoom(order=1) to the nearest integer multiple of
    final_shape (zoom factors + 1e-7), then block-average down to final_shape."""
    from scipy.ndimage import zoom
    in_array = np.asarray(in_array, dtype=np.double)
    in_shape = in_array.shape
    assert len(in_shape) == len(final_shape)
    mults = [int(np.ceil(in_shape[i] / final_shape[i])) if final_shape[i] < in_shape[i] else 1
             for i in range(len(in_shape))]
    temp_shape = tuple(i * j for i, j in zip(final_shape, mults))
    zoom_multipliers = np.array(temp_shape) / np.array(in_shape) + 0.0000001
    assert zoom_multipliers.min() >= 1
    rescaled = zoom(in_array, zoom_multipliers, order=1)
    for ind, mult in enumerate(mults):
        if mult != 1:
            sh = list(rescaled.shape)
            assert sh[ind] % mult == 0
            rescaled.shape = sh[:ind] + [sh[ind] // mult, mult] + sh[ind + 1:]
            rescaled = np.mean(rescaled, axis=ind + 1)
    assert rescaled.shape == tuple(final_shape)
    return rescaled


def rescale_snip(data, S, local):
    """PileUpper._rescale_snip on one window (reference coolpup.py:1212-1228)."""
    import warnings
    if data.size == 0 or np.all(np.isnan(data)):
        return np.zeros((S, S))
    if local:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore", category=RuntimeWarning)
            data = np.nanmean(np.dstack((data, data.T)), 2)
    nans = np.isnan(data) * 1
    data = np.nan_to_num(data)
    data = zoom_array(data, (S, S))
    nanzoom = zoom_array(nans, (S, S))
    data[np.ceil(nanzoom).astype(bool)] = np.nan
    with np.errstate(divide="ignore"):
        data = data * (1 / np.isfinite(nanzoom))
    return data


def pileup_rescaled(bigdata, lo1, lo2, weight, cov, expv, r0, c0, h, w, flip, tile, n_tiles, S, ignore_diags, mode,
                    acc=None):
    """Rescaled pile-up (variable h x w windows zoomed to S x S): the reference's operation sequence,
    _stream_snips + _rescale_snip + _add_snip, on a region CSR (rows from global bin lo1, columns from lo2)."""
    if acc is None:
        acc = empty_acc(n_tiles, (S - 1) // 2)
    local = bool(mode & MODE_LOCAL)
    for s in range(len(r0)):
        rs, cs, hh, ww = int(r0[s]), int(c0[s]), int(h[s]), int(w[s])
        t = tile[s]
        have_exp = (mode & (MODE_OOE | MODE_EXPECTED)) and expv is not None
        if have_exp:
            ev = np.atleast_1d(expv)
            if ev.shape[0] == 1:
                exp_data = np.full((hh, ww), ev[0])
            else:
                d = np.abs((cs + np.arange(ww))[None, :] - (rs + np.arange(hh))[:, None])
                exp_data = np.where(d < ev.shape[0], ev[np.minimum(d, ev.shape[0] - 1)], np.nan)
        if mode & MODE_EXPECTED:
            data = exp_data.astype(float)
        else:
            data = bigdata[rs - lo1:rs - lo1 + hh, cs - lo2:cs - lo2 + ww].toarray().astype(float)
            if weight is not None:
                data[np.isnan(weight[rs:rs + hh]), :] = np.nan
                data[:, np.isnan(weight[cs:cs + ww])] = np.nan
            if ignore_diags >= 0:
                D = ((cs + np.arange(ww))[None, :] - (rs + np.arange(hh))[:, None]) < ignore_diags
                data[D] = np.nan
            if mode & MODE_OOE:
                with np.errstate(divide="ignore", invalid="ignore"):
                    data = data / (exp_data if have_exp else np.nan)
        data = rescale_snip(data, S, local)
        if (mode & MODE_COV) and cov is not None and not (mode & MODE_EXPECTED):
            acc["cov_start"][t] = np.nansum([acc["cov_start"][t], zoom_array(cov[rs:rs + hh], (S,))], axis=0)
            acc["cov_end"][t] = np.nansum([acc["cov_end"][t], zoom_array(cov[cs:cs + ww], (S,))], axis=0)
        if flip is not None and flip[s]:
            data = np.rot90(np.flipud(data))
        acc["sum"][t] = np.nansum([acc["sum"][t], data], axis=0)
        acc["num"][t] += np.isfinite(data).astype(int)
        acc["n"][t] += 1
    return acc


def windows_scipy(bigdata, lo1, lo2, weight, cov, expv, r0, c0, pad, ignore_diags, mode, h=None, w=None):
    """The per-snippet windows _stream_snips yields (reference coolpup.py:1104-1162), before any flip:
    (data [n,W,W], cov_start [n,W], cov_end [n,W]), W = 2*pad+1; with h / w the variable-size windows are passed
    through _rescale_snip (W = rescale_size).  Coverage rows are NaN unless MODE_COV.  Checker for pup_extract."""
    W = 2 * pad + 1
    n = len(r0)
    out = np.empty((n, W, W))
    cs_out = np.full((n, W), np.nan)
    ce_out = np.full((n, W), np.nan)
    rescale = h is not None
    local = bool(mode & MODE_LOCAL)
    for s in range(n):
        rs, cs = (int(c0[s]), int(r0[s])) if mode & MODE_TRANSPOSE else (int(r0[s]), int(c0[s]))
        hh, ww = (int(h[s]), int(w[s])) if rescale else (W, W)
        if rescale and (mode & MODE_TRANSPOSE):
            hh, ww = ww, hh
        have_exp = (mode & (MODE_OOE | MODE_EXPECTED)) and expv is not None
        if have_exp:
            ev = np.atleast_1d(expv)
            if ev.shape[0] == 1:
                exp_data = np.full((hh, ww), ev[0])
            else:
                d = np.abs((cs + np.arange(ww))[None, :] - (rs + np.arange(hh))[:, None])
                exp_data = np.where(d < ev.shape[0], ev[np.minimum(d, ev.shape[0] - 1)], np.nan)
        if mode & MODE_EXPECTED:
            data = exp_data.astype(float)
        else:
            data = bigdata[rs - lo1:rs - lo1 + hh, cs - lo2:cs - lo2 + ww].toarray().astype(float)
            if weight is not None:
                data[np.isnan(weight[rs:rs + hh]), :] = np.nan
                data[:, np.isnan(weight[cs:cs + ww])] = np.nan
            if ignore_diags >= 0:
                D = ((cs + np.arange(ww))[None, :] - (rs + np.arange(hh))[:, None]) < ignore_diags
                data[D] = np.nan
            if mode & MODE_OOE:
                with np.errstate(divide="ignore", invalid="ignore"):
                    data = data / (exp_data if have_exp else np.nan)
        if rescale:
            data = rescale_snip(data, W, local)
        out[s] = data
        if (mode & MODE_COV) and cov is not None and not (mode & MODE_EXPECTED):
            cs_out[s] = zoom_array(cov[rs:rs + hh], (W,)) if rescale else cov[rs:rs + hh]
            ce_out[s] = zoom_array(cov[cs:cs + ww], (W,)) if rescale else cov[cs:cs + ww]
    return out, cs_out, ce_out
