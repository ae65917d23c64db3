"""Synthetic Hi-C coolers and feature sets for the BASELINE.json configurations.

The reference's ``.cool`` fixtures are not in its tree (``.MISSING_LARGE_BLOBS``), so every workload
here is generated: a distance-decay upper-triangular pixel table per chromosome plus balancing
weights with masked bins and a ``cov_tot_raw`` column.  Generation is deterministic in the seed.
"""
import numpy as np
import pandas as pd

from coolpuppy_amd.cooler_lite import ArrayCooler

# chromosome lengths (bp) of the assemblies the BASELINE configs name
MM9 = {
    "chr1": 197195432, "chr2": 181748087, "chr3": 159599783, "chr4": 155630120, "chr5": 152537259,
    "chr6": 149517037, "chr7": 152524553, "chr8": 131738871, "chr9": 124076172, "chr10": 129993255,
    "chr11": 121843856, "chr12": 121257530, "chr13": 120284312, "chr14": 125194864, "chr15": 103494974,
    "chr16": 98319150, "chr17": 95272651, "chr18": 90772031, "chr19": 61342430, "chrX": 166650296,
}
HG38 = {
    "chr1": 248956422, "chr2": 242193529, "chr3": 198295559, "chr4": 190214555, "chr5": 181538259,
    "chr6": 170805979, "chr7": 159345973, "chr8": 145138636, "chr9": 138394717, "chr10": 133797422,
    "chr11": 135086622, "chr12": 133275309, "chr13": 114364328, "chr14": 107043718, "chr15": 101991189,
    "chr16": 90338345, "chr17": 83257441, "chr18": 80373285, "chr19": 58617616, "chr20": 64444167,
    "chr21": 46709983, "chr22": 50818468, "chrX": 156040895,
}


def _chrom_block(nb, lam, max_log10, rng):
    """Upper-triangular pixels of one chromosome: per row Poisson(lam) contacts at offsets
    floor(10**U(0,max_log10))-1, clipped to the chromosome, de-duplicated, count = 1+Poisson(0.3)."""
    n_r = rng.poisson(lam, nb)
    row = np.repeat(np.arange(nb, dtype=np.int64), n_r)
    d = np.floor(10.0 ** rng.uniform(0.0, max_log10, row.shape[0])).astype(np.int64) - 1
    col = row + d
    keep = col < nb
    key = np.unique(row[keep] * nb + col[keep])
    row = key // nb
    col = key - row * nb
    cnt = (1 + rng.poisson(0.3, key.shape[0])).astype(np.int32)
    return row, col, cnt


def _chrom_job(args):
    nb, lam, max_log10, seed = args
    return _chrom_block(nb, lam, max_log10, np.random.Generator(np.random.PCG64(seed)))


def make_cooler(chromsizes, binsize=10_000, lam=120.0, max_log10=3.5, nan_frac=0.02, seed=1000,
                trans_nnz=0, name="synthetic", parallel=False):
    """Build an :class:`ArrayCooler` with ``weight`` (NaN for masked bins) and ``cov_tot_raw`` / ``cov_cis_raw``.

    chromsizes: mapping name -> length (bp).  Per-chromosome RNG = PCG64(seed + chrom index).
    trans_nnz > 0 adds that many uniformly placed inter-chromosomal pixels (chrom i < chrom j).
    parallel=True generates chromosomes in forked worker processes (same result).
    """
    names = list(chromsizes)
    nb = np.array([-(-int(chromsizes[c]) // binsize) for c in names], dtype=np.int64)
    off = np.concatenate([[0], np.cumsum(nb)])
    nbins = int(off[-1])
    rows, cols, cnts = [], [], []
    jobs = [(int(nb[i]), lam, max_log10, seed + i) for i in range(len(names))]
    if parallel and len(jobs) > 1:
        # one worker per chromosome (fork: call this before any GPU runtime is initialised in the process)
        import multiprocessing as mp
        import os
        with mp.get_context("fork").Pool(min(len(jobs), max(1, (os.cpu_count() or 2) // 2))) as pool:
            blocks = pool.map(_chrom_job, jobs, chunksize=1)
    else:
        blocks = [_chrom_job(j) for j in jobs]
    for i, (r, c, k) in enumerate(blocks):
        rows.append(r + off[i]); cols.append(c + off[i]); cnts.append(k)
    if trans_nnz > 0:
        rng = np.random.Generator(np.random.PCG64(seed + 10_000))
        a = rng.integers(0, nbins, trans_nnz)
        b = rng.integers(0, nbins, trans_nnz)
        lo, hi = np.minimum(a, b), np.maximum(a, b)
        chrom_lo = np.searchsorted(off, lo, side="right")
        chrom_hi = np.searchsorted(off, hi, side="right")
        m = chrom_lo != chrom_hi
        key = np.unique(lo[m] * nbins + hi[m])
        rows.append(key // nbins); cols.append(key % nbins)
        cnts.append(np.ones(key.shape[0], np.int32))
    row = np.concatenate(rows); col = np.concatenate(cols); cnt = np.concatenate(cnts)
    if trans_nnz > 0:
        order = np.lexsort((col, row))
        row, col, cnt = row[order], col[order], cnt[order]
    bin1_offset = np.zeros(nbins + 1, np.int64)
    np.cumsum(np.bincount(row, minlength=nbins), out=bin1_offset[1:])
    rng = np.random.Generator(np.random.PCG64(seed + 20_000))
    weight = rng.uniform(0.5, 1.5, nbins) / np.sqrt(max(lam, 1.0))
    weight[rng.random(nbins) < nan_frac] = np.nan
    # raw marginals (each pixel counted on both of its bins, the main diagonal once per side)
    chrom_of = np.searchsorted(off, np.arange(nbins), side="right") - 1
    w64 = cnt.astype(np.float64)
    cov_tot = np.bincount(row, weights=w64, minlength=nbins) + np.bincount(col, weights=w64, minlength=nbins)
    cis = chrom_of[row] == chrom_of[col]
    cov_cis = (np.bincount(row[cis], weights=w64[cis], minlength=nbins)
               + np.bincount(col[cis], weights=w64[cis], minlength=nbins))
    return ArrayCooler(
        pd.Series({c: int(chromsizes[c]) for c in names}), binsize, bin1_offset, col.astype(np.int32), cnt,
        bins={"weight": weight, "cov_tot_raw": cov_tot, "cov_cis_raw": cov_cis}, filename=f"{name}.cool",
    )


def cis_expected(clr, weight_name="weight", value_col="balanced.avg"):
    """Per-chromosome by-diagonal mean of balanced values (cooltools ``expected_cis`` layout:
    region1, region2, dist, n_valid, balanced.sum, balanced.avg), computed from the pixel table."""
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()[weight_name][:].values
    row = np.repeat(np.arange(clr.nbins, dtype=np.int64), np.diff(indptr))
    val = cnt * w[row] * w[col]
    out = []
    for i, c in enumerate(clr.chromnames):
        lo, hi = int(clr.chrom_offset[i]), int(clr.chrom_offset[i + 1])
        n = hi - lo
        m = (row >= lo) & (row < hi) & (col < hi)
        d = (col[m] - row[m]).astype(np.int64)
        v = val[m]
        ok = np.isfinite(v)
        bal_sum = np.bincount(d[ok], weights=v[ok], minlength=n)[:n]
        good = ~np.isnan(w[lo:hi])
        # number of valid (both bins unmasked) cells per diagonal
        n_valid = np.array([np.count_nonzero(good[: n - k] & good[k:]) for k in range(n)]) if n <= 2048 else \
            _n_valid_fft(good)
        with np.errstate(divide="ignore", invalid="ignore"):
            avg = bal_sum / n_valid
        out.append(pd.DataFrame({"region1": c, "region2": c, "dist": np.arange(n), "n_valid": n_valid,
                                 "balanced.sum": bal_sum, value_col: avg}))
    return pd.concat(out, ignore_index=True)


def _n_valid_fft(good):
    """autocorrelation of the valid-bin indicator = number of valid cells per diagonal."""
    n = good.shape[0]
    f = np.fft.rfft(good.astype(np.float64), 2 * n)
    ac = np.fft.irfft(f * np.conj(f), 2 * n)[:n]
    return np.rint(ac).astype(np.int64)


def random_cis_pairs(clr, n_pairs, min_sep=230_000, max_sep=5_000_000, seed=42, strands=False):
    """BEDPE-style cis pairs: chromosome ~ length, anchor 1 uniform, separation log-uniform, 1-bin anchors."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sizes = clr.chromsizes.values.astype(np.float64)
    chrom_i = rng.choice(len(sizes), n_pairs, p=sizes / sizes.sum())
    sep = np.exp(rng.uniform(np.log(min_sep), np.log(max_sep), n_pairs)).astype(np.int64)
    res = clr.binsize
    length = clr.chromsizes.values[chrom_i]
    start1 = (rng.random(n_pairs) * np.maximum(length - sep - 2 * res, 1)).astype(np.int64) // res * res
    start2 = (start1 + sep) // res * res
    names = np.array(clr.chromnames, dtype=object)[chrom_i]
    df = pd.DataFrame({"chrom1": names, "start1": start1, "end1": start1 + res,
                       "chrom2": names, "start2": start2, "end2": start2 + res})
    if strands:
        df["strand1"] = rng.choice(np.array(["+", "-"], dtype=object), n_pairs)
        df["strand2"] = rng.choice(np.array(["+", "-"], dtype=object), n_pairs)
    return df


def random_trans_pairs(clr, n_pairs, seed=43):
    """BEDPE-style inter-chromosomal pairs with chrom1 before chrom2 in table order."""
    rng = np.random.Generator(np.random.PCG64(seed))
    nchr = len(clr.chromnames)
    a = rng.integers(0, nchr, n_pairs)
    b = rng.integers(0, nchr - 1, n_pairs)
    b = np.where(b >= a, b + 1, b)
    c1, c2 = np.minimum(a, b), np.maximum(a, b)
    res = clr.binsize
    L = clr.chromsizes.values
    s1 = (rng.random(n_pairs) * (L[c1] - res)).astype(np.int64) // res * res
    s2 = (rng.random(n_pairs) * (L[c2] - res)).astype(np.int64) // res * res
    names = np.array(clr.chromnames, dtype=object)
    return pd.DataFrame({"chrom1": names[c1], "start1": s1, "end1": s1 + res,
                         "chrom2": names[c2], "start2": s2, "end2": s2 + res})


def patched_cooler(clr, patch):
    """A copy of the in-memory cooler `clr` with some bins columns overwritten: patch = {column: {value_name: [bins]}},
    value_name one of "nan", "inf", "-inf", "zero", plus {"drop": [columns]} (JSON-friendly: golden scenarios store the patch
    in their meta)."""
    from coolpuppy_amd.cooler_lite import ArrayCooler
    vals = {"nan": np.nan, "inf": np.inf, "-inf": -np.inf, "zero": 0.0}
    cols = {}
    for c in ("weight", "cov_tot_raw", "cov_cis_raw"):
        try:
            cols[c] = np.array(clr.bins()[c][:].values, dtype=np.float64)
        except KeyError:
            pass
    for name in (patch or {}).get("drop", []):          # {"drop": [columns]}: a cooler that lacks them
        cols.pop(name, None)
    count = clr.count
    for col, edits in (patch or {}).items():
        if col == "drop":
            continue
        if col == "count_float_seed":                    # {"count_float_seed": s}: a FLOAT pixels/count column — every count times a
            # factor in [0.25, 1.75) drawn from PCG64(s): cooler allows float counts and coolpuppy multiplies them through
            count = clr.count.astype(np.float64) * (0.25 + 1.5 * np.random.Generator(np.random.PCG64(int(edits))).random(clr.count.shape[0]))
            continue
        for what, where in edits.items():
            cols[col][np.asarray(where, dtype=np.int64)] = vals[what]
    return ArrayCooler(clr.chromsizes, clr.binsize, clr.bin1_offset, clr.bin2_id, count, bins=cols, filename=clr.filename)
