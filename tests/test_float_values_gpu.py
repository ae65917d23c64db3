"""Float pixel values (a float pixels/count column: cooler allows it, the reference multiplies it through, coolpup.py:1053-1057):
pup_load_pixel_values + the kernels on the balanced value table against the C oracle compiled for float64 values."""
import numpy as np
import pytest

import synth
from coolpuppy_amd import coolpup

pytestmark = pytest.mark.gpu


def _group(r0, c0, tile, T):
    o = np.argsort(tile, kind="stable")
    return r0[o], c0[o], tile[o], np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=T))]).astype(np.int64)


@pytest.mark.parametrize("pad,trans", [(4, False), (10, False), (20, False), (25, True)])
def test_float_values_against_the_float_oracle(hip_lib, oracle_mod, pad, trans):
    from coolpuppy_amd.engine import PileupEngine
    po = oracle_mod
    clr = synth.make_cooler({"chrA": 16_000_000, "chrB": 9_000_000}, lam=80, seed=5, trans_nnz=40_000 if trans else 0)
    indptr, col, cnt = clr.pixel_table()
    rng = np.random.default_rng(3)
    vals = cnt.astype(np.float64) * (0.25 + 1.5 * rng.random(len(cnt)))
    w = clr.bins()["weight"][:].values
    W = 2 * pad + 1
    n, T = 30_000, 3
    if trans:
        r0 = rng.integers(0, 1600 - W, n).astype(np.int32)
        c0 = rng.integers(1600, 2500 - W, n).astype(np.int32)
        igd = -1
    else:
        r0 = rng.integers(0, 1500, n).astype(np.int32)
        c0 = (r0 + rng.integers(0, 90, n)).astype(np.int32)
        igd = 2
    tile = rng.integers(0, T, n).astype(np.int32)
    r0, c0, tile, tp = _group(r0, c0, tile, T)
    e = synth.cis_expected(clr)
    expv = None if trans else e[e.region1 == "chrA"]["balanced.avg"].values.copy()
    with PileupEngine(0) as eng:
        eng.load_pixels(indptr, col, vals)
        assert eng.float_values
        eng.build_index(clr.chrom_offset)
        for weight, mode in ((w, 0), (None, 0)) + (() if trans else ((w, po.MODE_OOE),)):
            eng.load_bins(weight, None)
            eng.set_expected(expv if mode else None)
            for variant in (0, 8):                         # 8 = "force the staged kernel": a float table must decline it
                eng.set_tuning(0, variant)
                eng.reset(T, pad)
                eng.accumulate(r0, c0, tp, ignore_diags=igd, mode=mode)
                got = eng.fetch()
                assert eng.last_kernel() not in ("staged", "wide", "wide_fact"), eng.last_kernel()
                want = po.pileup_c(indptr, col, vals, weight, None, expv if mode else None, r0, c0, None, tile, T, pad, igd, mode)
                np.testing.assert_array_equal(got["n"], want["n"])
                np.testing.assert_array_equal(got["num"], want["num"])
                np.testing.assert_allclose(got["sum"], want["sum"], rtol=1e-12, atol=0)
        # coverage of a float table (round 6: f64 sums of the same streaming pass) against the numpy restatement
        for igd_cov in (0, 2):
            cis, tot = eng.coverage(clr.chrom_offset, ignore_diags=igd_cov)
            want_cis, want_tot = po.coverage_numpy(indptr, col, vals, clr.chrom_offset, igd_cov)
            np.testing.assert_allclose(cis, want_cis, rtol=1e-12, atol=0)
            np.testing.assert_allclose(tot, want_tot, rtol=1e-12, atol=0)
        # whole numbers in a float column ARE counts: the integer tables, every kernel
        eng.load_pixels(indptr, col, cnt.astype(np.float64))
        assert not eng.float_values
        cis, tot = eng.coverage(clr.chrom_offset, ignore_diags=0)
        np.testing.assert_array_equal(tot, clr.bins()["cov_tot_raw"][:].values)


def test_pileup_on_a_float_cooler_matches_the_integer_cooler_scaled(hip_lib):
    """pileup() end to end: a cooler whose counts are all multiplied by 0.5 (a float column) gives half the sums — ratios of ROI
    to control, and every integer, unchanged."""
    from coolpuppy_amd.cooler_lite import ArrayCooler
    clr = synth.make_cooler({"chrA": 20_000_000, "chrB": 12_000_000}, lam=60, seed=8)
    half = ArrayCooler(clr.chromsizes, clr.binsize, clr.bin1_offset, clr.bin2_id, clr.count * 0.5,
                       bins={"weight": clr.bins()["weight"][:].values}, filename="half.cool")
    pairs = synth.random_cis_pairs(clr, 4000, min_sep=230_000, max_sep=2_000_000, seed=2)
    kw = dict(features_format="bedpe", flank=100_000, nshifts=2, seed=1)
    a = coolpup.pileup(clr, pairs, **kw)
    b = coolpup.pileup(half, pairs, **kw)
    np.testing.assert_array_equal(a["num"].iloc[0], b["num"].iloc[0])
    assert int(a["n"].iloc[0]) == int(b["n"].iloc[0])
    np.testing.assert_allclose(a["data"].iloc[0], b["data"].iloc[0], rtol=1e-12, equal_nan=True)


def test_stream_loading_a_count_table_forgets_the_float_values(hip_lib):
    """a context that took pup_load_pixel_values and is then reloaded through pup_load_pixels_stream is a context of
    counts again — staged kernels allowed, coverage served, nothing read from the old value buffer (which was sized for the
    old table).  Compared with a fresh context on the same table."""
    from coolpuppy_amd.engine import PileupEngine
    small = synth.make_cooler({"chrA": 6_000_000}, lam=40, seed=2)
    big = synth.make_cooler({"chrA": 16_000_000, "chrB": 9_000_000}, lam=80, seed=5)
    si, sc, sn = small.pixel_table()
    indptr, col, cnt = big.pixel_table()
    w = big.bins()["weight"][:].values
    rng = np.random.default_rng(11)
    n, pad = 40_000, 10
    r0 = rng.integers(0, 1450, n).astype(np.int32)                    # (chrA: 1600 bins — every window inside it: the staged kernels' condition)
    c0 = (r0 + rng.integers(0, 90, n)).astype(np.int32)
    tp = np.array([0, n], np.int64)

    def fill(first, m, colv, cntv):
        colv[:] = col[first:first + m]
        cntv[:] = cnt[first:first + m]

    def run(eng, variant):
        eng.build_index(big.chrom_offset)
        eng.load_bins(w, None)
        eng.set_tuning(0, variant)
        eng.reset(1, pad)
        eng.accumulate(r0, c0, tp, ignore_diags=2, mode=0)
        return eng.fetch(), eng.last_kernel()

    with PileupEngine(0) as fresh:
        fresh.load_pixels(indptr, col, cnt)
        want, _ = run(fresh, 0)
        want_cov = fresh.coverage(big.chrom_offset, ignore_diags=0)
    with PileupEngine(0) as eng:
        eng.load_pixels(si, sc, sn.astype(np.float64) * 0.37)          # a SMALLER float table first
        assert eng.float_values
        eng.load_pixels_stream(indptr, len(col), col.dtype, fill, slab_pixels=1 << 16)
        assert not eng.float_values
        for variant in (0, 8):
            got, kern = run(eng, variant)
            if variant == 8:
                assert kern == "staged", kern                          # the forced staged kernel runs again on a table of counts
            np.testing.assert_array_equal(got["n"], want["n"])
            np.testing.assert_array_equal(got["num"], want["num"])
            np.testing.assert_allclose(got["sum"], want["sum"], rtol=1e-12, atol=0)
        cov = eng.coverage(big.chrom_offset, ignore_diags=0)           # PUP_ENOTSUP while the flag was stale
        np.testing.assert_array_equal(cov[0], want_cov[0])
        np.testing.assert_array_equal(cov[1], want_cov[1])


def test_coverage_norm_on_a_float_cooler_without_stored_coverage(hip_lib):
    """coverage_norm on a float-count cooler that stores no cov_*_raw columns — the reference computes them itself
    (coolpup.py:955-963) and carries on; here K3 sums the float values.  Halving every count halves the coverage and the sums:
    the coverage-normalised ROI / control ratio is unchanged, the stored column is half the integer cooler's."""
    from coolpuppy_amd.cooler_lite import ArrayCooler
    clr = synth.make_cooler({"chrA": 20_000_000, "chrB": 12_000_000}, lam=60, seed=8)
    half = ArrayCooler(clr.chromsizes, clr.binsize, clr.bin1_offset, clr.bin2_id, clr.count * 0.5,
                       bins={"weight": clr.bins()["weight"][:].values}, filename="half_cov.cool")
    whole = ArrayCooler(clr.chromsizes, clr.binsize, clr.bin1_offset, clr.bin2_id, clr.count,
                        bins={"weight": clr.bins()["weight"][:].values}, filename="whole_cov.cool")
    pairs = synth.random_cis_pairs(clr, 4000, min_sep=230_000, max_sep=2_000_000, seed=2)
    kw = dict(features_format="bedpe", flank=100_000, nshifts=2, seed=1, clr_weight_name=None, coverage_norm=True)
    a = coolpup.pileup(whole, pairs, **kw)
    b = coolpup.pileup(half, pairs, **kw)
    np.testing.assert_allclose(half.bins()["cov_tot_raw"][:].values, 0.5 * whole.bins()["cov_tot_raw"][:].values, rtol=1e-12)
    np.testing.assert_array_equal(a["num"].iloc[0], b["num"].iloc[0])
    np.testing.assert_allclose(a["data"].iloc[0], b["data"].iloc[0], rtol=1e-9, equal_nan=True)
