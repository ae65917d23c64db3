"""The staged kernels at real-data scale: limits that used to drop a call — silently — onto the per-window
kernels.  get_data has no such cliffs (coolpuppy/coolpup.py:1024-1057): a deep map, many groups or a coverage-normalised pile-up
are the same loop to the reference.

  * a pixel table of more than 2^30 pixels (round 3: `nnz + 64 < 2^30` or K1r, 10x slower): four copies of the human-scale
    synthetic table side by side as one genome of 92 chromosomes, 1.34e9 pixels; windows on every copy — the last ones sit
    beyond pixel 2^30 — through K1q (band staging AND index staging, whose pixel positions are 64-bit now) and K1w;
  * more than 64 tiles (round 3: `T <= 64`): 100 and 300 groups on the staged kernel (records are numbered tile + workgroup now);
  * coverage vectors beside the lean staged kernel (round 3: COV forced the 8-wave geometry).
Every comparison is against the C oracle (bit-exact integers, 1e-9 on sums).  GPU only.
"""
import os

import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu
RTOL = 1e-9


def _compare(got, want):
    np.testing.assert_array_equal(got["n"], want["n"])
    np.testing.assert_array_equal(got["num"], want["num"])
    np.testing.assert_allclose(got["sum"], want["sum"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(got["cov_start"], want["cov_start"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(got["cov_end"], want["cov_end"], rtol=RTOL, atol=0)


@pytest.fixture(scope="module")
def big_table(hip_lib):
    """(bin1_offset, bin2_id, count, weight, chrom_offset) of `copies` human-scale tables side by side."""
    copies = 4
    one = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=4200, seed=1000, name="hg38_10kb", parallel=True)
    indptr, col, cnt = one.pixel_table()
    w = one.bins()["weight"][:].values
    nb, nnz = one.nbins, int(indptr[-1])
    big_indptr = np.concatenate([indptr[:-1] + k * nnz for k in range(copies)] + [[copies * nnz]]).astype(np.int64)
    big_col = np.concatenate([col.astype(np.int32) + np.int32(k * nb) for k in range(copies)])
    big_cnt = np.tile(cnt.astype(np.int32), copies)
    big_w = np.tile(w, copies)
    co = np.concatenate([one.chrom_offset[:-1] + k * nb for k in range(copies)] + [[copies * nb]]).astype(np.int64)
    assert big_indptr[-1] > (1 << 30)
    return big_indptr, big_col, big_cnt, big_w, co, nb, nnz


def _windows(co, nb, n_per_copy, pad, rng, copies=4, nshifts=10):
    """ROI corners + shifted control copies on every copy of the genome, tile-grouped (ROI, control), stream-like order."""
    W = 2 * pad + 1
    r0s, c0s = [], []
    base_co = co[: len(co) // copies + 1]
    lens = np.diff(base_co)
    for k in range(copies):
        ch = rng.choice(len(lens), n_per_copy, p=lens / lens.sum())
        sep = np.exp(rng.uniform(np.log(23 + W), np.log(440), n_per_copy)).astype(np.int64)      # (inside a 512-column band with room for W)
        r = (rng.random(n_per_copy) * np.maximum(lens[ch] - sep - 2 * W - 330, 1)).astype(np.int64) + 110      # (shifts of +-100 bins stay inside)
        r0s.append(base_co[ch] + r + k * nb)
        c0s.append(base_co[ch] + r + sep + k * nb)
    r0, c0 = np.concatenate(r0s), np.concatenate(c0s)
    sh = [np.zeros(len(r0), np.int64)] + [rng.integers(10, 100, len(r0)) * rng.choice([-1, 1], len(r0)) for _ in range(nshifts)]
    R = np.concatenate([r0 + s for s in sh]).astype(np.int32)
    C = np.concatenate([c0 + s for s in sh]).astype(np.int32)
    tile_ptr = np.array([0, len(r0), len(R)], np.int64)
    return R, C, tile_ptr


def test_table_beyond_2_30_pixels_stays_on_the_staged_kernels(big_table, oracle_mod):
    from coolpuppy_amd.engine import PileupEngine
    po = oracle_mod
    indptr, col, cnt, w, co, nb, nnz = big_table
    rng = np.random.default_rng(12)
    nthr = max(1, min(os.cpu_count() or 1, 64))
    eng = PileupEngine(0)
    eng.load_pixels(indptr, col, cnt)
    assert eng.build_index(co)
    eng.load_bins(w, None)
    timings = {}
    for name, pad, variant, n_per_copy in (("K1q band", 10, 0, 40_000), ("K1q index", 10, 1 << 27, 40_000), ("K1w", 25, 0, 20_000)):
        r0, c0, tile_ptr = _windows(co, nb, n_per_copy, pad, rng)
        tile = (np.arange(len(r0)) >= tile_ptr[1]).astype(np.int32)
        assert int(indptr[r0.max()]) > (1 << 30)                       # windows whose pixels sit beyond position 2^30
        want = po.pileup_c_mt(indptr, col, cnt, w, None, None, r0, c0, None, tile, 2, pad, 2, 0, nthr)
        eng.set_tuning(0, variant)
        eng.set_profiling(3)
        eng.reset(2, pad)
        eng.accumulate(r0, c0, tile_ptr, ignore_diags=2, mode=0)
        got = eng.fetch()
        st = eng.stats()
        assert st["staged_regions"] > 0 and eng.last_kernel() in ("staged", "wide_fact"), (name, eng.last_kernel())
        _compare(got, want)
        timings[name] = round(st["k1_ms"], 3)
        eng.clear_stats()
    eng.close()
    print("K1 ms on the 1.34e9-pixel table:", timings)


@pytest.mark.parametrize("T", [100, 300])
def test_more_than_64_tiles_on_the_staged_kernel(hip_lib, oracle_mod, T):
    from coolpuppy_amd.engine import PileupEngine
    po = oracle_mod
    clr = synth.make_cooler({"chrA": 120_000_000, "chrB": 60_000_000}, lam=300, seed=21)
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    rng = np.random.default_rng(T)
    n, pad = 300_000, 10
    lo, hi = clr.extent("chrA")
    r0 = rng.integers(lo + 50, hi - 700, n)
    c0 = r0 + rng.integers(25, 600, n)
    tile = rng.integers(0, T, n).astype(np.int32)
    flip = (rng.random(n) < 0.3).astype(np.uint8)
    order = np.lexsort((flip, tile))
    r0, c0, tile, flip = r0[order].astype(np.int32), c0[order].astype(np.int32), tile[order], flip[order]
    tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=T))]).astype(np.int64)
    ff = tile_ptr[1:] - np.bincount(tile[flip.astype(bool)], minlength=T)
    want = po.pileup_c_mt(indptr, col, cnt, w, None, None, r0, c0, flip, tile, T, pad, 2, 0, max(1, min(os.cpu_count() or 1, 32)))
    eng = PileupEngine(0)
    eng.load_pixels(indptr, col, cnt)
    eng.build_index(clr.chrom_offset)
    eng.load_bins(w, None)
    eng.set_tuning(0, 8)
    for rep in range(2):
        eng.reset(T, pad)
        eng.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=2, mode=0)
        got = eng.fetch()
        assert eng.last_kernel() == "staged" and eng.stats()["staged_regions"] > 0
        _compare(got, want)
    eng.close()


@pytest.mark.parametrize("pad", [10, 25])
def test_coverage_vectors_beside_the_lean_staged_kernel(hip_lib, oracle_mod, pad):
    """PUP_MODE_COV on a call the staged kernels take: the pile-up runs on the lean kernel (band staging, 16 waves), the
    coverage vectors as a pass of their own — same vectors as the oracle's, flips and tile grouping included."""
    from coolpuppy_amd.engine import PileupEngine
    po = oracle_mod
    clr = synth.make_cooler({"chrA": 150_000_000, "chrB": 50_000_000}, lam=250, seed=5)
    indptr, col, cnt = clr.pixel_table()
    cov = clr.bins()["cov_tot_raw"][:].values.astype(np.float64).copy()
    cov[::97] = np.nan                                  # NaN coverage adds nothing (nansum)
    rng = np.random.default_rng(3)
    n, T, W = 450_000, 4, 2 * pad + 1
    lo, hi = clr.extent("chrA")
    r0 = rng.integers(lo + 10, hi - 900, n)
    c0 = r0 + W + 2 + rng.integers(0, 600, n)
    tile = rng.integers(0, T, n).astype(np.int32)
    flip = (rng.random(n) < 0.5).astype(np.uint8)
    order = np.lexsort((flip, tile))
    r0, c0, tile, flip = r0[order].astype(np.int32), c0[order].astype(np.int32), tile[order], flip[order]
    tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=T))]).astype(np.int64)
    ff = tile_ptr[1:] - np.bincount(tile[flip.astype(bool)], minlength=T)
    want = po.pileup_c_mt(indptr, col, cnt, None, cov, None, r0, c0, flip, tile, T, pad, 0, po.MODE_COV, max(1, min(os.cpu_count() or 1, 32)))
    eng = PileupEngine(0)
    eng.load_pixels(indptr, col, cnt)
    eng.build_index(clr.chrom_offset)
    eng.load_bins(None, cov)
    eng.reset(T, pad)
    eng.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=0, mode=po.MODE_COV)
    got = eng.fetch()
    assert eng.last_kernel() in ("staged", "wide_fact", "wide"), eng.last_kernel()
    _compare(got, want)
    assert np.abs(got["cov_start"]).sum() > 0
    eng.close()


def test_trans_filter_on_a_genome_whose_exact_bitmap_does_not_fit(hip_lib, oracle_mod):
    """1.2e6 bins (a 2.5 kb human map): the exact presence bitmap of the inter-chromosomal kernel would take nbins^2 / 8 = 180 GB.
    Round 3 dropped to bisection per window row; now the filter holds a bit per 2^s columns, s the smallest that fits a quarter
    of the free memory — the kernel stays K1s and every window equals the oracle's.  The table (1e8 uniformly placed
    inter-chromosomal pixels) is made on the GPU with torch: sorting it on one host core would take minutes."""
    import torch
    from coolpuppy_amd.engine import PileupEngine
    sys_path_tools = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools")
    import importlib.util
    spec = importlib.util.spec_from_file_location("probe_trans_bins", os.path.join(sys_path_tools, "probe_trans_bins.py"))
    probe = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(probe)
    po = oracle_mod
    co, indptr, col, cnt, w = probe.make_table(1_200_000, 100_000_000, 5)
    torch.cuda.empty_cache()
    pad, n = 25, 100_000
    r0, c0 = probe.windows(co, n, pad, 9)
    tile = np.zeros(n, np.int32)
    want = po.pileup_c_mt(indptr, col, cnt, w, None, None, r0, c0, None, tile, 1, pad, -1, 0, max(1, min(os.cpu_count() or 1, 64)))
    assert want["sum"].sum() > 0
    eng = PileupEngine(0)
    eng.load_pixels(indptr, col, cnt)
    eng.build_index(co)
    eng.load_bins(w, None)
    for rep in range(2):
        eng.reset(1, pad)
        eng.accumulate(r0, c0, np.array([0, n], np.int64), ignore_diags=-1, mode=0)
        got = eng.fetch()
        assert eng.last_kernel() == "sparse"
        _compare(got, want)
    eng.close()
