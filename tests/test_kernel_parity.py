"""HIP engine (through the C ABI) vs the CPU oracle on identical seeded inputs — bit-exact integers,
1e-6 relative (observed ~1e-15) on float sums.  GPU only."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

RTOL = 1e-6   # north_star tolerance; sums differ from the oracle only by f64 addition order


@pytest.fixture(scope="module")
def small_clr():
    return synth.make_cooler({"chrA": 30_000_000, "chrB": 20_000_000, "chrC": 9_000_000}, lam=80, seed=11,
                             trans_nnz=40_000)


@pytest.fixture(scope="module", params=["indexed", "search", "staged"])
def engine(hip_lib, small_clr, request):
    """All ways of locating a row's pixels must give identical results: the rank-bitmap index (cis windows), the
    binary search (index ignored: variant 1), and the workgroup-staged kernel (variant 8 forces it — device block sort +
    LDS-staged regions — for every eligible call, however small)."""
    from coolpuppy_amd.engine import PileupEngine
    eng = PileupEngine(0)
    eng.load_pixels(*small_clr.pixel_table())
    assert eng.build_index(small_clr.chrom_offset)
    eng._variant = {"indexed": 0, "search": 1, "staged": 8}[request.param]
    eng.set_tuning(0, eng._variant)
    yield eng
    eng.close()


def _snippets(clr, n, pad, rng, lo, hi, near=True):
    W = 2 * pad + 1
    r0 = rng.integers(lo, hi - W - 400, n)
    if near:
        c0 = r0 + rng.integers(-8, 380, n)
    else:
        c0 = rng.integers(lo, hi - W, n)
    c0 = np.clip(c0, lo, hi - W)
    return r0.astype(np.int32), c0.astype(np.int32)


def _group(r0, c0, flip, tile, n_tiles):
    """Sort by (tile, flip); returns (..., tile_ptr) — use _flip_from() for the engine's flip argument."""
    key = tile.astype(np.int64) * 2 + (0 if flip is None else flip.astype(np.int64))
    order = np.argsort(key, kind="stable")
    tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=n_tiles))]).astype(np.int64)
    return r0[order], c0[order], (None if flip is None else flip[order]), tile[order], tile_ptr


def _flip_from(flip, tile, tile_ptr):
    if flip is None:
        return None
    return tile_ptr[1:] - np.bincount(tile[flip.astype(bool)], minlength=len(tile_ptr) - 1)


def _compare(got, want):
    np.testing.assert_array_equal(got["n"], want["n"])
    np.testing.assert_array_equal(got["num"], want["num"])
    np.testing.assert_allclose(got["sum"], want["sum"], rtol=RTOL, atol=0, equal_nan=True)
    np.testing.assert_allclose(got["cov_start"], want["cov_start"], rtol=RTOL, atol=0)
    np.testing.assert_allclose(got["cov_end"], want["cov_end"], rtol=RTOL, atol=0)


@pytest.mark.parametrize("pad", [10, 3, 25, 0, 40, 70, 16, 32, 127, 200, 300])
@pytest.mark.parametrize("scenario", ["balanced", "raw_cov", "ooe", "expected_only", "flip_groups"])
def test_cis_parity(engine, small_clr, oracle_mod, pad, scenario):
    """Every window width the reference can slice (coolpuppy/coolpup.py:1115-1121 has no limit): pads 127 / 200 / 300 are windows
    of 255 / 401 / 601 bins — the banded per-window kernel takes them in column panels, the wide staged kernel (K1w, forced by
    the 'staged' engine) as a grid of sub-windows, the expected-only pass by diagonals."""
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    cov = clr.bins()["cov_tot_raw"][:].values
    lo, hi = clr.extent("chrA")
    rng = np.random.default_rng(100 + pad)
    n, T = (1500 if pad <= 25 else (300 if pad <= 70 else 60)), 4
    r0, c0 = _snippets(clr, n, pad, rng, lo, hi)
    tile = rng.integers(0, T, n).astype(np.int32)
    flip = None
    mode, igd, weight, covv, expv = 0, 2, w, None, None
    if scenario == "raw_cov":
        mode, igd, weight, covv = po.MODE_COV, 0, None, cov
    elif scenario in ("ooe", "expected_only"):
        e = synth.cis_expected(clr)
        expv = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
        expv[5] = 0.0          # exp == 0 under pixels: inf in sum, excluded from num
        mode = po.MODE_OOE if scenario == "ooe" else po.MODE_EXPECTED
    elif scenario == "flip_groups":
        flip = (rng.random(n) < 0.5).astype(np.uint8)
    r0, c0, flip, tile, tile_ptr = _group(r0, c0, flip, tile, T)
    want = po.pileup_c(indptr, col, cnt, weight, covv, expv, r0, c0, flip, tile, T, pad, igd, mode)
    engine.load_bins(weight, covv)
    engine.set_expected(expv)
    engine.reset(T, pad)
    ff = _flip_from(flip, tile, tile_ptr)
    engine.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=igd, mode=mode)
    got = engine.fetch()
    _compare(got, want)
    if engine._variant == 8 and pad >= 16 and scenario in ("balanced", "ooe", "flip_groups"):
        assert engine.last_kernel().startswith("wide"), engine.last_kernel()     # K1w really ran
        if pad == 100:        # 201-bin windows = 16 sub-window groups: their keys must stay within the hand-written binning
            assert engine.last_prepass() == "binning", engine.last_prepass()
    if scenario == "expected_only" and engine._variant != 2:
        assert engine.last_kernel() == "expected_diag"
    # running accumulation: a second identical call doubles everything exactly for integers
    engine.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=igd, mode=mode)
    got2 = engine.fetch()
    np.testing.assert_array_equal(got2["num"], 2 * want["num"])
    np.testing.assert_array_equal(got2["n"], 2 * want["n"])


@pytest.mark.parametrize("pad", [16, 25, 31, 32, 50, 100, 150])
@pytest.mark.parametrize("scenario", ["balanced", "ooe", "ooe_scalar", "flip_two_tiles", "raw"])
def test_wide_staged_factorised(hip_lib, small_clr, oracle_mod, pad, scenario):
    """K1w with factorised counts: every window clears the masked diagonals (c0 - r0 >= igd + W - 1), so num comes from the
    per-batch mask words and the window loop touches values only.  Oracle parity and the same integers as the per-window
    banded kernel (variant 16 forbids the staged kernels)."""
    from coolpuppy_amd.engine import PileupEngine
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    lo, hi = clr.extent("chrA")
    W, igd = 2 * pad + 1, 2
    rng = np.random.default_rng(900 + pad)
    n, T = (4000 if pad <= 32 else 600), 2
    room = 1024 - (W + igd) - W                       # inside the 1024-column band: c0 + W - 1 - r0 < 1024
    assert room > 8
    r0 = rng.integers(lo, hi - 2 * W - igd - room - 2, n)
    c0 = r0 + W + igd + rng.integers(0, room, n)
    r0, c0 = r0.astype(np.int32), c0.astype(np.int32)
    tile = (rng.random(n) < 0.15).astype(np.int32)
    flip, mode, weight, expv = None, 0, w, None
    if scenario == "ooe":
        e = synth.cis_expected(clr)
        expv = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
        mode = po.MODE_OOE
    elif scenario == "ooe_scalar":
        expv, mode = np.array([2.5e-3]), po.MODE_OOE
    elif scenario == "flip_two_tiles":
        flip = (rng.random(n) < 0.5).astype(np.uint8)
    elif scenario == "raw":
        weight = None
    r0, c0, flip, tile, tile_ptr = _group(r0, c0, flip, tile, T)
    ff = _flip_from(flip, tile, tile_ptr)
    want = po.pileup_c(indptr, col, cnt, weight, None, expv, r0, c0, flip, tile, T, pad, igd, mode)
    got = {}
    for name, variant in (("wide", 8), ("band", 16)):
        eng = PileupEngine(0)
        eng.load_pixels(indptr, col, cnt)
        eng.build_index(clr.chrom_offset)
        eng.set_tuning(0, variant)
        eng.load_bins(weight, None)
        eng.set_expected(expv)
        eng.reset(T, pad)
        eng.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=igd, mode=mode)
        got[name] = eng.fetch()
        kern = eng.last_kernel()
        # bit-reproducible: a second engine pass gives the same doubles
        eng.reset(T, pad)
        eng.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=igd, mode=mode)
        again = eng.fetch()
        eng.close()
        _compare(got[name], want)
        np.testing.assert_array_equal(again["sum"], got[name]["sum"])
        assert kern == ("wide_fact" if name == "wide" else "band"), kern
    np.testing.assert_array_equal(got["wide"]["num"], got["band"]["num"])


@pytest.mark.parametrize("transpose", [False, True])
def test_trans_parity(engine, small_clr, oracle_mod, transpose):
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    pad, T, n = 25, 2, 800
    W = 2 * pad + 1
    rng = np.random.default_rng(5)
    loA, hiA = clr.extent("chrA")
    loB, hiB = clr.extent("chrB")
    ra = rng.integers(loA, hiA - W, n).astype(np.int32)
    cb = rng.integers(loB, hiB - W, n).astype(np.int32)
    tile = rng.integers(0, T, n).astype(np.int32)
    # reference frame: rows in chrB, cols in chrA when transpose (region1 after region2 in the table)
    r0, c0 = (ra, cb)            # what the engine gets: r0 always in the earlier region
    mode = po.MODE_OOE | (po.MODE_TRANSPOSE if transpose else 0)
    expv = np.array([3.25e-4])
    r0, c0, _, tile, tile_ptr = _group(r0, c0, None, tile, T)
    want = po.pileup_c(indptr, col, cnt, w, None, expv, r0, c0, None, tile, T, pad, -1, mode)
    engine.load_bins(w, None)
    engine.set_expected(expv)
    engine.reset(T, pad)
    engine.accumulate(r0, c0, tile_ptr, ignore_diags=-1, mode=mode)
    _compare(engine.fetch(), want)


def test_many_chunks_two_level_reduction(engine, small_clr, oracle_mod):
    """Small chunks force the slice level of the reduction tree; result must not depend on chunking."""
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    lo, hi = clr.extent("chrA")
    rng = np.random.default_rng(77)
    pad, T, n = 10, 2, 6000
    r0, c0 = _snippets(clr, n, pad, rng, lo, hi)
    tile = (np.arange(n) >= 1000).astype(np.int32)
    r0, c0, _, tile, tile_ptr = _group(r0, c0, None, tile, T)
    want = po.pileup_c(indptr, col, cnt, w, None, None, r0, c0, None, tile, T, pad, 2, 0)
    engine.load_bins(w, None)
    engine.set_expected(None)
    outs = []
    for chunk in (1, 7, 16, 0):
        engine.set_tuning(chunk, engine._variant)
        engine.reset(T, pad)
        engine.accumulate(r0, c0, tile_ptr, ignore_diags=2, mode=0)
        got = engine.fetch()
        _compare(got, want)
        outs.append(got["sum"].copy())
    engine.set_tuning(0, engine._variant)
    # determinism: same chunking twice -> bit-identical sums
    engine.reset(T, pad)
    engine.accumulate(r0, c0, tile_ptr, ignore_diags=2, mode=0)
    np.testing.assert_array_equal(engine.fetch()["sum"], outs[-1])


def test_errors_are_loud(engine, small_clr):
    from coolpuppy_amd.engine import PupError
    engine.load_bins(None, None)
    engine.set_expected(None)
    engine.reset(1, 10)
    with pytest.raises(PupError):          # OOE without expected
        engine.accumulate(np.zeros(1, np.int32), np.zeros(1, np.int32), np.array([0, 1]), mode=0x01)
    with pytest.raises(PupError):          # tile_ptr does not cover n
        engine.accumulate(np.zeros(4, np.int32), np.zeros(4, np.int32), np.array([0, 3]))
    # window leaves the table: detected on device, surfaced at sync/fetch
    engine.accumulate(np.array([small_clr.nbins - 5], np.int32), np.array([0], np.int32), np.array([0, 1]))
    with pytest.raises(PupError):
        engine.fetch()
    engine.reset(1, 10)
    engine.fetch()                         # error state cleared
    engine.reset(1, 400)                   # no width ceiling any more (round 4): 801 x 801 is served
    engine.accumulate(np.array([100], np.int32), np.array([900], np.int32), np.array([0, 1]))
    assert int(engine.fetch()["n"][0]) == 1
    engine.load_bins(None, None)
    engine.set_expected(np.ones(10))
    engine.reset(1, 70)                    # 141x141 expected-only pass: by diagonals, no LDS tile
    engine.accumulate(np.array([100], np.int32), np.array([100], np.int32), np.array([0, 1]), mode=0x02)
    got = engine.fetch()
    assert int(got["num"][0].sum()) == 141 + 2 * sum(141 - d for d in range(1, 10))      # ones on |d| < 10, NaN beyond
    engine.set_expected(None)


def test_index_block_boundaries(engine, small_clr, oracle_mod):
    """Windows starting at every column phase around the 448-column index block edges and 64-bit word edges."""
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    lo, hi = clr.extent("chrA")
    pad, T = 10, 1
    W = 2 * pad + 1
    starts = []
    for edge in (448, 896, 1344, 64, 128, 448 + 384):
        for d in range(-W - 2, 3):
            starts.append(edge + d)
    c0 = np.array(sorted(set(s for s in starts if s >= 0)), np.int64) + lo
    r0 = np.maximum(c0 - 30, lo)
    r0 = np.concatenate([r0, c0])            # also on-diagonal windows
    c0 = np.concatenate([c0, c0])
    ok = (c0 + W <= hi) & (r0 + W <= hi)
    r0, c0 = r0[ok].astype(np.int32), c0[ok].astype(np.int32)
    tile = np.zeros(len(r0), np.int32)
    tile_ptr = np.array([0, len(r0)], np.int64)
    want = po.pileup_c(indptr, col, cnt, w, None, None, r0, c0, None, tile, T, pad, 0, 0)
    engine.load_bins(w, None)
    engine.set_expected(None)
    engine.reset(T, pad)
    engine.accumulate(r0, c0, tile_ptr, ignore_diags=0, mode=0)
    _compare(engine.fetch(), want)
    # W = 64: the widest window the index serves (pad 31 is not expressible -> use per-snippet check at pad=31: W=63)
    pad = 31
    W = 2 * pad + 1
    r0b = np.arange(lo, lo + 700, 7, dtype=np.int32)
    c0b = (r0b + 385).astype(np.int32)
    want = po.pileup_c(indptr, col, cnt, w, None, None, r0b, c0b, None, np.zeros(len(r0b), np.int32), 1, pad, 2, 0)
    engine.reset(1, pad)
    engine.accumulate(r0b, c0b, np.array([0, len(r0b)], np.int64), ignore_diags=2, mode=0)
    _compare(engine.fetch(), want)


def test_cis_windows_near_chromosome_ends_and_trans_fallback(engine, small_clr, oracle_mod):
    """Windows touching the last bins of a chromosome (index tail block) and windows whose columns sit in
    another chromosome (must take the search path even when the index exists)."""
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    pad = 10
    W = 21
    loA, hiA = clr.extent("chrA")
    loB, hiB = clr.extent("chrB")
    engine.load_bins(w, None)
    engine.set_expected(None)
    # cis windows on the chromosome's last bins (diagonal mask 0: the engine reads the upper triangle only)
    r0 = np.array([hiA - W, hiA - W - 1, hiA - W - 3, loA, hiA - W - 40], np.int32)
    c0 = np.array([hiA - W, hiA - W, hiA - W, loA, hiA - W], np.int32)
    tile = np.zeros(len(r0), np.int32)
    want = po.pileup_c(indptr, col, cnt, w, None, None, r0, c0, None, tile, 1, pad, 0, 0)
    engine.reset(1, pad)
    engine.accumulate(r0, c0, np.array([0, len(r0)], np.int64), ignore_diags=0, mode=0)
    _compare(engine.fetch(), want)
    # trans windows (rows in chrA, columns in chrB), including ones hugging both chromosome edges
    r0 = np.array([hiA - W, hiA - W - 5, loA, hiA - 30], np.int32)
    c0 = np.array([loB, loB + 2, hiB - W, loB + 700], np.int32)
    tile = np.zeros(len(r0), np.int32)
    want = po.pileup_c(indptr, col, cnt, w, None, None, r0, c0, None, tile, 1, pad, -1, 0)
    engine.reset(1, pad)
    engine.accumulate(r0, c0, np.array([0, len(r0)], np.int64), ignore_diags=-1, mode=0)
    _compare(engine.fetch(), want)


@pytest.mark.parametrize("pad", [10, 25, 31])
@pytest.mark.parametrize("scenario", ["balanced", "raw_cov_flip", "ooe_exp_zero", "ooe_exp_nan"])
def test_trans_sparse_kernel_vs_oracle_and_dense(hip_lib, small_clr, oracle_mod, pad, scenario):
    """Inter-chromosomal windows take the sparse kernel (num from factorised bad-row / bad-column counts): oracle
    parity, and the same integers as the dense kernels (variant 32 switches the sparse one off)."""
    from coolpuppy_amd.engine import PileupEngine
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    cov = clr.bins()["cov_tot_raw"][:].values
    W, T, n = 2 * pad + 1, 3, 1200
    rng = np.random.default_rng(50 + pad)
    loA, hiA = clr.extent("chrA")
    loC, hiC = clr.extent("chrC")
    r0 = rng.integers(loA, hiA - W, n).astype(np.int32)
    c0 = rng.integers(loC, hiC - W, n).astype(np.int32)
    tile = rng.integers(0, T, n).astype(np.int32)
    flip, mode, weight, covv, expv = None, 0, w, None, None
    if scenario == "raw_cov_flip":
        mode, weight, covv = po.MODE_COV, None, cov
        flip = (rng.random(n) < 0.4).astype(np.uint8)
    elif scenario == "ooe_exp_zero":
        mode, expv = po.MODE_OOE, np.array([0.0])
    elif scenario == "ooe_exp_nan":
        mode, expv = po.MODE_OOE, np.array([np.nan])
    r0, c0, flip, tile, tile_ptr = _group(r0, c0, flip, tile, T)
    ff = _flip_from(flip, tile, tile_ptr)
    want = po.pileup_c(indptr, col, cnt, weight, covv, expv, r0, c0, flip, tile, T, pad, -1, mode)
    got = {}
    import os
    # (round 6: the sparse kernel keeps per-lane hit queues; tuning bit 21 = its first form, walking every window with the whole
    # wave — same adds in the same order: identical doubles; variant 1 = no presence filter at all: every row is a queued hit)
    for name, variant, shift in (("sparse", 0, None), ("sparse_fine_filter", 0, "2"), ("sparse_coarse_filter", 0, "7"),
                                 ("sparse_first_form", 1 << 21, None), ("sparse_no_filter", 1, None), ("sparse_first_form_no_filter", (1 << 21) | 1, None),
                                 ("dense", 32, None)):
        # (round 4: the presence bitmap is a filter of 2^k columns per bit — 16 by default, coarser when that does not fit; forced here.
        # round 6: 16-bit words, k >= 2)
        if shift is None:
            os.environ.pop("COOLPUPPY_AMD_TBITS_SHIFT", None)
        else:
            os.environ["COOLPUPPY_AMD_TBITS_SHIFT"] = shift
        eng = PileupEngine(0)
        eng.load_pixels(indptr, col, cnt)
        eng.build_index(clr.chrom_offset)
        eng.set_tuning(0, variant)
        eng.load_bins(weight, covv)
        eng.set_expected(expv)
        eng.reset(T, pad)
        eng.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=-1, mode=mode)
        got[name] = eng.fetch()
        if variant != 32:
            assert eng.last_kernel() == "sparse"
        eng.close()
        _compare(got[name], want)
    os.environ.pop("COOLPUPPY_AMD_TBITS_SHIFT", None)
    np.testing.assert_array_equal(got["sparse"]["num"], got["dense"]["num"])
    for k in ("sparse_fine_filter", "sparse_coarse_filter", "sparse_first_form", "sparse_no_filter", "sparse_first_form_no_filter"):
        np.testing.assert_array_equal(got[k]["num"], got["dense"]["num"])
        np.testing.assert_array_equal(got[k]["sum"], got["sparse"]["sum"])


@pytest.mark.parametrize("pad", [1, 7, 19, 31])
def test_trans_sparse_queue_kernel_chunk_shapes(hip_lib, small_clr, oracle_mod, pad):
    """The queued sparse kernel (round 6) over the shapes its batches can take: chunk sizes that leave one window, a full batch, a
    batch + 1 and several batches to a wave; filters of 4 / 16 / 512 columns per bit; coverage vectors, flips, a scalar expected;
    windows touching the table's last bins.  Against the C oracle, and bit for bit against the kernel's first form (tuning bit 21)."""
    import os
    from coolpuppy_amd.engine import PileupEngine
    po = oracle_mod
    clr = small_clr
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    cov = clr.bins()["cov_tot_raw"][:].values
    W, T, n = 2 * pad + 1, 2, 2500
    rng = np.random.default_rng(600 + pad)
    loA, hiA = clr.extent("chrA")
    loC, hiC = clr.extent("chrC")
    r0 = rng.integers(loA, hiA - W, n).astype(np.int32)
    c0 = rng.integers(loC, hiC - W + 1, n).astype(np.int32)
    c0[:40] = hiC - W                                       # windows ending on the table's last bin
    tile = rng.integers(0, T, n).astype(np.int32)
    flip = (rng.random(n) < 0.3).astype(np.uint8)
    r0, c0, flip, tile, tile_ptr = _group(r0, c0, flip, tile, T)
    ff = _flip_from(flip, tile, tile_ptr)
    try:
        for shift, chunk in (("2", 1), ("4", 64), ("4", 65), ("9", 333), (None, 0)):
            if shift is None:
                os.environ.pop("COOLPUPPY_AMD_TBITS_SHIFT", None)
            else:
                os.environ["COOLPUPPY_AMD_TBITS_SHIFT"] = shift
            for mode, weight, covv, expv in ((0, w, None, None), (po.MODE_COV, None, cov, None), (po.MODE_OOE, w, None, np.array([0.37]))):
                want = po.pileup_c(indptr, col, cnt, weight, covv, expv, r0, c0, flip, tile, T, pad, -1, mode)
                got = {}
                for form, variant in (("queued", 0), ("first", 1 << 21)):
                    eng = PileupEngine(0)
                    eng.load_pixels(indptr, col, cnt)
                    eng.build_index(clr.chrom_offset)
                    eng.set_tuning(chunk, variant)
                    eng.load_bins(weight, covv)
                    eng.set_expected(expv)
                    eng.reset(T, pad)
                    eng.accumulate(r0, c0, tile_ptr, flip_from=ff, ignore_diags=-1, mode=mode)
                    got[form] = eng.fetch()
                    assert eng.last_kernel() == "sparse"
                    eng.close()
                    _compare(got[form], want)
                for k in ("sum", "num", "cov_start", "cov_end"):
                    np.testing.assert_array_equal(got["queued"][k], got["first"][k], err_msg=f"{k} shift {shift} chunk {chunk} mode {mode}")
    finally:
        os.environ.pop("COOLPUPPY_AMD_TBITS_SHIFT", None)
