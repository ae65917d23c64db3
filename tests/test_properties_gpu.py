"""GPU tests at sizes the CPU oracle cannot reach in seconds: size-independent properties of the pile-up.

Linearity over snippet sets, independence from how a row's pixels are located (rank-bitmap index vs binary
search) and from which kernel accumulates (register tile vs LDS tile), invariance to snippet order and chunking,
the flip (anti-transpose) and transpose identities, and agreement with the oracle on a strided sample."""
import numpy as np
import pytest

import synth

pytestmark = pytest.mark.gpu

PAD, W = 10, 21


@pytest.fixture(scope="module")
def big():
    clr = synth.make_cooler({c: synth.HG38[c] for c in ("chr20", "chr21", "chr22")}, lam=1500, seed=1000)
    rng = np.random.default_rng(123)
    n = 300_000
    lo = np.array([clr.extent(c)[0] for c in clr.chromnames])
    hi = np.array([clr.extent(c)[1] for c in clr.chromnames])
    k = rng.integers(0, 3, n)
    r0 = lo[k] + (rng.random(n) * (hi[k] - lo[k] - 600)).astype(np.int64)
    c0 = r0 + rng.integers(-5, 520, n)
    c0 = np.clip(c0, lo[k], hi[k] - W)
    order = np.lexsort((c0, r0))
    return clr, r0[order].astype(np.int32), c0[order].astype(np.int32)


@pytest.fixture(scope="module")
def eng(hip_lib, big):
    from coolpuppy_amd.engine import PileupEngine
    clr = big[0]
    e = PileupEngine(0)
    e.load_pixels(*clr.pixel_table())
    assert e.build_index(clr.chrom_offset)
    e.load_bins(clr.bins()["weight"][:].values, clr.bins()["cov_tot_raw"][:].values)
    yield e
    e.close()


def run(eng, r0, c0, variant=0, chunk=0, group=0, mode=0, igd=2, flip_all=False):
    eng.set_tuning(chunk, variant | (group << 8))
    eng.reset(1, PAD)
    n = len(r0)
    eng.accumulate(r0, c0, np.array([0, n]), flip_from=np.array([0]) if flip_all else None, ignore_diags=igd, mode=mode)
    out = eng.fetch()
    eng.set_tuning(0, 0)
    return out


def test_linearity_and_order_invariance(eng, big):
    _, r0, c0 = big
    full = run(eng, r0, c0)
    half = len(r0) // 3
    a, b = run(eng, r0[:half], c0[:half]), run(eng, r0[half:], c0[half:])
    np.testing.assert_array_equal(a["num"] + b["num"], full["num"])
    np.testing.assert_array_equal(a["n"] + b["n"], full["n"])
    np.testing.assert_allclose(a["sum"] + b["sum"], full["sum"], rtol=1e-11, atol=0)
    perm = np.random.default_rng(1).permutation(len(r0))
    shuf = run(eng, r0[perm], c0[perm])
    np.testing.assert_array_equal(shuf["num"], full["num"])
    np.testing.assert_allclose(shuf["sum"], full["sum"], rtol=1e-11, atol=0)


def test_index_vs_search_vs_lds_kernel(eng, big):
    _, r0, c0 = big
    ref = run(eng, r0, c0)
    for variant in (1, 2, 3):      # 1: binary search only, 2: LDS-tile kernel, 3: both
        other = run(eng, r0, c0, variant=variant)
        np.testing.assert_array_equal(other["num"], ref["num"])
        np.testing.assert_allclose(other["sum"], ref["sum"], rtol=1e-11, atol=0)
    for chunk, group in ((16, 4), (1000, 1), (77, 512)):
        other = run(eng, r0, c0, chunk=chunk, group=group)
        np.testing.assert_array_equal(other["num"], ref["num"])
        np.testing.assert_allclose(other["sum"], ref["sum"], rtol=1e-11, atol=0)
    again = run(eng, r0, c0)
    np.testing.assert_array_equal(again["sum"], ref["sum"])        # same configuration -> bit-identical


def test_flip_is_antitranspose(eng, big):
    _, r0, c0 = big
    ref = run(eng, r0, c0, mode=0x04)
    fl = run(eng, r0, c0, mode=0x04, flip_all=True)
    np.testing.assert_array_equal(fl["num"][0], np.rot90(np.flipud(ref["num"][0])))
    np.testing.assert_array_equal(fl["sum"][0], np.rot90(np.flipud(ref["sum"][0])))
    np.testing.assert_array_equal(fl["cov_start"], ref["cov_start"])   # the reference flips data only


def test_oracle_on_strided_sample(eng, big, oracle_mod):
    clr, r0, c0 = big
    idx = np.arange(0, len(r0), 97)
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    cov = clr.bins()["cov_tot_raw"][:].values
    want = oracle_mod.pileup_c(indptr, col, cnt, w, cov, None, r0[idx], c0[idx], None,
                               np.zeros(len(idx), np.int32), 1, PAD, 2, 0x04)
    got = run(eng, r0[idx], c0[idx], mode=0x04)
    np.testing.assert_array_equal(got["num"], want["num"])
    np.testing.assert_allclose(got["sum"], want["sum"], rtol=1e-6, atol=0)
    np.testing.assert_allclose(got["cov_start"], want["cov_start"], rtol=1e-12)
    np.testing.assert_allclose(got["cov_end"], want["cov_end"], rtol=1e-12)


def test_block_staged_kernel_takes_over_large_overlapping_calls(hip_lib):
    """>= 1e6 overlapping cis windows: the engine sorts them by block on the device and piles the dense tile up from
    LDS-staged regions (K1q), the sparse tile with the plain kernel; same integers, same sums up to addition order.
    Pre-blocked input (PileupEngine.block_order) takes the same path without the sort."""
    import synth
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chrA": 40_000_000, "chrB": 25_000_000}, lam=120, seed=5)
    rng = np.random.default_rng(9)
    n_dense, n_sparse, pad = 1_300_000, 30_000, 10
    W = 2 * pad + 1
    lo, hi = clr.extent("chrA")
    lo2, hi2 = clr.extent("chrB")
    r0 = np.concatenate([rng.integers(lo, hi - W - 300, n_dense // 2), rng.integers(lo2, hi2 - W - 300, n_dense - n_dense // 2),
                         rng.integers(lo, hi - W - 300, n_sparse)])
    c0 = r0 + rng.integers(0, 280, len(r0))
    tile_ptr = np.array([0, n_dense, n_dense + n_sparse], np.int64)
    r0, c0 = r0.astype(np.int32), c0.astype(np.int32)
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    eng.load_bins(clr.bins()["weight"][:].values, clr.bins()["cov_tot_raw"][:].values)
    res = {}
    for name, variant in (("plain", 16), ("auto", 0)):
        eng.set_tuning(0, variant)
        eng.reset(2, pad)
        eng.accumulate(r0, c0, tile_ptr, ignore_diags=2, mode=0x04)
        res[name] = (eng.fetch(), eng.stats()["staged_regions"])
    assert res["plain"][1] == 0 and res["auto"][1] > 0
    # pre-blocked input: no sort, same kernel
    tile = np.repeat([0, 1], [n_dense, n_sparse])
    o = PileupEngine.block_order(r0, c0, clr.chrom_offset, tile=tile, pad=pad)
    eng.reset(2, pad)
    eng.accumulate(r0[o], c0[o], tile_ptr, ignore_diags=2, mode=0x04)
    res["blocked"] = (eng.fetch(), eng.stats()["staged_regions"])
    assert res["blocked"][1] > 0
    for name in ("auto", "blocked"):
        got, want = res[name][0], res["plain"][0]
        np.testing.assert_array_equal(got["n"], want["n"])
        np.testing.assert_array_equal(got["num"], want["num"])
        for k in ("sum", "cov_start", "cov_end"):
            np.testing.assert_allclose(got[k], want[k], rtol=1e-11, atol=0)
    eng.close()


@pytest.mark.parametrize("pad", list(range(1, 16)))
def test_block_staged_kernel_equals_plain_kernel_for_every_width(hip_lib, pad):
    """All 15 x 2 instantiations of the workgroup-staged kernel (W = 3 .. 31, plain / OOE): forced on, against the plain
    register-tile kernel on the same random inputs — several chromosomes, windows touching chromosome starts and ends,
    windows below the diagonal, flips, three tiles, expected with zeros / NaN, raw counts with coverage."""
    import synth
    from coolpuppy_amd.engine import MODE_COV, MODE_OOE, PileupEngine
    clr = synth.make_cooler({"chrA": 12_000_000, "chrB": 7_000_000, "chrC": 3_000_000}, lam=60, seed=21)
    W, T, n = 2 * pad + 1, 3, 6000
    rng = np.random.default_rng(1000 + pad)
    r0l, c0l = [], []
    for ch in clr.chromnames:
        lo, hi = clr.extent(ch)
        m = n // 3
        r = rng.integers(lo, hi - W + 1, m)
        c = np.clip(r + rng.integers(-6, 120, m), lo, hi - W)
        r[:8] = lo; c[:8] = lo + np.arange(8)                       # chromosome start
        r[8:16] = hi - W - np.arange(8); c[8:16] = hi - W           # chromosome end
        r0l.append(r); c0l.append(c)
    r0 = np.concatenate(r0l).astype(np.int32); c0 = np.concatenate(c0l).astype(np.int32)
    tile = rng.integers(0, T, len(r0)).astype(np.int32)
    flip = (rng.random(len(r0)) < 0.3)
    key = tile.astype(np.int64) * 2 + flip
    o = np.argsort(key, kind="stable")
    r0, c0, tile, flip = r0[o], c0[o], tile[o], flip[o]
    tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=T))]).astype(np.int64)
    flip_from = tile_ptr[1:] - np.bincount(tile[flip], minlength=T)
    w = clr.bins()["weight"][:].values
    cov = clr.bins()["cov_tot_raw"][:].values
    e = synth.cis_expected(clr)
    expv = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
    expv[3] = 0.0; expv[7] = np.nan
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    for weight, covv, mode, igd in ((w, None, 0, 2), (w, None, MODE_OOE, 2), (None, cov, MODE_COV, 0)):
        eng.load_bins(weight, covv)
        eng.set_expected(expv if mode & MODE_OOE else None)
        res = {}
        # staged: the workgroup-staged kernel with 8 waves (default) and 4 waves (+128), and the one-wave kernel (+64)
        for name, variant in (("plain", 16), ("staged", 8), ("staged4", 8 | 128), ("staged1", 8 | 64)):
            eng.set_tuning(0, variant)
            eng.reset(T, pad)
            eng.accumulate(r0, c0, tile_ptr, flip_from=flip_from, ignore_diags=igd, mode=mode)
            res[name] = (eng.fetch(), eng.stats()["staged_regions"])
        assert res["plain"][1] == 0
        for name in ("staged", "staged4", "staged1"):
            assert res[name][1] > 0
            for k in ("n", "num"):
                np.testing.assert_array_equal(res[name][0][k], res["plain"][0][k])
            for k in ("sum", "cov_start", "cov_end"):
                np.testing.assert_allclose(res[name][0][k], res["plain"][0][k], rtol=1e-11, atol=0, equal_nan=True)
    eng.close()


@pytest.mark.parametrize("pad,T", [(10, 10), (10, 42), (3, 4), (7, 16), (10, 8)])
def test_staged_kernel_sets_of_tile_pairs(hip_lib, pad, T):
    """Grouped pile-ups (T/2 groups, each a ROI tile t and its control tile t + T/2): four pairs share a pass of the staged
    kernel — one staging of a region serves eight tiles, each piled up by its own team of waves (key digit = slot, team
    table per unit, partial last set, empty tiles, tiles of very different sizes, flips) — against the plain register-tile
    kernel and against the same call with the pairs piled up one by one (tuning bit 28); windows far from the diagonal
    (factorised counts, 16 waves) and near it (per-cell validity, 8 waves); repeated: bit-identical."""
    import synth
    from coolpuppy_amd.engine import MODE_OOE, PileupEngine
    clr = synth.make_cooler({"chrA": 30_000_000, "chrB": 9_000_000}, lam=50, seed=33)
    W = 2 * pad + 1
    H = T // 2
    rng = np.random.default_rng(900 + pad + T)
    n = 40_000
    w = clr.bins()["weight"][:].values
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    eng.load_bins(w, None)
    for label, near in (("far", False), ("near", True)):
        r0l, c0l = [], []
        for ch in clr.chromnames:
            lo, hi = clr.extent(ch)
            m = n // 2
            r = rng.integers(lo, hi - W + 1, m)
            c = np.clip(r + (rng.integers(-4, 300, m) if near else rng.integers(W + 2, 300, m)), lo, hi - W)
            r0l.append(r); c0l.append(c)
        r0 = np.concatenate(r0l).astype(np.int32); c0 = np.concatenate(c0l).astype(np.int32)
        if not near:
            keep = c0 - r0 >= W + 2                              # (the clip at the chromosome end may have pulled some in)
            r0, c0 = r0[keep], c0[keep]
        # group g: ROI tile g (few windows), control tile H + g (ten times as many); group sizes differ, one group is empty
        share = rng.random(H) ** 2
        if H >= 3:
            share[1] = 0.0
        grp = rng.choice(H, size=len(r0), p=share / share.sum())
        tile = np.where(rng.random(len(r0)) < 1 / 11, grp, H + grp).astype(np.int32)
        flip = rng.random(len(r0)) < 0.25
        o = np.argsort(tile.astype(np.int64) * 2 + flip, kind="stable")
        r0, c0, tile, flip = r0[o], c0[o], tile[o], flip[o]
        tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=T))]).astype(np.int64)
        flip_from = tile_ptr[1:] - np.bincount(tile[flip], minlength=T)
        e = synth.cis_expected(clr)
        clean = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
        clean[:2] = np.nan
        dirty = clean.copy(); dirty[40] = 0.0
        # plain; observed over expected with a usable expected (factorised counts) and with a zero on a kept diagonal
        for what, mode, expv in (("plain", 0, None), ("ooe", MODE_OOE, clean), ("ooe dirty", MODE_OOE, dirty)):
            if what != "plain" and (T, pad) not in ((10, 10), (16, 7)):
                continue
            res = {}
            # (round 6: tuning bit 22 = the progressive staging experiment — no barrier between blocks, windows of a wave in row-bucket
            # order; instantiated for 21-bin windows: elsewhere the bit changes nothing)
            for name, variant in (("plain", 16), ("sets", 8), ("pairs", 8 | (1 << 28)), ("sets again", 8), ("sets sparse", 8 | (1 << 27)),
                                  ("sets progressive", 8 | (1 << 22)), ("pairs progressive", 8 | (1 << 28) | (1 << 22)),
                                  ("sets progressive again", 8 | (1 << 22))):
                eng.set_tuning(0, variant)
                eng.set_expected(expv)
                eng.reset(T, pad)
                eng.accumulate(r0, c0, tile_ptr, flip_from=flip_from, ignore_diags=2, mode=mode)
                res[name] = (eng.fetch(), eng.stats()["staged_regions"])
            assert res["plain"][1] == 0
            assert 0 < res["sets"][1] < res["pairs"][1] or H < 2, (res["sets"][1], res["pairs"][1])
            for name in ("sets", "pairs", "sets sparse", "sets progressive", "pairs progressive"):
                for k in ("n", "num"):
                    np.testing.assert_array_equal(res[name][0][k], res["plain"][0][k], err_msg=f"{label} {what} {name} {k}")
                np.testing.assert_allclose(res[name][0]["sum"], res["plain"][0]["sum"], rtol=1e-11, atol=0, equal_nan=True,
                                           err_msg=f"{label} {what} {name}")
            np.testing.assert_array_equal(res["sets again"][0]["sum"], res["sets"][0]["sum"])
            np.testing.assert_array_equal(res["sets progressive again"][0]["sum"], res["sets progressive"][0]["sum"])
    eng.close()


@pytest.mark.parametrize("pad", [3, 10])
def test_staged_kernel_observed_over_expected_with_factorised_counts(hip_lib, pad):
    """Observed over expected where every unusable diagonal of the expected is an ignored one: the staged kernel then counts
    `num` from row / column masks as it does without expected (16 waves, big regions).  Against the plain register-tile
    kernel: a clean expected vector and a per-region table (factorised), a vector with a NaN on a kept diagonal, windows
    reaching past the end of a short vector, and a scalar expected (each must fall back to per-cell validity or stay exact)."""
    import synth
    from coolpuppy_amd.engine import MODE_OOE, PileupEngine
    clr = synth.make_cooler({"chrA": 40_000_000, "chrB": 12_000_000}, lam=50, seed=35)
    W = 2 * pad + 1
    rng = np.random.default_rng(77 + pad)
    n = 30_000
    r0l, c0l = [], []
    for ch in clr.chromnames:
        lo, hi = clr.extent(ch)
        r = rng.integers(lo, hi - W - 400, n // 2)
        c = np.clip(r + rng.integers(W + 2, 380, n // 2), lo, hi - W)
        keep = c - r >= W + 2
        r0l.append(r[keep]); c0l.append(c[keep])
    r0 = np.concatenate(r0l).astype(np.int32); c0 = np.concatenate(c0l).astype(np.int32)
    m = len(r0)
    tile_ptr = np.array([0, m // 6, m], np.int64)
    w = clr.bins()["weight"][:].values
    e = synth.cis_expected(clr)
    vecA = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
    vecB = e[e.region1 == "chrB"]["balanced.avg"].values.copy()
    vecA[:2] = np.nan; vecB[:2] = 0.0                       # the ignored diagonals: unusable, as cooltools leaves them
    far = int(np.argmax(~(np.isfinite(vecA[2:]) & (vecA[2:] != 0)))) + 2 if not (np.isfinite(vecA[2:]) & (vecA[2:] != 0)).all() else len(vecA)
    assert far > 400, far                                   # usable on every diagonal the windows below reach
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    eng.load_bins(w, None)
    loA, hiA = clr.extent("chrA"); loB, hiB = clr.extent("chrB")
    dirty = vecA.copy(); dirty[9] = np.nan
    cases = {
        "clean vector": lambda: eng.set_expected(vecA),
        "clean table": lambda: eng.set_expected_table([loA, loB], [hiA, hiB], vectors=[vecA, vecB]),
        "NaN on a kept diagonal": lambda: eng.set_expected(dirty),
        "short vector": lambda: eng.set_expected(vecA[:200]),
        "scalar": lambda: eng.set_expected(np.array([2.5])),
    }
    for name, setter in cases.items():
        res = {}
        for label, variant in (("plain", 16), ("staged", 8), ("staged sparse", 8 | (1 << 27)), ("staged per-cell", 8 | 4)):
            eng.set_tuning(0, variant)
            setter()
            eng.reset(2, pad)
            eng.accumulate(r0, c0, tile_ptr, ignore_diags=2, mode=MODE_OOE)
            res[label] = (eng.fetch(), eng.stats()["staged_regions"])
        assert res["plain"][1] == 0 and res["staged"][1] > 0
        for label in ("staged", "staged sparse", "staged per-cell"):
            for k in ("n", "num"):
                np.testing.assert_array_equal(res[label][0][k], res["plain"][0][k], err_msg=f"{name} {label} {k}")
            np.testing.assert_allclose(res[label][0]["sum"], res["plain"][0]["sum"], rtol=1e-11, atol=0, equal_nan=True,
                                       err_msg=f"{name} {label}")
    eng.close()


@pytest.mark.parametrize("columns", [256, 512])
def test_staged_kernel_with_a_narrow_band(hip_lib, columns, monkeypatch):
    """Tables too large for the 1024-column band of counts get a 512- or 256-column one (here forced through
    COOLPUPPY_AMD_BAND_COLUMNS): calls whose windows stay inside it are staged from the band, calls with a window beyond it
    through the index — same results as the plain kernel either way."""
    import synth
    from coolpuppy_amd.engine import PileupEngine
    monkeypatch.setenv("COOLPUPPY_AMD_BAND_COLUMNS", str(columns))
    clr = synth.make_cooler({"chrA": 30_000_000, "chrB": 6_000_000}, lam=50, seed=39)
    pad, W = 10, 21
    rng = np.random.default_rng(columns)
    n = 30_000
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    eng.load_bins(clr.bins()["weight"][:].values, None)
    lo, hi = clr.extent("chrA")
    r0 = rng.integers(lo, hi - W - 1200, n).astype(np.int32)
    tile_ptr = np.array([0, n // 8, n], np.int64)
    for reach in (columns - W - 2, columns + 300):           # every window inside the band / some beyond it
        c0 = np.clip(r0 + rng.integers(W + 2, reach, n), lo, hi - W).astype(np.int32)
        res = {}
        for name, variant in (("plain", 16), ("staged", 8), ("staged, no band", 8 | (1 << 27))):
            eng.set_tuning(0, variant)
            eng.reset(2, pad)
            eng.accumulate(r0, c0, tile_ptr, ignore_diags=2)
            res[name] = (eng.fetch(), eng.stats()["staged_regions"])
        assert res["plain"][1] == 0 and res["staged"][1] > 0
        for name in ("staged", "staged, no band"):
            for k in ("n", "num"):
                np.testing.assert_array_equal(res[name][0][k], res["plain"][0][k], err_msg=f"{columns} {reach} {name} {k}")
            np.testing.assert_allclose(res[name][0]["sum"], res["plain"][0]["sum"], rtol=1e-11, atol=0, equal_nan=True)
    eng.close()


def test_staged_kernel_launched_on_the_previous_verdict(hip_lib):
    """A call shape seen before is launched on the previous call's verdict before this call's is known.  Calls of ONE shape
    whose verdicts differ — windows clear of the masked diagonals (factorised counts), windows touching them (per-cell
    validity), windows beyond the band of counts, a window the index does not cover (per-window kernels take the call) — in
    an order that makes every guess wrong at least once; every result against the plain kernel's."""
    import synth
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chrA": 40_000_000, "chrB": 10_000_000}, lam=50, seed=41)
    pad, W = 10, 21
    rng = np.random.default_rng(11)
    lo, hi = clr.extent("chrA")
    loB, hiB = clr.extent("chrB")
    n = 30_000
    r0 = rng.integers(lo, hi - W - 1500, n).astype(np.int32)
    tile_ptr = np.array([0, n // 6, n], np.int64)
    far = np.clip(r0 + rng.integers(W + 2, 380, n), lo, hi - W).astype(np.int32)
    near = np.clip(r0 + rng.integers(-6, 380, n), lo, hi - W).astype(np.int32)
    wide = far.copy(); wide[::97] = np.clip(r0[::97] + 1200, lo, hi - W)             # beyond the 1024-column band
    trans = far.copy(); trans[5] = loB + 100                                           # one inter-chromosomal window
    cases = {"far": far, "near": near, "wide": wide, "trans": trans}
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    eng.load_bins(clr.bins()["weight"][:].values, None)
    want = {}
    eng.set_tuning(0, 16)
    for name, c0 in cases.items():
        eng.reset(2, pad)
        eng.accumulate(r0, c0, tile_ptr, ignore_diags=2 if name != "trans" else 2)
        want[name] = eng.fetch()
    eng.set_tuning(0, 8)
    for name in ("far", "far", "near", "near", "far", "wide", "far", "trans", "far", "near", "trans", "wide", "wide", "far"):
        eng.reset(2, pad)
        eng.accumulate(r0, cases[name], tile_ptr, ignore_diags=2)
        got = eng.fetch()
        for k in ("n", "num"):
            np.testing.assert_array_equal(got[k], want[name][k], err_msg=f"{name} {k}")
        np.testing.assert_allclose(got["sum"], want[name]["sum"], rtol=1e-11, atol=0, equal_nan=True, err_msg=name)
    eng.close()


def test_block_columns_renumbered_when_a_window_leaves_the_band(hip_lib):
    """Block columns are numbered from the block row's own diagonal while every window lies inside the dense band (fewer key bits:
    grouped calls stay within the hand-written binning).  A FIRST call on a fresh engine with windows beyond the band — and windows
    below the diagonal — makes the engine redo its prepass with the plain numbering; the engine remembers that for the table, a new
    index forgets it.  Every result against the per-window kernels; grouped calls (eight tiles: tile sets) included."""
    import synth
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chrA": 40_000_000, "chrB": 10_000_000}, lam=50, seed=43)
    pad, W = 10, 21
    rng = np.random.default_rng(12)
    lo, hi = clr.extent("chrA")
    n = 40_000
    r0 = rng.integers(lo + 200, hi - W - 1600, n).astype(np.int32)
    inside = (r0 + rng.integers(W + 2, 900, n)).astype(np.int32)
    beyond = inside.copy(); beyond[::53] = r0[::53] + 1300                              # beyond the 1024-column band
    below = inside.copy(); below[::41] = r0[::41] - 150                                 # well below the diagonal
    for T in (2, 8):
        tile = np.sort(rng.integers(0, T, n)).astype(np.int64)
        tile_ptr = np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=T))]).astype(np.int64)
        eng = PileupEngine(0)
        eng.load_pixels(*clr.pixel_table())
        eng.build_index(clr.chrom_offset)
        eng.load_bins(clr.bins()["weight"][:].values, None)
        for order in (("beyond", "inside", "below"), ("inside", "below", "inside", "beyond", "inside")):
            eng.build_index(clr.chrom_offset)                 # (forgets what earlier calls taught the engine about this table)
            for name in order:
                c0 = {"inside": inside, "beyond": beyond, "below": below}[name]
                res = {}
                for variant in (16, 8):
                    eng.set_tuning(0, variant)
                    eng.reset(T, pad)
                    eng.accumulate(r0, c0, tile_ptr, ignore_diags=2)
                    res[variant] = (eng.fetch(), eng.stats()["staged_regions"], eng.last_kernel())
                assert res[8][1] > 0 and res[8][2] == "staged", (T, name, res[8][1:])
                for k in ("n", "num"):
                    np.testing.assert_array_equal(res[8][0][k], res[16][0][k], err_msg=f"{T} {name} {k}")
                np.testing.assert_allclose(res[8][0]["sum"], res[16][0]["sum"], rtol=1e-11, atol=0, equal_nan=True, err_msg=f"{T} {name}")
        eng.close()


@pytest.mark.parametrize("S", [11, 33, 99])
def test_rescaled_zoom_by_weights_equals_the_per_sample_zoom(hip_lib, S):
    """The rescaled pile-up's zoom as two sets of weights (separable bilinear interpolation + block mean) against the per-sample
    loop it replaced (tuning bit 20 switches the weights off; until round 6 the test set bit 11, which pup_set_tuning reads as a group size: both runs took the weights): windows smaller than the output (upsampling), equal, up to 9 x
    larger, rectangular, local (symmetrised) and plain, with masked bins inside, observed over expected; accumulated tiles and
    per-window emission.  Same NaN pattern and counts; sums within rounding of the different addition order."""
    import synth
    from coolpuppy_amd.engine import PileupEngine, MODE_OOE
    clr = synth.make_cooler({"chrA": 30_000_000, "chrB": 12_000_000}, lam=60, seed=71)
    w = clr.bins()["weight"][:].values
    rng = np.random.default_rng(S)
    lo, hi = clr.extent("chrA")
    n = 400
    hgt = rng.integers(3, 9 * S, n).astype(np.int32)
    hgt[:40] = S; hgt[40:60] = rng.integers(2, S, 20)
    wid = hgt.copy()
    wid[n // 2:] = np.maximum(2, (hgt[n // 2:] * rng.uniform(0.3, 1.5, n - n // 2)).astype(np.int32))
    r0 = rng.integers(lo + 5, hi - 9 * S - 1500, n).astype(np.int32)
    c0 = r0.copy()
    c0[n // 2:] += rng.integers(0, 300, n - n // 2).astype(np.int32)
    tile_ptr = np.array([0, n // 2, n], np.int64)
    e = synth.cis_expected(clr)
    expv = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
    pad = (S - 1) // 2
    for mode, igd in ((0x20, 2), (0, 0), (MODE_OOE | 0x20, 2)):
        res = {}
        for variant in (0, 1 << 20):                     # tuning bit 20: the zoom sample by sample instead of by separable weights
            eng = PileupEngine(0)
            eng.load_pixels(*clr.pixel_table())
            eng.build_index(clr.chrom_offset)
            eng.load_bins(w, None)
            eng.set_expected(expv if mode & MODE_OOE else None)
            eng.set_tuning(0, variant)
            eng.reset(2, pad)
            eng.accumulate_rescaled(r0, c0, hgt, wid, tile_ptr, ignore_diags=igd, mode=mode)
            acc = eng.fetch()
            snips = eng.extract(r0[:60], c0[:60], pad, height=hgt[:60], width=wid[:60], ignore_diags=igd, mode=mode)
            res[variant] = (acc, snips[0] if isinstance(snips, tuple) else snips)
            eng.close()
        a, b = res[0], res[1 << 20]
        np.testing.assert_array_equal(a[0]["num"], b[0]["num"])
        np.testing.assert_array_equal(a[0]["n"], b[0]["n"])
        np.testing.assert_allclose(a[0]["sum"], b[0]["sum"], rtol=1e-11, atol=0)
        np.testing.assert_array_equal(np.isnan(a[1]), np.isnan(b[1]))
        np.testing.assert_allclose(a[1], b[1], rtol=1e-12, atol=1e-300, equal_nan=True)
        assert a[0]["num"].sum() > 0 and np.isfinite(a[1]).any()


def test_rescaled_output_tiles_beyond_lds(hip_lib, oracle_mod):
    """rescale_size above ~115: the S x S output tile no longer fits LDS and lives in the workgroup's stretch of a global scratch
    buffer (the reference zooms to any odd size, coolpup.py:1193-1234; round 4 returned PUP_ENOTSUP).  Against the scipy
    restatement of _rescale_snip + zoom_array, windows above and below the output size, with and without expected."""
    import synth
    from coolpuppy_amd.engine import PileupEngine, MODE_OOE
    po = oracle_mod
    clr = synth.make_cooler({"chrA": 30_000_000, "chrB": 12_000_000}, lam=60, seed=71)
    indptr, col, cnt = clr.pixel_table()
    w = clr.bins()["weight"][:].values
    e = synth.cis_expected(clr)
    expv = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
    nb = indptr.shape[0] - 1
    big = po.symmetric_csr(indptr, col, cnt, w, 0, nb, 0, nb)
    rng = np.random.default_rng(4)
    lo, hi = clr.extent("chrA")
    for S in (131, 201):
        pad = (S - 1) // 2
        n = 24
        hgt = rng.integers(20, 2 * S, n).astype(np.int32)
        hgt[:4] = S
        wid = hgt.copy()
        r0 = rng.integers(lo + 5, hi - 2 * S - 600, n).astype(np.int32)
        c0 = r0.copy()
        tile = (np.arange(n) >= n // 2).astype(np.int32)
        tile_ptr = np.array([0, n // 2, n], np.int64)
        for mode, igd, ex in ((0x20, 2, None), (MODE_OOE | 0x20, 2, expv)):
            with PileupEngine(0) as eng:
                eng.load_pixels(indptr, col, cnt)
                eng.build_index(clr.chrom_offset)
                eng.load_bins(w, None)
                eng.set_expected(ex)
                eng.reset(2, pad)
                eng.accumulate_rescaled(r0, c0, hgt, wid, tile_ptr, ignore_diags=igd, mode=mode)
                got = eng.fetch()
                snips = eng.extract(r0[:3], c0[:3], pad, height=hgt[:3], width=wid[:3], ignore_diags=igd, mode=mode)
            want = po.pileup_rescaled(big, 0, 0, w, None, ex, r0, c0, hgt, wid, None, tile, 2, S, igd, mode)
            np.testing.assert_array_equal(got["n"], want["n"])
            np.testing.assert_array_equal(got["num"], want["num"])
            np.testing.assert_allclose(got["sum"], want["sum"], rtol=1e-9, atol=1e-300)
            one = po.windows_scipy(big, 0, 0, w, None, ex, r0[:3], c0[:3], pad, igd, mode, h=hgt[:3], w=wid[:3])[0]
            np.testing.assert_allclose(snips, one, rtol=1e-9, atol=1e-300, equal_nan=True)


def test_staged_kernel_with_an_empty_tile_of_a_pair(hip_lib):
    """A tile pair whose first or second tile has no window at all (a group without controls in this region, or the other way
    round): its team has no wave, its record stays invalid, the partner gets every wave."""
    import synth
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chrA": 30_000_000}, lam=50, seed=37)
    pad, W = 10, 21
    rng = np.random.default_rng(5)
    lo, hi = clr.extent("chrA")
    n = 25_000
    r0 = rng.integers(lo, hi - W - 400, n).astype(np.int32)
    c0 = np.clip(r0 + rng.integers(W + 2, 380, n), lo, hi - W).astype(np.int32)
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    eng.load_bins(clr.bins()["weight"][:].values, None)
    for tile_ptr in ([0, 0, n], [0, n, n], [0, 0, n // 3, n // 3, n], [0, n // 2, n // 2, n, n]):
        tile_ptr = np.array(tile_ptr, np.int64)
        T = len(tile_ptr) - 1
        res = {}
        for name, variant in (("plain", 16), ("staged", 8), ("staged sparse", 8 | (1 << 27)), ("pairs one by one", 8 | (1 << 28))):
            eng.set_tuning(0, variant)
            eng.reset(T, pad)
            eng.accumulate(r0, c0, tile_ptr, ignore_diags=2)
            res[name] = (eng.fetch(), eng.stats()["staged_regions"])
        assert res["plain"][1] == 0 and res["staged"][1] > 0
        for name in ("staged", "staged sparse", "pairs one by one"):
            for k in ("n", "num"):
                np.testing.assert_array_equal(res[name][0][k], res["plain"][0][k], err_msg=f"{tile_ptr} {name} {k}")
            np.testing.assert_allclose(res[name][0]["sum"], res["plain"][0]["sum"], rtol=1e-11, atol=0, equal_nan=True)
    eng.close()


@pytest.mark.parametrize("pad", [2, 10, 15])
def test_staged_kernel_many_workgroups(hip_lib, pad):
    """The workgroup-staged kernel with several workgroups sharing every CU (one block per workgroup, ~1500 of them):
    the configuration in which a register-allocation dependent fault of its hand-issued LDS reads showed (see
    lds_read_b64 in pup_kernels.hpp) while every single-workgroup-per-CU case passed.  Plain / OOE / coverage, paired
    and single tiles, windows near and far from the diagonal (per-cell and factorised validity), against the plain
    register-tile kernel; repeated, because the fault was timing dependent."""
    import synth
    from coolpuppy_amd.engine import MODE_COV, MODE_OOE, PileupEngine
    clr = synth.make_cooler({"chrA": 60_000_000, "chrB": 15_000_000}, lam=40, seed=31)
    W = 2 * pad + 1
    rng = np.random.default_rng(500 + pad)
    lo, hi = clr.extent("chrA")
    n = 12_000
    r0 = rng.integers(lo, hi - W - 400, n).astype(np.int32)
    w = clr.bins()["weight"][:].values
    cov = clr.bins()["cov_tot_raw"][:].values
    e = synth.cis_expected(clr)
    expv = e[e.region1 == "chrA"]["balanced.avg"].values.copy()
    expv[5] = np.nan
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.build_index(clr.chrom_offset)
    for label, off in (("near", rng.integers(-8, 380, n)), ("far", rng.integers(W + 2, 380, n))):
        c0 = np.clip(r0 + off, lo, hi - W).astype(np.int32)
        for T in (1, 2):
            tile_ptr = np.array([0, n], np.int64) if T == 1 else np.array([0, n // 5, n], np.int64)
            for weight, covv, mode, igd in ((w, None, 0, 2), (w, None, MODE_OOE, 2), (None, cov, MODE_COV, 1)):
                eng.load_bins(weight, covv)
                eng.set_expected(expv if mode & MODE_OOE else None)
                eng.set_tuning(0, 16)
                eng.reset(T, pad)
                eng.accumulate(r0, c0, tile_ptr, ignore_diags=igd, mode=mode)
                want = eng.fetch()
                for variant in (8, 8 | 4, 8 | 64):          # default, per-cell validity forced, tiles not paired
                    first = None
                    for rep in range(3):
                        eng.set_tuning(1, variant)           # one block per workgroup
                        eng.reset(T, pad)
                        eng.accumulate(r0, c0, tile_ptr, ignore_diags=igd, mode=mode)
                        got = eng.fetch()
                        assert eng.stats()["staged_regions"] > 150
                        for k in ("n", "num"):
                            np.testing.assert_array_equal(got[k], want[k], err_msg=f"{label} T={T} mode={mode} variant={variant}")
                        for k in ("sum", "cov_start", "cov_end"):
                            np.testing.assert_allclose(got[k], want[k], rtol=1e-11, atol=0, equal_nan=True)
                        if first is None:
                            first = got
                        else:                                # the same call again: bit-identical (fixed merge order)
                            np.testing.assert_array_equal(got["sum"], first["sum"])
    eng.close()


def test_page_locked_staging_gives_the_same_tiles(hip_lib):
    """Windows handed over from pup_host_alloc memory (asynchronous DMA on the engine's stream, no staging copy) and from
    ordinary numpy arrays (blocking copy) pile up to identical tiles; a second call may reuse the staging right away."""
    from coolpuppy_amd import engine as E
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chrA": 30_000_000, "chrB": 14_000_000}, lam=120, seed=11)
    rng = np.random.default_rng(3)
    n = 300_000
    r0 = rng.integers(0, 2900, n).astype(np.int32)
    c0 = (r0 + rng.integers(25, 400, n)).astype(np.int32)
    keep = c0 + 21 <= 3000
    r0, c0 = r0[keep], c0[keep]
    n = len(r0)
    tile_ptr = np.array([0, n // 3, n], np.int64)
    pr0, pc0 = E.pinned_empty(n), E.pinned_empty(n)
    assert E._POOL.broken is False, "pup_host_alloc failed on a GPU box"
    pr0[:], pc0[:] = r0, c0
    with PileupEngine(0) as eng:
        eng.load_pixels(*clr.pixel_table())
        eng.load_bins(clr.bins()["weight"][:].values, None)
        eng.build_index(clr.chrom_offset)
        eng.reset(2, 10)
        eng.accumulate(r0, c0, tile_ptr, ignore_diags=2, mode=0)
        want = eng.fetch()
        eng.reset(2, 10)
        eng.accumulate(pr0, pc0, tile_ptr, ignore_diags=2, mode=0)
        eng.accumulate(pr0, pc0, tile_ptr, ignore_diags=2, mode=0)       # queued behind the first call's kernels
        got = eng.fetch()
    np.testing.assert_array_equal(got["n"], 2 * want["n"])
    np.testing.assert_array_equal(got["num"], 2 * want["num"])
    np.testing.assert_allclose(got["sum"], 2 * want["sum"], rtol=1e-12, atol=0)


def test_sort_keys_relative_columns_and_their_fallback(hip_lib):
    """Chromosomes long enough for the block sort to store block columns relative to the block row (7 bits instead of the
    absolute column's): a call near the diagonal uses them; one with windows more than 127 blocks away from the diagonal
    falls back to absolute columns (and the context remembers).  Both must equal the per-window kernel."""
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chrL": 80_000_000, "chrM": 61_000_000}, lam=25, seed=21)
    nb = int(clr.chrom_offset[1])
    rng = np.random.default_rng(8)
    n = 120_000
    r0 = rng.integers(0, nb - 400, n)
    near = (r0 + rng.integers(0, 300, n)).astype(np.int32)
    far = near.copy()
    pick = rng.choice(n, 500, replace=False)
    far[pick] = np.minimum(r0[pick] + rng.integers(5700, 7600, 500), nb - 22).astype(np.int32)
    r0 = r0.astype(np.int32)
    tile_ptr = np.array([0, n // 2, n], np.int64)
    with PileupEngine(0) as eng:
        eng.load_pixels(*clr.pixel_table())
        eng.load_bins(clr.bins()["weight"][:].values, None)
        eng.build_index(clr.chrom_offset)
        for c0 in (near, far, near):                          # the third call runs after the fallback was remembered
            res = []
            for variant in (8, 16):
                eng.set_tuning(0, variant)
                eng.reset(2, 10)
                eng.accumulate(r0, c0, tile_ptr, ignore_diags=2, mode=0)
                res.append((eng.fetch(), eng.stats()["staged_regions"]))
            (a, staged), (b, _) = res
            assert staged > 0
            np.testing.assert_array_equal(a["n"], b["n"])
            np.testing.assert_array_equal(a["num"], b["num"])
            np.testing.assert_allclose(a["sum"], b["sum"], rtol=1e-12, atol=0)
