"""CPU tests of host-side helpers that the benchmark and the tools rely on."""
import numpy as np
import pytest

from coolpuppy_amd import coolpup
import synth


def test_snippet_batches_matches_plan():
    clr = synth.make_cooler({"chrA": 12_000_000, "chrB": 8_000_000}, lam=30, seed=3)
    pairs = synth.random_cis_pairs(clr, 500, min_sep=230_000, max_sep=2_000_000, seed=1)
    np.random.seed(5)
    cc = coolpup.CoordCreator(pairs, clr.binsize, features_format="bedpe", flank=100_000, nshifts=3)
    r0, c0, kind = coolpup.snippet_batches(cc, clr, control=True)
    np.random.seed(5)
    cc2 = coolpup.CoordCreator(pairs, clr.binsize, features_format="bedpe", flank=100_000, nshifts=3)
    pu = coolpup.PileUpper(clr, cc2, control=True)
    pu.ignore_group_order = False
    batches = [(a, b, pu.region_snippets(a, b)) for a, b in pu._region_pairs()]
    plan = pu.make_plan(batches, [])
    assert len(plan["calls"]) == 1                       # regions without expected merge into one engine call
    call = plan["calls"][0]
    assert len(call["r0"]) == len(r0) and int(call["tile_ptr"][1]) == int((kind == 0).sum())
    assert sorted(zip(call["r0"].tolist(), call["c0"].tolist())) == sorted(zip(r0.tolist(), c0.tolist()))
    W = 21
    lo = {c: clr.extent(c) for c in clr.chromnames}
    for r, c in zip(r0[:50], c0[:50]):
        assert any(a <= r and r + W <= b and a <= c and c + W <= b for a, b in lo.values())


def test_rows_of_table_keeps_only_owned_rows():
    """A rank's partial pixel table: same number of rows, the rows it does not own empty, the others unchanged."""
    rng = np.random.default_rng(0)
    lens = rng.integers(0, 9, 50)
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
    col = rng.integers(0, 50, indptr[-1]).astype(np.int32)
    cnt = rng.integers(1, 99, indptr[-1]).astype(np.int32)
    rows = coolpup._merge_ranges([(30, 41), (5, 12), (10, 14)])
    assert rows == [(5, 14), (30, 41)]
    p2, c2, n2 = coolpup._rows_of_table(indptr, col, cnt, rows)
    assert len(p2) == len(indptr)
    for r in range(50):
        own = any(a <= r < b for a, b in rows)
        got = (c2[p2[r]:p2[r + 1]], n2[p2[r]:p2[r + 1]])
        if own:
            assert np.array_equal(got[0], col[indptr[r]:indptr[r + 1]]) and np.array_equal(got[1], cnt[indptr[r]:indptr[r + 1]])
        else:
            assert len(got[0]) == 0
    p3, c3, _ = coolpup._rows_of_table(indptr, col, cnt, [])
    assert p3[-1] == 0 and len(c3) == 0


def test_skip_region_leaves_the_generator_where_the_windows_would():
    """Draw-and-discard: stepping the control RNG past a region must consume exactly what generating it consumes."""
    import golden_util as gu
    for name in ("G3_nshifts3", "G9b_bed_combinations_controls_strand", "G7c_trans_bedpe_controls", "G5c_local_controls"):
        z, meta, features, view, expected, kw = gu.load(name)
        clr = gu.scenario_cooler(meta)
        cc = coolpup.CoordCreator(features, clr.binsize, features_format=kw["features_format"], flank=kw["flank"],
                                  nshifts=kw["nshifts"], seed=kw["seed"], local=kw.get("local", False),
                                  trans=kw.get("trans", False), mindist=kw.get("mindist", "auto"), maxdist=kw.get("maxdist"))
        chroms = list(clr.chromnames)
        regs = [(c, 0, int(clr.chromsizes[c])) for c in chroms]
        pairs = [(regs[0], regs[1]), (regs[0], regs[2])] if kw.get("trans") else [(r, None) for r in regs]
        np.random.seed(5)
        for a, b in pairs:
            cc.region_table(a, b, control=True)
        want = np.random.randint(0, 1 << 30)
        np.random.seed(5)
        for a, b in pairs:
            cc.skip_region(a, b, control=True)
        assert np.random.randint(0, 1 << 30) == want, name


def test_block_order_groups_snippets_by_tile_and_block():
    """PileupEngine.block_order (host helper, no GPU): tile-major, then blocks of (rows - W + 1) x (columns - W + 1) corners
    of the staged kernel's LDS region (128 x 128 up to W = 21, else 64 x 128) anchored at chromosome starts, position inside
    a block; it is a permutation."""
    from coolpuppy_amd.engine import PileupEngine
    rng = np.random.default_rng(4)
    chrom_offset = np.array([0, 1000, 1700, 2500])
    r0 = rng.integers(0, 2400, 5000)
    c0 = r0 + rng.integers(0, 90, 5000)
    tile = rng.integers(0, 3, 5000)
    for pad, (side_r, side_c) in ((10, (108, 108)), (3, (122, 122)), (15, (34, 98))):
        o = PileupEngine.block_order(r0, c0, chrom_offset, tile=tile, pad=pad)
        assert sorted(o.tolist()) == list(range(5000))
        start = chrom_offset[np.searchsorted(chrom_offset, r0, side="right") - 1]
        key = np.stack([tile, start + (r0 - start) // side_r, (c0 - start) // side_c, r0, c0], axis=1)[o]
        assert all(tuple(key[i]) <= tuple(key[i + 1]) for i in range(len(key) - 1))


def test_sign_draws_equal_numpy_choice():
    """_draw_signs must stay np.random.choice([-1, 1], m): same values, same generator state afterwards."""
    for m in (0, 1, 7, 100_003):
        np.random.seed(123)
        want = np.random.choice([-1, 1], m)
        after_want = np.random.randint(0, 1 << 30)
        np.random.seed(123)
        got = coolpup._draw_signs(m)
        assert np.array_equal(got, want) and got.dtype == want.dtype
        assert np.random.randint(0, 1 << 30) == after_want


def test_library_draws_equal_the_legacy_generator():
    """pup_host_mt_randint against np.random.randint of the installed numpy: the numbers, the generator state afterwards
    (key and position — hence every later draw), for ranges with and without rejection, a power-of-two range, the full
    32-bit range, negative bounds, a single-valued range, requests that end inside the current block / on a block boundary /
    many blocks later, several starting positions, int32 and int64 outputs, and draw-and-discard."""
    from coolpuppy_amd import engine as E
    cases = [((100_000, 1_000_000), 300_000), ((0, 2), 250_001), ((0, 1 << 32), 70_000), ((-50, 77), 5_000), ((5, 6), 4_000),
             ((0, 1 << 31), 50_000), ((10, 4_000_000_000), 20_000), ((0, 3), 2_048), ((0, 1 << 20), 624 * 8), ((7, 1000), 2_500)]
    for k, ((lo, hi), m) in enumerate(cases):
        for burn in (0, 1, 623, 624, 1000 + k):
            np.random.seed(1000 + k)
            np.random.randint(0, 1 << 30, burn)                   # some position inside a block
            st0 = np.random.get_state()
            want = np.random.randint(lo, hi, m)
            after = np.random.get_state()
            for dtype, discard in ((np.int64, False), (np.int32, False), (np.int64, True)):
                if dtype == np.int32 and max(abs(lo), abs(hi)) >= 2 ** 31:
                    continue
                np.random.set_state(st0)
                got = E.legacy_randint(lo, hi, m, discard=discard, dtype=dtype)
                now = np.random.get_state()
                assert np.array_equal(now[1], after[1]) and now[2] == after[2], (lo, hi, m, burn)
                if not discard:
                    assert got.dtype == dtype and np.array_equal(got, want), (lo, hi, m, burn)
    np.random.seed(5)
    want = 2 * np.random.randint(0, 2, 10_000) - 1
    np.random.seed(5)
    assert np.array_equal(E.legacy_randint(0, 2, 10_000, scale=2, offset=-1), want)


def test_planned_draws_equal_the_calls_made_one_by_one():
    """pup_host_mt_randint_plan (a pile-up's draws as one job: twister thread a buffer ahead of a worker pool) against
    np.random.randint called in sequence: numbers of every call and the final generator state, for starting positions inside /
    at the end of a block, int32 and int64 outputs, discarded calls, single-valued ranges, empty calls, a full 32-bit range,
    sequences that end exactly on a block and on a buffer boundary; two calls rejecting over different ranges are declined
    with the generator untouched."""
    from coolpuppy_amd import engine as E

    def check(calls, seed, burn, dtype):
        np.random.seed(seed)
        np.random.randint(0, 1000, burn)
        st0 = np.random.get_state()
        want = [of + sc * np.random.randint(lo, hi, m) for lo, hi, m, sc, of, _ in calls]
        end = np.random.get_state()
        np.random.set_state(st0)
        mine = [(lo, hi, m, sc, of, np.empty(m, dtype) if keep else None) for lo, hi, m, sc, of, keep in calls]
        assert E.legacy_randint_plan(mine)
        now = np.random.get_state()
        assert np.array_equal(now[1], end[1]) and now[2] == end[2], (seed, burn)
        for w, c in zip(want, mine):
            if c[5] is not None:
                assert np.array_equal(w, c[5]), (seed, burn, c[:3])
    rng = np.random.default_rng(1)
    for seed in range(3):
        for burn in (0, 1, 623, 624, 1000):
            calls = []
            for m in rng.integers(0, 5000, 5):
                calls += [(100_000, 1_000_000, int(m), 1, 0, True), (0, 2, int(m), 2, -1, True)]
            check(calls, seed, burn, np.int32)
            check(calls, seed, burn, np.int64)
    check([(5, 6, 10, 1, 0, True), (100_000, 1_000_000, 70_000, 1, 0, True), (0, 2, 70_000, 2, -1, False), (100_000, 1_000_000, 0, 1, 0, True),
           (0, 2, 0, 1, 0, True), (100_000, 1_000_000, 300_001, 1, 0, False), (0, 1 << 32, 1000, 1, 0, True), (0, 16, 123_457, 3, 7, True),
           (7, 8, 3, 2, 1, True)], 3, 17, np.int64)
    calls = []
    for m in rng.integers(100_000, 400_000, 6):                    # several buffers
        calls += [(100_000, 1_000_000, int(m), 1, 0, True), (0, 2, int(m), 2, -1, True)]
    check(calls, 11, 5, np.int32)
    for burn in (0, 624):
        check([(0, 2, 624 * 300 - burn, 1, 0, True)], 0, burn, np.int32)           # ends on a block boundary
        check([(0, 2, 624 * 256 * 3 - burn, 1, 0, True)], 0, burn, np.int32)       # ... and on a buffer boundary (256 blocks)
    np.random.seed(0)
    st = np.random.get_state()
    assert not E.legacy_randint_plan([(0, 3, 10, 1, 0, np.empty(10, np.int32)), (0, 5, 10, 1, 0, np.empty(10, np.int32))])
    assert np.array_equal(np.random.get_state()[1], st[1]) and np.random.get_state()[2] == st[2]


def test_library_window_pass_equals_numpy():
    """pup_host_windows (shift, bounds test, compaction; no GPU needed) against the numpy statement of the same rules —
    np.round's half-to-even included — for several shapes; pup_host_group_tiles against a stable argsort."""
    from coolpuppy_amd import engine as E
    rng = np.random.default_rng(0)
    for n, ns, res in ((0, 3, 10_000), (1, 0, 10_000), (1000, 3, 10_000), (257, 10, 5_000), (40_000, 2, 20_000), (150_001, 4, 10_000)):
        st1 = rng.integers(0, 5000, n).astype(np.int32)
        st2 = (st1 + rng.integers(-5, 300, n)).astype(np.int32)
        code = rng.integers(0, 7, n).astype(np.int32)
        shift = rng.integers(100_000, 1_000_000, n * ns)
        shift[: min(len(shift), 50)] = res * rng.integers(10, 90, min(len(shift), 50)) + res // 2      # exact halves: rounding rule
        sign = rng.choice([-1, 1], n * ns)
        for use_code in (True, False):
            r0, c0, co, nroi = E.host_windows(st1, st2, code if use_code else None, shift if ns else None, sign if ns else None,
                                              ns, res, 100, 100, 150, 5000, 150, 5200, 21, 21)
            d = np.round(shift * sign / res).astype(int).astype(np.int32)
            R = np.concatenate([st1, np.tile(st1, ns) + d]) + 100
            C = np.concatenate([st2, np.tile(st2, ns) + d]) + 100
            ok = (R >= 150) & (R + 21 <= 5000) & (C >= 150) & (C + 21 <= 5200)
            assert np.array_equal(r0, R[ok]) and np.array_equal(c0, C[ok]) and nroi == int(ok[:n].sum())
            if use_code:
                assert np.array_equal(co, np.concatenate([code, np.tile(code, ns)])[ok])
            else:
                assert co is None
    n = 30_000
    r0 = rng.integers(0, 9999, n).astype(np.int32)
    c0 = rng.integers(0, 9999, n).astype(np.int32)
    tile = rng.integers(0, 11, n).astype(np.int32)
    cuts = [0, 7, 7, 12_000, 29_999, n]                                   # parts of very different sizes, one empty
    parts = [(r0[a:b], c0[a:b], tile[a:b]) for a, b in zip(cuts[:-1], cuts[1:])]
    a, b, tp = E.group_tiles(parts, 11)
    o = np.argsort(tile, kind="stable")
    assert np.array_equal(a, r0[o]) and np.array_equal(b, c0[o])
    assert np.array_equal(tp, np.concatenate([[0], np.cumsum(np.bincount(tile, minlength=11))]))
    with pytest.raises(ValueError):
        E.group_tiles([(r0[:5], c0[:5], np.array([0, 1, 2, 11, 3], np.int32))], 11)
    # run-coded parts (engine.RunTile: the first `split` windows of a part in one tile, the rest in another) mixed with array-coded ones
    runs = [E.RunTile(0, 7, 3, 9), None, E.RunTile(5000, 11_993, 2, 10), E.RunTile(17_999, 17_999, 0, 99), None]
    mixed = [(p[0], p[1], p[2] if rt is None else rt) for p, rt in zip(parts, runs)]
    plain = [(p[0], p[1], p[2] if rt is None else np.asarray(rt)) for p, rt in zip(parts, runs)]
    got, want = E.group_tiles(mixed, 11), E.group_tiles(plain, 11)
    assert all(np.array_equal(x, y) for x, y in zip(got, want))
    tile_all = np.concatenate([p[2] for p in plain])
    o = np.argsort(tile_all, kind="stable")
    assert np.array_equal(got[0], r0[o]) and np.array_equal(got[2], np.concatenate([[0], np.cumsum(np.bincount(tile_all, minlength=11))]))
    with pytest.raises(ValueError):
        E.group_tiles([(r0[:5], c0[:5], E.RunTile(2, 5, 0, 11))], 11)


def test_draw_ahead_thread_issues_the_serial_draws(monkeypatch):
    """_DrawAhead (control shifts drawn on a helper thread, a few regions ahead) hands out the numbers the main thread would
    have drawn itself, region by region, and leaves numpy's legacy generator in the same state; a consumer out of step is an
    error, and close() after an abandoned run still ends the sequence where a serial run would."""
    from coolpuppy_amd import coolpup

    class CC:
        minshift, maxshift, trans = 100_000, 1_000_000, False
        _draw_raw_now = coolpup.CoordCreator._draw_raw_now
        _draw_dtype = coolpup.CoordCreator._draw_dtype
    sizes = [5000, 1, 70_000, 2048, 300_000, 12]
    for trans, planned in ((False, True), (True, True), (False, False), (True, False)):
        # (planned: the whole sequence as one library job, pup_host_mt_randint_plan; else call by call)
        monkeypatch.setenv("COOLPUPPY_AMD_NO_DRAW_PLAN", "") if planned else monkeypatch.setenv("COOLPUPPY_AMD_NO_DRAW_PLAN", "1")
        cc = CC()
        cc.trans = trans
        np.random.seed(7)
        want = [cc._draw_raw_now(m) for m in sizes]
        end = np.random.get_state()
        np.random.seed(7)
        ahead = coolpup._DrawAhead(cc, sizes, depth=2)
        got = [ahead.take(m) for m in sizes]
        ahead.close()
        for (a, b), (c, d) in zip(want, got):
            assert np.array_equal(a, c) and np.array_equal(b, d)
        now = np.random.get_state()
        assert now[2] == end[2] and np.array_equal(now[1], end[1])
        np.random.seed(7)
        ahead = coolpup._DrawAhead(cc, sizes, depth=2)
        ahead.take(5000)
        with pytest.raises(RuntimeError):
            ahead.take(99)
        ahead.close()
        now = np.random.get_state()
        assert now[2] == end[2] and np.array_equal(now[1], end[1])


def test_coordcreator_leaves_the_callers_frame_alone_and_factorises_by_identity():
    """CoordCreator works on a shallow copy of the feature frame: the caller's frame keeps its columns, dtypes and values whatever
    the options.  The chromosome columns are factorised by object identity in the library: same codes and uniques as pandas,
    missing values and equal-but-distinct string objects included."""
    import pandas as pd
    import synth
    from coolpuppy_amd import coolpup, engine as E
    clr = synth.make_cooler({"chr1": 30_000_000, "chr2": 20_000_000, "chrX": 9_000_000}, lam=5, seed=2)
    feats = synth.random_cis_pairs(clr, 80_000, seed=3, strands=True)
    feats = feats.sample(frac=1.0, random_state=1).reset_index(drop=True)          # unsorted: the sort permutes a new frame
    for kw in (dict(nshifts=3, seed=1), dict(nshifts=0, subset=1000, seed=2), dict(nshifts=2, mindist=0, maxdist=3_000_000, seed=3)):
        before = feats.copy()
        arrays = {c: feats[c].to_numpy() for c in feats.columns}
        coolpup.CoordCreator(feats, clr.binsize, features_format="bedpe", flank=50_000, **kw)
        assert list(feats.columns) == list(before.columns)
        pd.testing.assert_frame_equal(feats, before)
        for c in feats.columns:
            assert feats[c].to_numpy() is arrays[c] or np.array_equal(feats[c].to_numpy(), arrays[c])
    names = np.array([f"chr{k}" for k in range(23)] + [None, float("nan")], dtype=object)
    a = names[np.random.default_rng(0).integers(0, 25, 120_000)]
    a[5] = "chr" + "1"                                       # a distinct object equal to an existing string
    c1, u1 = E.factorize_objects(a)
    c2, u2 = pd.factorize(a)
    assert np.array_equal(c1, c2) and list(u1) == list(u2) and (c1 < 0).any()
    many = np.array([f"s{k % 90_000}" for k in range(100_000)], dtype=object)       # too many distinct objects: pandas takes it
    c1, u1 = E.factorize_objects(many)
    c2, u2 = pd.factorize(many)
    assert np.array_equal(c1, c2) and list(u1) == list(u2)


def test_host_pool_serves_threads_and_forked_children():
    """The array passes of the library share one pool of helper threads (pup_host.cpp: parallel_chunks): two Python threads calling
    at once both get the right answer (the second caller starts threads of its own), and a forked child — which inherits none of
    the helpers — makes a pool of its own instead of waiting for threads that do not exist."""
    import os
    import threading
    from coolpuppy_amd import engine as E
    rng = np.random.default_rng(9)
    keys = [rng.integers(0, 1 << 23, 400_000).astype(np.uint64) for _ in range(2)]
    want = [np.argsort(k, kind="stable") for k in keys]
    assert np.array_equal(E.stable_argsort(keys[0], 23), want[0])          # (the pool exists from here on)
    got = [None, None]

    def work(i):
        for _ in range(5):
            got[i] = E.stable_argsort(keys[i], 23)
    th = [threading.Thread(target=work, args=(i,)) for i in range(2)]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    pid = os.fork()
    if pid == 0:
        ok = False
        try:
            ok = np.array_equal(E.stable_argsort(keys[1], 23), want[1])
        finally:
            os._exit(0 if ok else 1)
    for _ in range(600):
        done, status = os.waitpid(pid, os.WNOHANG)
        if done:
            break
        import time
        time.sleep(0.05)
    else:
        os.kill(pid, 9)
        os.waitpid(pid, 0)
        raise AssertionError("the forked child hung in a library pass")
    assert os.WIFEXITED(status) and os.WEXITSTATUS(status) == 0


def test_sorted_combinations_equal_the_walk_over_offsets(monkeypatch):
    """CoordCreator._combination_table_sorted (BED features, no control draws: the pairs of every feature by two bisections in the
    order of the centres, brought into the reference's order by stable sorts) against the walk over the offsets the reference makes
    (coolpup.py:682-700): same rows, same order, same columns and dtypes — for several densities, ties between centres, float and
    automatic bounds, one- and two-feature chromosomes."""
    import pandas as pd
    from coolpuppy_amd import coolpup

    def tables(feats, walk, **kw):
        monkeypatch.setenv("COOLPUPPY_AMD_WALK_COMBINATIONS", "1") if walk else monkeypatch.delenv("COOLPUPPY_AMD_WALK_COMBINATIONS", raising=False)
        cc = coolpup.CoordCreator(features=feats, resolution=10_000, features_format="bed", flank=100_000, **kw)
        cc.process()
        return [cc.region_table((ch, 0, 10 ** 9), None, control=False, columns=None) for ch in ("chr1", "chr2")]
    rng = np.random.default_rng(0)
    for trial, n in enumerate((50, 400, 1500, 3, 1, 2)):
        rows = []
        for ch in ("chr1", "chr2"):
            st = np.sort(rng.integers(0, 30_000_000, n)) if trial != 2 else np.sort(rng.integers(0, 3_000_000, n) // 5000 * 5000)
            # (odd trials: ends that make the centres leave the order of the starts)
            rows.append(pd.DataFrame({"chrom": ch, "start": st, "end": st + rng.integers(1, 3 if trial % 2 == 0 else 400, n) * 1000}))
        feats = pd.concat(rows, ignore_index=True)
        for kw in (dict(mindist=300_000, maxdist=1_000_000), dict(mindist=0, maxdist=250_000.5), dict(mindist="auto"),
                   dict(mindist=220_000.25, maxdist=3e6)):
            for x, y in zip(tables(feats, True, **kw), tables(feats, False, **kw)):
                assert (x is None) == (y is None), (trial, kw)
                if x is not None:
                    assert list(x.keys()) == list(y.keys())
                    for k in x:
                        assert x[k].dtype == y[k].dtype and np.array_equal(x[k], y[k]), (trial, kw, k)


def test_library_tile_normalisation_equals_numpy():
    """pup_host_normalise_tiles (data / num, the ratio to the control's, +inf -> NaN on whole tile arrays, several threads) against
    the numpy expressions of the finaliser: same values, NaN where numpy has NaN, -inf kept, signs of zeros included."""
    from coolpuppy_amd.lib import puputils as P
    rng = np.random.default_rng(0)
    for ctrl in (False, True):
        S = rng.normal(size=(2000, 21, 21))
        N = rng.integers(0, 5, S.shape).astype(np.int64)
        S[N == 0] = 0.0
        S[0, 0, 0], N[0, 0, 0] = 1.0, 0              # +inf -> NaN
        S[0, 0, 1], N[0, 0, 1] = -1.0, 0             # -inf stays
        Sc, Nc = rng.normal(size=S.shape), rng.integers(0, 3, S.shape).astype(np.int64)
        with np.errstate(all="ignore"):
            want = S / N
            if ctrl:
                want = want / (Sc / Nc)
        want = np.where(want == np.inf, np.nan, want)
        got = P._normalise_tiles(S.copy(), N, Sc.copy() if ctrl else None, Nc if ctrl else None)
        assert np.array_equal(want, got, equal_nan=True)
        fin = ~np.isnan(want)
        assert np.array_equal(np.signbit(want[fin]), np.signbit(got[fin]))


def test_sorted_combinations_with_controls_equal_the_walk(monkeypatch):
    """The same with random-shift controls (nshifts > 0): the draws of all offsets as one library job and the rows gathered in one
    go (CoordCreator._combination_controls) against the walk's one randint + one choice and one small table per offset — same
    rows, columns, dtypes, and the legacy generator left in the same state."""
    import pandas as pd
    from coolpuppy_amd import coolpup

    def tables(feats, walk, **kw):
        monkeypatch.setenv("COOLPUPPY_AMD_WALK_COMBINATIONS", "1") if walk else monkeypatch.delenv("COOLPUPPY_AMD_WALK_COMBINATIONS", raising=False)
        np.random.seed(3)
        cc = coolpup.CoordCreator(features=feats, resolution=10_000, features_format="bed", flank=100_000, **kw)
        cc.process()
        out = [cc.region_table((ch, 0, 10 ** 9), None, control=True, columns=None) for ch in ("chr1", "chr2")]
        return out, np.random.get_state()
    rng = np.random.default_rng(1)
    for trial, n in enumerate((60, 700, 2, 1)):
        rows = []
        for ch in ("chr1", "chr2"):
            st = np.sort(rng.integers(0, 30_000_000, n))
            rows.append(pd.DataFrame({"chrom": ch, "start": st, "end": st + rng.integers(1, 3 if trial % 2 == 0 else 300, n) * 1000}))
        feats = pd.concat(rows, ignore_index=True)
        for kw in (dict(mindist=300_000, maxdist=1_000_000, nshifts=3), dict(mindist="auto", maxdist=2_000_000, nshifts=1),
                   dict(mindist=0, maxdist=250_000.5, nshifts=2, minshift=20_000, maxshift=77_777)):
            (ta, sa), (tb, sb) = tables(feats, True, **kw), tables(feats, False, **kw)
            assert sa[2] == sb[2] and np.array_equal(sa[1], sb[1]), (trial, kw)
            for x, y in zip(ta, tb):
                assert (x is None) == (y is None), (trial, kw)
                if x is not None:
                    assert list(x.keys()) == list(y.keys())
                    for k in x:
                        assert x[k].dtype == y[k].dtype and np.array_equal(x[k], y[k]), (trial, kw, k)


def test_library_lut_and_band_passes_equal_numpy():
    """pup_host_lut_i32 (tile numbers from group codes, the controls' half offset) and pup_host_count_le (distance bands =
    searchsorted(edges, d, "right")) against numpy, values on and beside the edges included; short inputs take numpy itself."""
    from coolpuppy_amd import engine as E
    from coolpuppy_amd.coolpup import _default_band_edges
    rng = np.random.default_rng(0)
    lut = rng.integers(0, 40, 25).astype(np.int32)
    for n, cut in ((500_000, 123_456), (500_000, 0), (500_000, 500_000), (1000, 500)):
        codes = rng.integers(0, 25, n).astype(np.int32)
        want = lut[codes].copy()
        want[cut:] += 7
        got = E.lut_codes(lut, codes, cut, 7)
        assert got.dtype == np.int32 and np.array_equal(got, want)
    edges = _default_band_edges()
    for n in (400_000, 1000):
        d = rng.integers(0, 6_000_000, n) + 0.5
        d[:10] = edges[:10]
        d[10:20] = edges[:10] - 0.5
        assert np.array_equal(E.count_le(edges, d), np.searchsorted(edges, d, side="right"))
    assert np.array_equal(E.count_le([10, 20, 30], np.full(200_000, 20.0)), np.full(200_000, 2))


def test_window_arena_never_reuses_memory_somebody_still_holds():
    """engine._ARENA (scratch for the per-region window arrays of a grouped pile-up, kept between pile-ups): reset() reuses the buffer
    only when no array handed out earlier — or a view of one — is still alive."""
    from coolpuppy_amd import engine as E
    a = E._ARENA
    a.reset()
    x = a.take(10)
    x[:] = 7
    first = id(a.buf)
    a.reset()                                  # x is alive: the buffer is left to it
    assert id(a.buf) != first and (x == 7).all()
    del x
    second = id(a.buf)
    a.reset()                                  # nobody holds anything: reused
    assert id(a.buf) == second
    y = a.take(5)
    z = y[:2]
    del y
    a.reset()                                  # a view of a view still pins it
    assert id(a.buf) != second
    del z


def test_library_argsort_equals_numpy_stable_argsort():
    from coolpuppy_amd import engine as E
    rng = np.random.default_rng(5)
    for n, bits in ((1, 1), (1000, 7), (200_000, 23), (300_001, 40), (150_000, 63)):
        keys = rng.integers(0, 2 ** min(bits, 62), n, dtype=np.int64)
        keys[:: 7] = keys[0]                                  # ties: index order
        assert np.array_equal(E.stable_argsort(keys, bits), np.argsort(keys, kind="stable"))


def test_library_take_rows_equals_numpy_take():
    """pup_host_take_rows (the permutation of the feature frame's numeric columns, several threads) against numpy's fancy
    index, every element size; object columns and short frames go through numpy; a bad index raises."""
    from coolpuppy_amd import engine as E
    rng = np.random.default_rng(3)
    n_src, n = 300_000, 250_000
    order = rng.integers(0, n_src, n)
    cols = [rng.integers(-2**40, 2**40, n_src), rng.random(n_src), rng.integers(0, 2**31 - 1, n_src).astype(np.int32),
            rng.integers(0, 60000, n_src).astype(np.uint16), rng.random(n_src) < 0.5, rng.random(n_src).astype(np.float32),
            np.array([f"chr{k % 7}" for k in range(n_src)], dtype=object), rng.random((n_src, 2))[:, 0]]
    got = E.take_rows(cols, order)
    for c, g in zip(cols, got):
        assert g.dtype == c.dtype and np.array_equal(g, c[order])
    small = E.take_rows([cols[0][:100]], np.array([5, 3, 99]))
    assert np.array_equal(small[0], cols[0][[5, 3, 99]])
    with pytest.raises(IndexError):
        E.take_rows([cols[0]], np.full(10_000, n_src))


def test_coverage_restatement_matches_hand_derived_answers():
    """oracle.coverage_numpy against the hand-worked table of tests/coverage_kat.py (cooltools' documented semantics:
    both bins of a pixel, the diagonal twice, ignore_diags on global ids incl. trans pixels, cis vs total)."""
    import coverage_kat as kat
    from oracle import pileup_oracle as po
    indptr, col, cnt = kat.table()
    for igd, (cis, tot) in kat.ANSWERS.items():
        got_cis, got_tot = po.coverage_numpy(indptr, col, cnt, kat.CHROM_OFFSET, igd)
        np.testing.assert_array_equal(got_cis, np.array(cis, float), err_msg=f"cis, ignore_diags={igd}")
        np.testing.assert_array_equal(got_tot, np.array(tot, float), err_msg=f"tot, ignore_diags={igd}")


def test_library_sort_pairs_equals_the_numpy_filter_and_sort():
    """pup_host_sort_pairs (centres, mindist / maxdist filter, stable sort by (chrom1, chrom2, start1, start2)) against numpy's
    statement of the same steps: rows kept, their order (ties in file order), the gathered columns, the two flags; inputs it must
    decline (negative start, keys beyond 63 bits)."""
    from coolpuppy_amd import engine as E
    rng = np.random.default_rng(11)
    for n, nu, lo, hi, unit in ((0, 3, 0, np.inf, 1), (1, 1, 0, np.inf, 1), (5000, 4, 50_000, 900_000, 10_000),
                                (200_000, 24, 230_000, np.inf, 10_000), (150_001, 7, 0, 2_000_000, 1)):
        s1 = rng.integers(0, 20_000, n) * unit
        s2 = s1 + rng.integers(-30, 200, n) * unit
        s2 = np.abs(s2)
        e1, e2 = s1 + rng.integers(1, 3, n) * unit, s2 + rng.integers(1, 3, n) * unit
        c1 = rng.integers(0, nu, n).astype(np.int32)
        c2 = np.where(rng.random(n) < 0.9, c1, rng.integers(0, nu, n)).astype(np.int32)
        if n > 100:
            s1[50:60], s2[50:60], c1[50:60], c2[50:60] = s1[40], s2[40], c1[40], c2[40]          # ties: file order
        rank = rng.permutation(nu).astype(np.int64)
        got = E.sort_pairs(s1, e1, s2, e2, c1, c2, rank, lo, hi)
        ca, cb = (s1 + e1) / 2, (s2 + e2) / 2
        keep = (lo <= np.abs(cb - ca)) & (np.abs(cb - ca) <= hi)
        idx = np.flatnonzero(keep)
        order = idx[np.lexsort((s2[idx], s1[idx], rank[c2[idx]], rank[c1[idx]]))]
        rows, a1, b1, a2, b2, k1, k2, filtered, permuted = got
        assert np.array_equal(rows, order)
        for g, w in ((a1, s1), (b1, e1), (a2, s2), (b2, e2), (k1, c1), (k2, c2)):
            assert np.array_equal(g, w[order]) and g.dtype == w.dtype
        assert filtered == (len(idx) != n) and permuted == (not np.array_equal(order, idx))
    s = np.arange(10, dtype=np.int64) * 1000
    got = E.sort_pairs(s, s + 10, s + 5000, s + 5010, np.zeros(10, np.int32), np.zeros(10, np.int32), np.zeros(1, np.int64), 0, np.inf)
    assert np.array_equal(got[0], np.arange(10)) and not got[7] and not got[8]                     # in order already
    neg = s.copy(); neg[3] = -5
    assert E.sort_pairs(neg, s + 10, s + 5000, s + 5010, np.zeros(10, np.int32), np.zeros(10, np.int32), np.zeros(1, np.int64), 0, np.inf) is None
    big = rng.integers(0, 2**40, 1000) | 1                                                         # odd: no common divisor
    assert E.sort_pairs(big, big + 1, big + 7, big + 9, np.zeros(1000, np.int32), np.zeros(1000, np.int32), np.zeros(1, np.int64), 0, np.inf) is None


def test_fused_plan_equals_the_per_region_plan():
    """PileUpper._fused_plan (ROI windows of every region, then the regions' control copies as the draws arrive, written straight
    into the engine call's arrays) against region_snippets -> make_plan -> group_tiles: same windows in the same order, same tile
    boundaries, same generator state afterwards, same lazily built per-region items — on whole chromosomes and on a view whose
    regions cut features and shifted copies off."""
    import pandas as pd
    clr = synth.make_cooler({"chr1": 30_000_000, "chr2": 20_000_000, "chrX": 9_000_000}, lam=3, seed=2)
    feats = synth.random_cis_pairs(clr, 30_000, seed=3).sample(frac=1.0, random_state=1)
    view = pd.DataFrame({"chrom": ["chr1", "chr1", "chr2"], "start": [0, 16_000_000, 1_000_000], "end": [15_000_000, 30_000_000, 18_000_000],
                         "name": ["a", "b", "c"]})
    for view_df in (None, view):
        plans = []
        for fused in (True, False):
            np.random.seed(11)
            cc = coolpup.CoordCreator(feats, clr.binsize, features_format="bedpe", flank=100_000, nshifts=4, seed=11)
            pu = coolpup.PileUpper(clr, cc, view_df=view_df, control=True)
            pu.ignore_group_order = False
            pairs = pu._region_pairs()
            if fused:
                plan = pu._fused_plan(pairs)
            else:
                batches = [(a, b, pu.region_snippets(a, b)) for a, b in pairs]
                plan = pu.make_plan(batches, [])
            plans.append((plan, np.random.randint(0, 1 << 30)))
        (pf, sf), (pp, sp) = plans
        assert sf == sp
        assert len(pf["calls"]) == len(pp["calls"]) == 1
        cf, cp = pf["calls"][0], pp["calls"][0]
        for k in ("r0", "c0", "tile_ptr"):
            assert np.array_equal(cf[k], cp[k]), k
        assert cf["region1"] == cp["region1"] and cf["mode"] == cp["mode"] and cf["ignore_diags"] == cp["ignore_diags"]
        for k in ("T", "G", "gid", "order", "want_control", "grouped", "pad", "n_regions", "region_groups", "weight_name", "cov_name"):
            assert pf[k] == pp[k], k
        assert len(pf["region_items"]) == len(pp["region_items"])
        for a, b in zip(pf["region_items"], pp["region_items"]):
            assert a[0] == b[0] and a[11] == b[11] and np.array_equal(a[3], b[3]) and np.array_equal(a[4], b[4])
            assert np.array_equal(np.asarray(a[6]), np.asarray(b[6]))
        if view_df is not None:
            assert int(cf["tile_ptr"][2]) < 5 * int(cf["tile_ptr"][1]) + 5 * 30_000      # (windows were dropped somewhere)
