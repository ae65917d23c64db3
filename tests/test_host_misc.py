"""CPU tests of host-side helpers that the benchmark and the tools rely on."""
import numpy as np

from coolpuppy_amd import coolpup, synth


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


def test_slice_call_partitions_every_segment():
    from coolpuppy_amd import dist as pdist
    rng = np.random.default_rng(0)
    n, T = 1000, 5
    tile = np.sort(rng.integers(0, T, n)).astype(np.int32)
    flip = np.zeros(n, bool)
    for t in range(T):                                    # flipped snippets last inside each tile
        idx = np.flatnonzero(tile == t)
        flip[idx[len(idx) // 3 * 2:]] = True
    call = coolpup._engine_call("r", "r", None, np.arange(n), np.arange(n) + 7, flip, tile.astype(np.int64), T, 2, 0)
    seen = []
    for rank in range(3):
        s = pdist.slice_call(call, rank, 3)
        seen.append(s["r0"])
        assert np.array_equal(np.concatenate([[0], np.cumsum(np.bincount(s["tile"], minlength=T))]), s["tile_ptr"])
        for t in range(T):
            seg = s["flip"][s["tile_ptr"][t]:s["tile_ptr"][t + 1]].astype(bool)
            k = int(s["flip_from"][t] - s["tile_ptr"][t])
            assert not seg[:k].any() and seg[k:].all()
    assert sorted(np.concatenate(seen).tolist()) == list(range(n))



def test_block_order_groups_snippets_by_tile_and_block():
    """PileupEngine.block_order (host helper, no GPU): tile-major, then (65 - W)^2 blocks of corners anchored at
    chromosome starts, position inside a block; it is a permutation."""
    from coolpuppy_amd.engine import PileupEngine
    rng = np.random.default_rng(4)
    chrom_offset = np.array([0, 1000, 1700, 2500])
    r0 = rng.integers(0, 2400, 5000)
    c0 = r0 + rng.integers(0, 90, 5000)
    tile = rng.integers(0, 3, 5000)
    for pad, side in ((10, 44), (3, 58), (15, 34)):
        o = PileupEngine.block_order(r0, c0, chrom_offset, tile=tile, pad=pad)
        assert sorted(o.tolist()) == list(range(5000))
        start = chrom_offset[np.searchsorted(chrom_offset, r0, side="right") - 1]
        key = np.stack([tile, start + (r0 - start) // side, (c0 - start) // side, r0, c0], axis=1)[o]
        assert all(tuple(key[i]) <= tuple(key[i + 1]) for i in range(len(key) - 1))
