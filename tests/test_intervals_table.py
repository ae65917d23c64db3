"""The column store behind CoordCreator.intervals (coolpuppy_amd/intervals.py) against the pandas statement of the reference's
process() steps (CoordCreator._process_frame, forced with COOLPUPPY_AMD_FRAME_PATH=1): same frame — columns, order, dtypes, index
labels, values — same chromosome lists, same region rows, for bed and bedpe input, filters, subsets, rescaling, unsorted input,
extension dtypes; and the inputs the array path must decline."""
import numpy as np
import pandas as pd
import pytest

from coolpuppy_amd import coolpup
import golden_util as gu
import synth


def _both(monkeypatch, features, res, **kw):
    seed = kw.pop("_seed", 3)
    np.random.seed(seed)
    monkeypatch.delenv("COOLPUPPY_AMD_FRAME_PATH", raising=False)
    fast = coolpup.CoordCreator(features, res, **kw)
    np.random.seed(seed)
    monkeypatch.setenv("COOLPUPPY_AMD_FRAME_PATH", "1")
    slow = coolpup.CoordCreator(features, res, **kw)
    monkeypatch.delenv("COOLPUPPY_AMD_FRAME_PATH", raising=False)
    assert not slow._tbl.lazy
    return fast, slow


def _same(fast, slow):
    assert fast._tbl.lazy, "the array path declined an input it should take"
    assert fast.final_chroms == slow.final_chroms and fast.basechroms == slow.basechroms
    # engine-side columns first (before the frame exists), then the frame itself
    for name in ("stBin1", "endBin1", "stBin2", "endBin2") if fast.kind == "bedpe" else ("stBin", "endBin"):
        assert np.array_equal(fast._col(name), slow._col(name)), name
    assert fast._tbl._frame is None
    pd.testing.assert_frame_equal(fast.intervals, slow.intervals, check_exact=True)
    assert fast._tbl.names == list(slow.intervals.columns)


def test_array_table_equals_the_pandas_steps(monkeypatch):
    clr = synth.make_cooler({"chr1": 30_000_000, "chr2": 20_000_000, "chr10": 12_000_000, "chrX": 9_000_000}, lam=3, seed=2)
    pairs = synth.random_cis_pairs(clr, 60_000, seed=3, strands=True).sample(frac=1.0, random_state=1)      # unsorted, shuffled index
    pairs["score"] = np.random.default_rng(0).random(len(pairs)).astype(np.float32)
    pairs["strand1"] = pairs["strand1"].astype("category")
    pairs["label"] = pd.array(np.arange(len(pairs)), dtype="Int64")
    for kw in (dict(nshifts=3), dict(nshifts=0), dict(nshifts=2, mindist=0, maxdist=900_000), dict(nshifts=0, subset=5000, seed=4),
               dict(nshifts=0, rescale_flank=1.0), dict(nshifts=1, chroms=["chr2", "chrX", "chrNope"])):
        fast, slow = _both(monkeypatch, pairs, clr.binsize, features_format="bedpe", flank=50_000, **kw)
        _same(fast, slow)
        for reg in (("chr1", 0, 30_000_000), ("chr2", 2_000_000, 15_000_000), ("chr5", 0, 1)):
            a, b = fast._rows_pairs_region(reg), slow._rows_pairs_region(reg)
            a = np.arange(a.start, a.stop) if isinstance(a, slice) else a
            b = np.arange(b.start, b.stop) if isinstance(b, slice) else b
            assert np.array_equal(a, b)
    sorted_pairs = pairs.sort_values(["chrom1", "chrom2", "start1", "start2"]).reset_index(drop=True)
    _same(*_both(monkeypatch, sorted_pairs, clr.binsize, features_format="bedpe", flank=50_000, nshifts=1))    # no permutation
    trans = synth.random_trans_pairs(clr, 20_000, seed=5).sample(frac=1.0, random_state=2)
    fast, slow = _both(monkeypatch, trans, clr.binsize, features_format="bedpe", flank=250_000, nshifts=2, trans=True)
    _same(fast, slow)
    r1, r2 = ("chr1", 0, 30_000_000), ("chr2", 0, 20_000_000)
    a, b = fast._rows_trans_pairs(r1, r2), slow._rows_trans_pairs(r1, r2)
    assert np.array_equal(np.arange(a.start, a.stop) if isinstance(a, slice) else a, b)
    # bed features: unsorted, a string chromosome column given as a categorical, duplicates of (chrom, start) keep file order
    bed = pd.DataFrame({"chrom": pairs["chrom1"].to_numpy(), "start": pairs["start1"].to_numpy(), "end": pairs["end1"].to_numpy() + 40_000,
                        "name": np.arange(len(pairs)).astype(str)})
    bed = pd.concat([bed, bed.iloc[:500]], ignore_index=True)
    for kw in (dict(local=True), dict(nshifts=2, mindist=100_000, maxdist=600_000), dict(rescale_flank=0.5, local=True)):
        _same(*_both(monkeypatch, bed, clr.binsize, features_format="bed", flank=100_000, **kw))
    bed["chrom"] = bed["chrom"].astype("category")
    _same(*_both(monkeypatch, bed, clr.binsize, features_format="bed", flank=100_000, local=True))


def test_array_table_on_the_reference_feature_files(monkeypatch):
    for name in ("G1_bedpe_balanced", "G3_nshifts3", "G9_bed_combinations", "G5_local_expected_diag2", "G7_trans_bedpe_expected",
                 "G7b_trans_bed_product", "G12_rescale_local", "G12d_rescale_bedpe_controls", "KAT_bystrand_controls", "KAT_stripes"):
        assert name in gu.SCENARIOS
        z, meta, features, view, expected, kw = gu.load(name)
        clr = gu.scenario_cooler(meta)
        args = dict(features_format=kw["features_format"], flank=kw.get("flank", 100_000), nshifts=kw.get("nshifts", 0),
                    local=kw.get("local", False), trans=kw.get("trans", False), mindist=kw.get("mindist", "auto"),
                    maxdist=kw.get("maxdist"), rescale_flank=kw.get("rescale_flank") if kw.get("rescale") else None)
        _same(*_both(monkeypatch, features, clr.binsize, **args))


def test_array_path_declines_what_it_cannot_word_like_pandas(monkeypatch):
    clr = synth.make_cooler({"chr1": 30_000_000, "chr2": 20_000_000}, lam=3, seed=2)
    pairs = synth.random_cis_pairs(clr, 3000, seed=3)
    odd = pairs.copy()
    odd["start1"] = odd["start1"].astype(float)                    # float coordinates
    missing = pairs.copy()
    missing.loc[5, "chrom1"] = None                                # a missing chromosome name
    for frame in (odd, missing):
        fast, slow = _both(monkeypatch, frame, clr.binsize, features_format="bedpe", flank=50_000, nshifts=1)
        assert not fast._tbl.lazy
        pd.testing.assert_frame_equal(fast.intervals, slow.intervals)
    with pytest.warns(UserWarning, match="No regions in features"):
        cc = coolpup.CoordCreator(pairs, clr.binsize, features_format="bedpe", flank=50_000, mindist=10**9)
    assert cc.final_chroms == [] and len(cc.intervals) == 0
    # an assigned frame replaces the table
    cc = coolpup.CoordCreator(pairs, clr.binsize, features_format="bedpe", flank=50_000)
    sub = cc.intervals.iloc[:100]
    cc.intervals = sub
    assert cc._tbl.n == 100 and cc.intervals is sub
    assert len(cc._col("stBin1")) == 100


def test_bedpe_index_is_renumbered_even_when_the_filter_drops_nothing(monkeypatch):
    """the reference resets the index after its distance filter unconditionally (coolpup.py:321) and only then sorts — the
    labels of CoordCreator.intervals are the kept rows' POSITIONS in the input, whatever labels the caller's frame carried (here:
    a subset of a bigger frame, shuffled).  Both paths; with and without rows dropped."""
    clr = synth.make_cooler({"chr1": 30_000_000, "chr2": 20_000_000}, lam=3, seed=2)
    big = synth.random_cis_pairs(clr, 20_000, seed=3)
    pairs = big.iloc[5_000:15_000].sample(frac=1.0, random_state=1)            # labels 5000..14999, shuffled
    for kw in (dict(mindist=0, maxdist=np.inf), dict(mindist=400_000, maxdist=900_000)):
        fast, slow = _both(monkeypatch, pairs, clr.binsize, features_format="bedpe", flank=50_000, nshifts=0, **kw)
        _same(fast, slow)
        # the pandas statement of the reference's steps on the same frame: filter, reset_index, stable sort
        c1 = (pairs["start1"] + pairs["end1"]) / 2
        c2 = (pairs["start2"] + pairs["end2"]) / 2
        ref = pairs[(kw["mindist"] <= (c2 - c1).abs()) & ((c2 - c1).abs() <= kw["maxdist"])].reset_index(drop=True)
        ref = ref.sort_values(["chrom1", "chrom2", "start1", "start2"], kind="stable")
        assert list(fast.intervals.index) == list(ref.index)
        assert np.array_equal(fast.intervals["start1"].to_numpy(), ref["start1"].to_numpy())
        if kw["mindist"] == 0:
            assert len(ref) == len(pairs) and sorted(fast.intervals.index) == list(range(len(pairs)))
