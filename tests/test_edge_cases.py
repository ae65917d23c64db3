"""Edge cases: argument validation mirrors the reference's exceptions (coolpup.py:339-340, 379-380, 965-974, 1434-1441,
1786-1788, 2089-2108), degenerate inputs on the host, and window-size limits of the engine."""
import warnings

import numpy as np
import pandas as pd
import pytest

import golden_util as gu
from coolpuppy_amd import coolpup


def _bed(clr, n=12, seed=1):
    rng = np.random.default_rng(seed)
    chrom = clr.chromnames[0]
    size = int(clr.chromsizes[chrom])
    start = np.sort(rng.integers(2_000_000, size - 2_000_000, n)) // clr.binsize * clr.binsize
    return pd.DataFrame({"chrom": chrom, "start": start, "end": start + clr.binsize,
                         "strand": rng.choice(["+", "-"], n)})


def _bedpe(clr, n=10, seed=2):
    b = _bed(clr, 2 * n, seed)
    a, c = b.iloc[:n].reset_index(drop=True), b.iloc[n:].reset_index(drop=True)
    return pd.DataFrame({"chrom1": a.chrom, "start1": a.start, "end1": a.end, "chrom2": c.chrom, "start2": c.start,
                         "end2": c.end})


def test_incompatible_options_raise_like_the_reference():
    clr = gu.cooler("small")
    bed, bedpe = _bed(clr), _bedpe(clr)
    with pytest.raises(ValueError, match="local"):
        coolpup.CoordCreator(bedpe, clr.binsize, features_format="bedpe", local=True)
    with pytest.raises(ValueError, match="local"):
        coolpup.CoordCreator(bed, clr.binsize, features_format="bed", local=True, trans=True)
    with pytest.raises(ValueError, match="kind"):
        coolpup.CoordCreator(bed, clr.binsize, features_format="bed12")
    with pytest.raises(ValueError, match="multiple of the resolution"):
        coolpup.CoordCreator(bed, clr.binsize, features_format="bed", flank=clr.binsize * 3 + 1)
    with pytest.raises(ValueError, match="Can't determine kind"):
        coolpup.CoordCreator(bed.rename(columns={"chrom": "chr"}), clr.binsize)
    cc = coolpup.CoordCreator(bed, clr.binsize, features_format="bed", local=True, flank=100_000)
    with pytest.raises(ValueError, match="rescale_flank"):
        coolpup.PileUpper(clr, cc, rescale=True)
    ccr = coolpup.CoordCreator(bed, clr.binsize, features_format="bed", local=True, rescale_flank=1)
    with pytest.raises(ValueError, match="odd rescale_size"):
        coolpup.PileUpper(clr, ccr, rescale=True, rescale_size=40)
    with pytest.raises(ValueError, match="coverage normalization"):
        coolpup.PileUpper(clr, cc, coverage_norm="cov_tot_raw", clr_weight_name="weight")
    with pytest.raises(ValueError, match="not found"):
        coolpup.PileUpper(clr, cc, coverage_norm="no_such_column", clr_weight_name=None)
    with pytest.raises(ValueError, match="expected is not valid"):
        coolpup.PileUpper(clr, cc, expected=pd.DataFrame({"region1": ["chrA"], "value": [1.0]}))
    with pytest.raises(ValueError, match="by-window pileups for local"):
        coolpup.PileUpper(clr, cc).pileupsByWindowWithControl()
    with pytest.raises(ValueError, match="ignore_diags"):
        coolpup.PileUpper(clr, cc, ignore_diags=-1)


def test_no_common_chromosomes_and_empty_feature_sets():
    clr = gu.cooler("small")
    bed = _bed(clr)
    other = bed.assign(chrom="chrZZ")
    with pytest.raises(ValueError, match="No chromosomes are in common"):
        coolpup.pileup(clr, other, features_format="bed", flank=100_000)
    # every pair closer than mindist: the reference warns and ends up with no chromosome to work on
    near = _bedpe(clr)
    near["start2"] = near["start1"] + clr.binsize
    near["end2"] = near["end1"] + clr.binsize
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        cc = coolpup.CoordCreator(near, clr.binsize, features_format="bedpe", flank=100_000)
    assert any("No regions in features" in str(x.message) for x in w)
    assert cc.final_chroms == [] and list(cc.pos_stream(None)) == []
    with pytest.raises(ValueError, match="No chromosomes are in common"):
        coolpup.PileUpper(clr, cc)


def test_windows_leaving_their_region_are_skipped_not_clipped(monkeypatch):
    """Reference :1111-1114: a window crossing the region edge is dropped and n is not incremented."""
    monkeypatch.setattr(coolpup.PileUpper, "run_plan", gu.oracle_run_plan)
    clr = gu.cooler("small")
    chrom = clr.chromnames[0]
    size = int(clr.chromsizes[chrom])
    feats = pd.DataFrame({"chrom1": chrom, "start1": [0, 3_000_000, size - clr.binsize],
                          "end1": [clr.binsize, 3_000_000 + clr.binsize, size],
                          "chrom2": chrom, "start2": [5_000_000, 8_000_000, size - clr.binsize],
                          "end2": [5_000_000 + clr.binsize, 8_000_000 + clr.binsize, size]})
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = coolpup.pileup(clr, feats, features_format="bedpe", flank=100_000, mindist=0)
    assert int(df["n"].iloc[0]) == 1          # first window leaves the chromosome start, last one its end
    assert df["data"].iloc[0].shape == (21, 21)


@pytest.mark.gpu
def test_widest_supported_window_and_beyond(hip_lib):
    """W = 255 (pad 127) was the banded kernel's limit until round 3; the reference slices any width (coolpuppy/coolpup.py:1115-1121)
    and so does the engine now: parity against the oracle at 255 and at 257 / 351 bins (column panels of the banded kernel)."""
    from coolpuppy_amd.engine import PileupEngine, PupError
    from oracle import pileup_oracle as po
    clr = gu.cooler("small")
    indptr, col, cnt = clr.pixel_table()
    weight = clr.bins()["weight"][:].values
    rng = np.random.default_rng(3)
    hiA = int(clr.chrom_offset[1])
    pad = 127
    W = 2 * pad + 1
    r0 = rng.integers(0, hiA - 2 * W, 40).astype(np.int32)
    c0 = (r0 + rng.integers(0, W, 40)).astype(np.int32)
    tile = np.zeros(40, np.int32)
    eng = PileupEngine(0)
    eng.load_pixels(indptr, col, cnt)
    eng.build_index(clr.chrom_offset)
    eng.load_bins(weight, None)
    eng.reset(1, pad)
    eng.accumulate(r0, c0, np.array([0, 40]), ignore_diags=2, mode=0)
    got = eng.fetch()
    want = po.pileup_c(indptr, col, cnt, weight, None, None, r0, c0, None, tile, 1, pad, 2, 0)
    np.testing.assert_array_equal(got["num"], want["num"])
    np.testing.assert_allclose(got["sum"], want["sum"], rtol=1e-12, atol=0)
    for pad in (128, 175):
        W = 2 * pad + 1
        r0 = rng.integers(0, hiA - 2 * W, 30).astype(np.int32)
        c0 = (r0 + rng.integers(0, W, 30)).astype(np.int32)
        eng.reset(1, pad)
        eng.accumulate(r0, c0, np.array([0, 30]), ignore_diags=2, mode=0)
        got = eng.fetch()
        want = po.pileup_c(indptr, col, cnt, weight, None, None, r0, c0, None, np.zeros(30, np.int32), 1, pad, 2, 0)
        np.testing.assert_array_equal(got["num"], want["num"])
        np.testing.assert_allclose(got["sum"], want["sum"], rtol=1e-12, atol=0)
    del PupError
    eng.close()


@pytest.mark.gpu
def test_empty_and_single_snippet_calls(hip_lib):
    from coolpuppy_amd.engine import PileupEngine
    clr = gu.cooler("small")
    indptr, col, cnt = clr.pixel_table()
    eng = PileupEngine(0)
    eng.load_pixels(indptr, col, cnt)
    eng.load_bins(clr.bins()["weight"][:].values, None)
    eng.reset(3, 10)
    eng.accumulate(np.zeros(0, np.int32), np.zeros(0, np.int32), np.array([0, 0, 0, 0]), ignore_diags=2, mode=0)
    out = eng.fetch()
    assert out["n"].tolist() == [0, 0, 0] and not out["sum"].any() and not out["num"].any()
    eng.accumulate(np.array([100], np.int32), np.array([130], np.int32), np.array([0, 0, 1, 1]), ignore_diags=2, mode=0)
    out = eng.fetch()
    assert out["n"].tolist() == [0, 1, 0] and out["num"][1].max() == 1 and not out["num"][0].any()
    assert eng.extract(np.zeros(0, np.int32), np.zeros(0, np.int32), 10).shape == (0, 21, 21)
    h, v = eng.stripes(np.zeros(0, np.int32), np.zeros(0, np.int32), 10)
    assert h.shape == (0, 21) and v.shape == (0, 21)
    eng.close()
