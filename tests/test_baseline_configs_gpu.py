"""The five BASELINE.json configurations as GPU tests, through the public API.

configs[0] / [1] use the reference's own feature files (tests/golden/ref_data: CH12_loops_Rao.bed, Bonev_CTCF+/-.bed.gz)
on an mm9-sized synthetic 10 kb table (the real Scc1-control.10000.cool is not in the tree) and are compared IN FULL
with the CPU oracle: the same pileup() call with the engine half of run_plan replaced by the oracle replay.
configs[2] - [4] run at BASELINE's full size on the human-scale synthetic table; configs[2] / [3] compare EVERY window
with the oracle (its row-sliced OpenMP form) and assert that the workgroup-staged kernel K1q served them; in addition the
oracle checks a strided sample of >= 6e4 windows of every engine call (which the engine serves with K1r), and the whole
run is checked through properties that do not depend on the size: window counts, additivity over a split of the windows, "all" = sum of the groups, num <= n, and (trans) agreement
between the kernel the engine picks and the plain per-window kernel.

Bit-exact for n / num; 1e-6 relative (BASELINE north_star) for the float sums — the test passes 1e-9.
"""
import gzip
import os
import warnings
from functools import partial

import numpy as np
import pandas as pd
import pytest

import golden_util as gu
from coolpuppy_amd import coolpup
import synth

pytestmark = pytest.mark.gpu
REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_data")
RTOL = 1e-9


@pytest.fixture(scope="module")
def mm9(hip_lib):
    return synth.make_cooler(synth.MM9, binsize=10_000, lam=120, seed=1000, name="synthetic_mm9_10kb", parallel=True)


@pytest.fixture(scope="module")
def hg38(hip_lib):
    return synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=4200, seed=1000,
                             name="synthetic_hg38_10kb", parallel=True, trans_nnz=50_000_000)


def _frames_equal(gpu, cpu):
    assert list(gpu.columns) == list(cpu.columns) and len(gpu) == len(cpu)
    for i in range(len(gpu)):
        assert int(gpu["n"].iloc[i]) == int(cpu["n"].iloc[i])
        np.testing.assert_array_equal(np.asarray(gpu["num"].iloc[i]), np.asarray(cpu["num"].iloc[i]))
        np.testing.assert_allclose(np.asarray(gpu["data"].iloc[i], float), np.asarray(cpu["data"].iloc[i], float),
                                   rtol=RTOL, atol=0, equal_nan=True)


def _both(clr, features, monkeypatch, **kw):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        gpu = coolpup.pileup(clr, features, **kw)
        with monkeypatch.context() as m:
            m.setattr(coolpup.PileUpper, "run_plan", gu.oracle_run_plan)
            cpu = coolpup.pileup(clr, features, **kw)
    return gpu, cpu


def test_config0_ch12_loops_full_oracle(mm9, monkeypatch, oracle_mod):
    """configs[0]: CH12_loops_Rao.bed (as bedpe), pad=10, nshifts=0, balanced — every window against the oracle."""
    loops = pd.read_csv(os.path.join(REF, "CH12_loops_Rao.bed"), sep="\t", header=None,
                        names=["chrom1", "start1", "end1", "chrom2", "start2", "end2"])
    gpu, cpu = _both(mm9, loops, monkeypatch, features_format="bedpe", flank=100_000, nshifts=0)
    _frames_equal(gpu, cpu)
    d = np.asarray(gpu["data"].iloc[0])
    assert d.shape == (21, 21) and not np.isnan(d).any()
    assert int(gpu["n"].iloc[0]) > 1000           # the file holds 1 388 usable loops on the chromosomes of the table


@pytest.mark.parametrize("sign", ["+", "-"])
def test_config1_ctcf_local_expected_full_oracle(mm9, sign, monkeypatch, oracle_mod):
    """configs[1]: Bonev_CTCF+/- local cis pile-up, pad=10, observed over expected — every window against the oracle."""
    with gzip.open(os.path.join(REF, f"Bonev_CTCF{sign}.bed.gz"), "rt") as f:
        bed = pd.read_csv(f, sep="\t", header=None, names=["chrom", "start", "end"])
    exp = synth.cis_expected(mm9)
    gpu, cpu = _both(mm9, bed, monkeypatch, features_format="bed", flank=100_000, local=True, expected_df=exp)
    _frames_equal(gpu, cpu)
    # ignore_diags=2 blanks the main diagonal and its neighbours of an on-diagonal window: 21 + 2*20 cells
    assert int(np.isnan(np.asarray(gpu["data"].iloc[0])).sum()) == 61
    assert int(gpu["n"].iloc[0]) > 10_000


# ---- full-size configurations -------------------------------------------------------------------------------------------
def _plan(clr, features, seed=None, groupby=(), modify=None, cols=(), **kw):
    if seed is not None:
        np.random.seed(seed)
    cc = coolpup.CoordCreator(features, clr.binsize, features_format="bedpe", flank=kw["flank"], nshifts=kw.get("nshifts", 0),
                              trans=kw.get("trans", False), chroms=list(clr.chromnames), seed=seed)
    pu = coolpup.PileUpper(clr, cc, control=kw.get("nshifts", 0) > 0, ignore_diags=2)
    pu.ignore_group_order = False
    batches = [(r1, r2, pu.region_snippets(r1, r2, groupby=list(groupby), modify_2Dintervals_func=modify, columns=cols))
               for r1, r2 in pu._region_pairs()]
    return pu, pu.make_plan(batches, list(groupby))


def _run_calls(pu, plan, calls):
    eng = coolpup._engine_for(pu._aclr, 0)
    bins = pu._aclr.bins()
    eng.load_bins(bins[plan["weight_name"]][:].values, None)
    eng.reset(plan["T"], plan["pad"])
    for c in calls:
        eng.accumulate(c["r0"], c["c0"], c["tile_ptr"], flip_from=c["flip_from"], ignore_diags=c["ignore_diags"], mode=c["mode"])
    return eng.fetch()


def _subset(plan, call, idx):
    flip = None if call["flip"] is None else call["flip"][idx]
    return coolpup._engine_call(call["region1"], call["region2"], call["expected"], call["r0"][idx].astype(np.int64),
                                call["c0"][idx].astype(np.int64), flip, call["tile"][idx].astype(np.int64), plan["T"],
                                call["ignore_diags"], call["mode"])


def _check_full_size(pu, plan, acc, min_sample=60_000):
    from oracle import pileup_oracle as po
    T = plan["T"]
    total = sum(len(c["r0"]) for c in plan["calls"])
    # window counts: every accepted window is counted once in its tile
    want_n = np.zeros(T, np.int64)
    for c in plan["calls"]:
        want_n += np.diff(c["tile_ptr"])
    np.testing.assert_array_equal(acc["n"], want_n)
    assert (acc["num"] <= acc["n"][:, None, None]).all() and (acc["num"] >= 0).all()
    assert np.isfinite(acc["sum"]).all()
    # additivity: even windows + odd windows = all windows (integers exactly, sums to rounding)
    parts = []
    for k in (0, 1):
        parts.append(_run_calls(pu, plan, [_subset(plan, c, np.arange(k, len(c["r0"]), 2)) for c in plan["calls"]]))
    np.testing.assert_array_equal(parts[0]["n"] + parts[1]["n"], acc["n"])
    np.testing.assert_array_equal(parts[0]["num"] + parts[1]["num"], acc["num"])
    np.testing.assert_allclose(parts[0]["sum"] + parts[1]["sum"], acc["sum"], rtol=1e-10, atol=0)
    # strided sample of every call against the oracle
    step = max(1, total // (min_sample + 2000))
    calls = [_subset(plan, c, np.arange(0, len(c["r0"]), step)) for c in plan["calls"]]
    n_s = sum(len(c["r0"]) for c in calls)
    assert n_s >= min_sample
    got = _run_calls(pu, plan, calls)
    indptr, col, cnt = pu._aclr.pixel_table()
    weight = pu._aclr.bins()[plan["weight_name"]][:].values
    ref = po.empty_acc(T, plan["pad"])
    for c in calls:
        po.pileup_c(indptr, col, cnt, weight, None, None, c["r0"], c["c0"], c["flip"], c["tile"], T, plan["pad"],
                    c["ignore_diags"], c["mode"], acc=ref)
    np.testing.assert_array_equal(got["n"], ref["n"])
    np.testing.assert_array_equal(got["num"], ref["num"])
    np.testing.assert_allclose(got["sum"], ref["sum"], rtol=RTOL, atol=0)
    return n_s


def _check_every_window_vs_oracle(pu, plan, acc, expect_staged=True):
    """ALL windows of the plan against the oracle (row-sliced OpenMP form of oracle/pileup_oracle.c, ~1-2 s for 1.1e7
    windows), and the engine must have taken the workgroup-staged kernel (K1q) for them: a strided sample is too small for
    the engine to choose K1q, so only this comparison puts K1q itself — at full size — against the oracle."""
    from oracle import pileup_oracle as po
    eng = coolpup._engine_for(pu._aclr, 0)
    st = eng.stats()
    if expect_staged:
        assert st["staged_regions"] > 0, "the engine did not take the workgroup-staged kernel at full size"
    indptr, col, cnt = pu._aclr.pixel_table()
    weight = pu._aclr.bins()[plan["weight_name"]][:].values
    T = plan["T"]
    ref = po.empty_acc(T, plan["pad"])
    nthr = max(1, min(os.cpu_count() or 1, 64))
    for c in plan["calls"]:
        po.pileup_c_mt(indptr, col, cnt, weight, None, None, c["r0"], c["c0"], c["flip"], c["tile"], T, plan["pad"],
                       c["ignore_diags"], c["mode"], nthr, acc=ref)
    np.testing.assert_array_equal(acc["n"], ref["n"])
    np.testing.assert_array_equal(acc["num"], ref["num"])
    np.testing.assert_allclose(acc["sum"], ref["sum"], rtol=RTOL, atol=0)


def test_config2_million_pairs_ten_shifts(hg38, oracle_mod):
    """configs[2]: 1e6 random cis pairs, nshifts=10 — the benchmark's workload, through the coordinate layer."""
    feats = synth.random_cis_pairs(hg38, 1_000_000, seed=42, strands=True)
    pu, plan = _plan(hg38, feats, seed=0, flank=100_000, nshifts=10)
    acc = _run_calls(pu, plan, plan["calls"])
    assert plan["T"] == 2 and acc["n"][0] > 990_000 and acc["n"][1] > 9_900_000
    _check_every_window_vs_oracle(pu, plan, acc)
    _check_full_size(pu, plan, acc)
    # and the public entry point returns the same numbers
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = coolpup.pileup(hg38, feats, features_format="bedpe", flank=100_000, nshifts=10, seed=0)
    assert int(df["n"].iloc[0]) == int(acc["n"][0]) and int(df["control_n"].iloc[0]) == int(acc["n"][1])
    np.testing.assert_array_equal(np.asarray(df["num"].iloc[0]), acc["num"][0])
    # (pileup() draws the control shifts on a helper thread, the plan above on the calling thread: same windows — and the
    # same pile-up, bit for bit, with the helper switched off)
    os.environ["COOLPUPPY_AMD_NO_DRAW_AHEAD"] = "1"
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            df2 = coolpup.pileup(hg38, feats, features_format="bedpe", flank=100_000, nshifts=10, seed=0)
    finally:
        os.environ.pop("COOLPUPPY_AMD_NO_DRAW_AHEAD", None)
    np.testing.assert_array_equal(np.asarray(df["data"].iloc[0]), np.asarray(df2["data"].iloc[0]))
    assert int(df2["control_n"].iloc[0]) == int(df["control_n"].iloc[0])


@pytest.mark.parametrize("case", ["ooe_expected_table", "flip_negative_strand"])
def test_staged_kernel_full_size_ooe_and_flip_vs_oracle(hg38, oracle_mod, case):
    """The workgroup-staged kernel at >= 1e6 windows in its two other shapes, EVERY window against the oracle
    (reference coolpup.py:1104-1157, lib/puputils.py:12-41): observed over the per-chromosome expected (the expected
    table: one engine call spanning all regions, OOE instantiation, no factorised num) and strand-flipped windows (flip
    segments, anti-transposed flush).  The plan is replayed on the C oracle by golden_util.oracle_run_plan."""
    ooe = case == "ooe_expected_table"           # (expected and control shifts exclude each other, reference :1437-1442)
    feats = synth.random_cis_pairs(hg38, 1_200_000 if ooe else 600_000, seed=7, strands=True)
    np.random.seed(3)
    cc = coolpup.CoordCreator(feats, hg38.binsize, features_format="bedpe", flank=100_000, nshifts=0 if ooe else 1,
                              chroms=list(hg38.chromnames), seed=3)
    kw = dict(expected=synth.cis_expected(hg38), ooe=True, control=False) if ooe else \
        dict(flip_negative_strand=True, control=True)
    pu = coolpup.PileUpper(hg38, cc, ignore_diags=2, **kw)
    pu.ignore_group_order = False
    # what pileupsWithControl passes on for flip_negative_strand (reference :1431-1475): the interval-marking function
    modify = None if ooe else partial(coolpup.flip_mark_intervals_func, flipby="strand", flip_negative_strand=True,
                                      extra_func=None)
    batches = [(r1, r2, pu.region_snippets(r1, r2, groupby=[], modify_2Dintervals_func=modify,
                                           columns=() if ooe else ["strand1"]))
               for r1, r2 in pu._region_pairs()]
    plan = pu.make_plan(batches, [])
    assert sum(len(c["r0"]) for c in plan["calls"]) >= 1_000_000
    if case == "flip_negative_strand":
        assert any(c["flip"] is not None and c["flip"].any() for c in plan["calls"])
    got = pu.run_plan(plan, reduce=False)
    assert coolpup._engine_for(pu._aclr, 0).stats()["staged_regions"] > 0, "K1q did not run"
    ref = gu.oracle_run_plan(pu, plan, reduce=False)
    np.testing.assert_array_equal(got["n"], ref["n"])
    np.testing.assert_array_equal(got["num"], ref["num"])
    np.testing.assert_allclose(got["sum"], ref["sum"], rtol=RTOL, atol=0)


def test_config3_by_distance_by_strand(hg38, oracle_mod):
    """configs[3]: the same pairs grouped by distance band and strand pair (42 tiles), nshifts=10."""
    feats = synth.random_cis_pairs(hg38, 1_000_000, seed=42, strands=True)
    modify = partial(coolpup.bin_distance_intervals, band_edges="default")
    pu, plan = _plan(hg38, feats, seed=0, flank=100_000, nshifts=10, groupby=["strand1", "strand2", "distance_band"],
                     modify=modify, cols=["distance"])
    assert plan["T"] >= 30
    acc = _run_calls(pu, plan, plan["calls"])
    # the grouped call (tile sets: slot and segment bits in the key) must stay on the hand-written binning (csrc/pup_bin.hpp) — the
    # library's radix sort is only for keys beyond 23 bits, and a change that silently pushed this shape there would cost 0.15 ms a call
    eng = coolpup._engine_for(pu._aclr, 0)
    assert eng.last_kernel() == "staged" and eng.last_prepass() == "binning", (eng.last_kernel(), eng.last_prepass())
    _check_every_window_vs_oracle(pu, plan, acc)
    _check_full_size(pu, plan, acc)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        df = coolpup.pileup(hg38, feats, features_format="bedpe", flank=100_000, nshifts=10, seed=0, by_distance=True,
                            by_strand=True)
    # "all" = sum of the groups, and the grouped run saw exactly the windows of the ungrouped one
    rows = df[df["group"].astype(str) != "all"]
    al = df[df["group"].astype(str) == "all"].iloc[0]
    assert int(al["n"]) == int(rows["n"].sum()) == int(acc["n"][:plan["G"]].sum())
    np.testing.assert_array_equal(np.asarray(al["num"]), np.sum([np.asarray(x) for x in rows["num"]], axis=0))
    assert int(al["control_n"]) == int(acc["n"][plan["G"]:].sum())


def test_config4_trans_pairs_pad25(hg38, oracle_mod):
    """configs[4]: 5e5 inter-chromosomal pairs over all chromosome-pair blocks, 51 x 51 windows."""
    feats = synth.random_trans_pairs(hg38, 500_000, seed=43)
    pu, plan = _plan(hg38, feats, flank=250_000, trans=True)
    assert plan["pad"] == 25
    acc = _run_calls(pu, plan, plan["calls"])
    assert int(acc["n"][0]) > 490_000
    eng = coolpup._engine_for(pu._aclr, 0)
    assert eng.last_kernel() == "sparse", eng.last_kernel()       # K1s served the calls ...
    _check_every_window_vs_oracle(pu, plan, acc, expect_staged=False)     # ... and EVERY window of them matches the oracle
    _check_full_size(pu, plan, acc)
    # the sparse trans kernel against the plain per-window kernel on everything
    os.environ["COOLPUPPY_AMD_VARIANT"] = "32"
    try:
        plain = _run_calls(pu, plan, plan["calls"])
    finally:
        os.environ.pop("COOLPUPPY_AMD_VARIANT")
        eng.set_tuning(0, 0)
    np.testing.assert_array_equal(plain["num"], acc["num"])
    np.testing.assert_allclose(plain["sum"], acc["sum"], rtol=1e-10, atol=0)


def test_dense_trans_table_sparse_kernel_every_window(hip_lib, oracle_mod):
    """K1s away from its sweet spot: a trans table with ~30 pixels per 51 x 51 window (the human-scale synthetic one holds
    ~3) — its O(W) lookup then finds a pixel in most rows it probes.  Every window against the oracle, K1s asserted, and the
    same integers as the dense per-window kernel."""
    sizes = {f"chr{k}": 60_000_000 for k in range(1, 7)}
    clr = synth.make_cooler(sizes, binsize=10_000, lam=200, seed=77, name="dense_trans", parallel=True, trans_nnz=8_000_000)
    feats = synth.random_trans_pairs(clr, 300_000, seed=5)
    pu, plan = _plan(clr, feats, flank=250_000, trans=True)
    acc = _run_calls(pu, plan, plan["calls"])
    eng = coolpup._engine_for(pu._aclr, 0)
    assert eng.last_kernel() == "sparse", eng.last_kernel()
    st = eng.stats()
    _check_every_window_vs_oracle(pu, plan, acc, expect_staged=False)
    # pixels per window of this table (counted by the statistics pass of the plain kernel)
    eng.set_tuning(0, 32); eng.set_profiling(1); eng.clear_stats()
    plain = _run_calls(pu, plan, plan["calls"])
    per_window = eng.stats()["pixels_in_windows"] / max(int(plain["n"].sum()), 1)
    eng.set_profiling(0); eng.set_tuning(0, 0)
    assert per_window >= 25, per_window
    np.testing.assert_array_equal(plain["num"], acc["num"])
    np.testing.assert_allclose(plain["sum"], acc["sum"], rtol=1e-10, atol=0)
    del st


@pytest.mark.parametrize("case", ["controls_flip", "expected_ooe", "expected_as_control"])
@pytest.mark.parametrize("flank", [1_270_000, 2_000_000])
def test_wide_windows_through_pileup_full_oracle(hip_lib, monkeypatch, oracle_mod, case, flank):
    """Windows of 255 and 401 bins through the PUBLIC entry point with everything that rides on them: random-shift controls
    and strand flips, observed over expected, expected as control (coolpuppy/coolpup.py:1115-1157, 1127-1139, 999-1005) — the same
    pileup() call with the engine half replaced by the oracle replay must give the same frame."""
    clr = synth.make_cooler({"chrA": 60_000_000, "chrB": 45_000_000}, binsize=10_000, lam=150, seed=31, name="wide_api")
    feats = synth.random_cis_pairs(clr, 400, min_sep=2 * flank + 200_000, max_sep=2 * flank + 3_000_000, seed=9, strands=True)
    kw = dict(features_format="bedpe", flank=flank)
    if case == "controls_flip":
        kw.update(nshifts=2, seed=4, flip_negative_strand=True, maxshift=2_000_000)
    else:
        kw.update(expected_df=synth.cis_expected(clr), ooe=(case == "expected_ooe"), nshifts=0)
    gpu, cpu = _both(clr, feats, monkeypatch, **kw)
    _frames_equal(gpu, cpu)
    assert np.asarray(gpu["data"].iloc[0]).shape == (2 * (flank // 10_000) + 1,) * 2
    assert int(gpu["n"].iloc[0]) > 100
    if case != "expected_ooe":
        np.testing.assert_array_equal(np.asarray(gpu["control_num"].iloc[0]), np.asarray(cpu["control_num"].iloc[0]))
