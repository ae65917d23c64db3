"""K3 (pup_coverage) vs the numpy restatement of cooltools' coverage — exact (integer sums)."""
import numpy as np
import pytest

from coolpuppy_amd import coolpup, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ignore_diags", [0, 2, 5])
def test_coverage_exact(hip_lib, oracle_mod, ignore_diags):
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chrA": 60_000_000, "chrB": 23_000_000, "chrC": 9_000_000}, lam=300, seed=4,
                            trans_nnz=200_000)
    indptr, col, cnt = clr.pixel_table()
    want_cis, want_tot = oracle_mod.coverage_numpy(indptr, col, cnt, clr.chrom_offset, ignore_diags)
    with PileupEngine(0) as eng:
        eng.load_pixels(indptr, col, cnt)
        cis, tot = eng.coverage(clr.chrom_offset, ignore_diags=ignore_diags)
    np.testing.assert_array_equal(cis, want_cis)
    np.testing.assert_array_equal(tot, want_tot)
    if ignore_diags == 0:   # the generator's own columns use the same convention
        np.testing.assert_array_equal(tot, clr.bins()["cov_tot_raw"][:].values)
        np.testing.assert_array_equal(cis, clr.bins()["cov_cis_raw"][:].values)


def test_pileup_computes_missing_coverage_column(hip_lib, oracle_mod):
    """coverage_norm=True on a cooler without cov_tot_raw: the column is computed (K3) and stored, and the
    pile-up equals the one obtained with the column supplied up front."""
    from coolpuppy_amd.cooler_lite import ArrayCooler
    full = synth.make_cooler({"chrA": 30_000_000, "chrB": 12_000_000}, lam=80, seed=9)
    indptr, col, cnt = full.pixel_table()
    _, tot = oracle_mod.coverage_numpy(indptr, col, cnt, full.chrom_offset, 2)
    bare = ArrayCooler(full.chromsizes, full.binsize, indptr, col, cnt, bins={"weight": full.bins()["weight"][:].values},
                       filename="bare.cool")
    given = ArrayCooler(full.chromsizes, full.binsize, indptr, col, cnt, bins={"cov_tot_raw": tot}, filename="given.cool")
    pairs = synth.random_cis_pairs(full, 3000, min_sep=230_000, max_sep=2_000_000, seed=2)
    kw = dict(features_format="bedpe", flank=100_000, clr_weight_name=None, coverage_norm=True)
    a = coolpup.pileup(bare, pairs, **kw)
    b = coolpup.pileup(given, pairs, **kw)
    assert "cov_tot_raw" in bare.bins().columns
    np.testing.assert_allclose(a["data"].iloc[0], b["data"].iloc[0], rtol=1e-12, equal_nan=True)
    np.testing.assert_array_equal(a["num"].iloc[0], b["num"].iloc[0])


def test_native_rccl_allreduce_single_rank(hip_lib):
    """pup_allreduce with a one-rank RCCL communicator (all the box offers): the call path through dlopen'd librccl
    runs on the engine's stream and leaves the accumulators as they were (sum over one rank)."""
    import ctypes as C
    import numpy as np
    from coolpuppy_amd import synth
    from coolpuppy_amd.engine import PileupEngine
    try:
        rccl = C.CDLL("librccl.so")
    except OSError:
        rccl = C.CDLL("/opt/rocm/lib/librccl.so")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    eng = PileupEngine(0)                                       # binds device 0 before the communicator is created
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    clr = synth.make_cooler({"chrA": 8_000_000, "chrB": 5_000_000}, lam=60, seed=3)
    eng.load_pixels(*clr.pixel_table())
    eng.load_bins(clr.bins()["weight"][:].values, None)
    rng = np.random.default_rng(0)
    r0 = rng.integers(0, 700, 500).astype(np.int32)
    c0 = (r0 + rng.integers(0, 60, 500)).astype(np.int32)
    eng.reset(2, 10)
    eng.accumulate(r0, c0, np.array([0, 200, 500]), ignore_diags=2, mode=0)
    before = eng.fetch()
    eng.allreduce(comm)
    after = eng.fetch()
    for k in before:
        np.testing.assert_array_equal(before[k], after[k])
    eng.close()
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)
