"""K3 (pup_coverage) vs the numpy restatement of cooltools' coverage — exact (integer sums)."""
import numpy as np
import pytest

from coolpuppy_amd import coolpup
import synth

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


@pytest.mark.parametrize("ignore_diags", [0, 1, 2, 3])
def test_coverage_known_answers(hip_lib, ignore_diags):
    """K3 against hand-derived numbers (tests/coverage_kat.py), not against the package's own numpy restatement."""
    import coverage_kat as kat
    from coolpuppy_amd.engine import PileupEngine
    indptr, col, cnt = kat.table()
    with PileupEngine(0) as eng:
        eng.load_pixels(indptr, col, cnt)
        cis, tot = eng.coverage(kat.CHROM_OFFSET, ignore_diags=ignore_diags)
    want_cis, want_tot = kat.ANSWERS[ignore_diags]
    np.testing.assert_array_equal(cis, np.array(want_cis, float))
    np.testing.assert_array_equal(tot, np.array(want_tot, float))


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
    import synth
    from coolpuppy_amd.engine import PileupEngine
    from coolpuppy_amd import _ffi
    rccl = _ffi.rccl()            # the librccl beside the HIP runtime of this process (pup_rccl_path): one ROCm stack

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


def test_tile_blocks_and_single_rank_allgather(hip_lib):
    """pup_pack_tiles / pup_unpack_tiles (the by-window exchange's two halves) and pup_allgather_tiles over a one-rank RCCL
    communicator: a packed block holds the listed tiles in the documented layout, unpack overwrites / adds / clears exactly those
    tiles, and the all-gather of a rank's own tiles leaves every accumulator as it was (pack, clear, broadcast to itself, add)."""
    import ctypes as C
    import torch
    from coolpuppy_amd.engine import PileupEngine, PupError
    from coolpuppy_amd import _ffi
    clr = synth.make_cooler({"chrA": 8_000_000, "chrB": 5_000_000}, lam=60, seed=3)
    eng = PileupEngine(0)
    eng.load_pixels(*clr.pixel_table())
    eng.load_bins(clr.bins()["weight"][:].values, clr.bins()["cov_tot_raw"][:].values)
    rng = np.random.default_rng(0)
    T, pad, W = 7, 4, 9
    n = 3000
    r0 = rng.integers(0, 700, n).astype(np.int32)
    c0 = (r0 + rng.integers(0, 60, n)).astype(np.int32)
    tp = np.array([0, 400, 400, 900, 1500, 1500, 2200, n])           # tiles 1 and 4 stay empty
    eng.reset(T, pad)
    eng.accumulate(r0, c0, tp, ignore_diags=2, mode=4)                # (coverage vectors ride along: the cov part of a block)
    before = eng.fetch()
    ids = np.array([5, 0, 3], np.int32)
    nf, ni = eng.tile_block_sizes(len(ids))
    assert (nf, ni) == (3 * (W * W + 2 * W), 3 * (W * W + 1))
    bf = torch.zeros(nf, dtype=torch.float64, device="cuda:0")
    bi = torch.zeros(ni, dtype=torch.int64, device="cuda:0")
    eng.pack_tiles(ids, bf.data_ptr(), bi.data_ptr())
    hf, hi = bf.cpu().numpy().reshape(3, -1), bi.cpu().numpy().reshape(3, -1)
    for k, t in enumerate(ids):
        assert np.array_equal(hf[k, :W * W], before["sum"][t].ravel()) and np.array_equal(hf[k, W * W:W * W + W], before["cov_start"][t])
        assert np.array_equal(hf[k, W * W + W:], before["cov_end"][t])
        assert np.array_equal(hi[k, :W * W], before["num"][t].ravel()) and hi[k, W * W] == before["n"][t]
    eng.unpack_tiles(ids, bf.data_ptr(), bi.data_ptr(), mode=1)       # add: the listed tiles double, the others stay
    twice = eng.fetch()
    for t in range(T):
        f = 2 if t in ids else 1
        assert np.array_equal(twice["sum"][t], f * before["sum"][t]) and np.array_equal(twice["num"][t], f * before["num"][t])
        assert twice["n"][t] == f * before["n"][t] and np.array_equal(twice["cov_end"][t], f * before["cov_end"][t])
    eng.unpack_tiles(ids[:2], mode=2)                                  # clear two of them
    eng.unpack_tiles(ids[2:], bf[2 * (W * W + 2 * W):].data_ptr(), bi[2 * (W * W + 1):].data_ptr(), mode=0)   # overwrite the third
    got = eng.fetch()
    assert not got["sum"][5].any() and not got["num"][0].any() and got["n"][5] == 0
    assert np.array_equal(got["sum"][3], before["sum"][3]) and np.array_equal(got["sum"][6], before["sum"][6])
    with pytest.raises(PupError):
        eng.pack_tiles(np.array([T], np.int32), bf.data_ptr(), bi.data_ptr())
    # the collective itself, one rank
    rccl = _ffi.rccl()

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
    eng.reset(T, pad)
    eng.accumulate(r0, c0, tp, ignore_diags=2, mode=4)
    mine = np.flatnonzero(np.diff(tp) > 0).astype(np.int32)
    eng.allgather_tiles(comm, mine, np.array([0, len(mine)], np.int64), 0)
    after = eng.fetch()
    for k in before:
        np.testing.assert_array_equal(before[k], after[k])
    eng.close()
    rccl.ncclCommDestroy.argtypes = [C.c_void_p]
    rccl.ncclCommDestroy(comm)


FRESH = r"""
import ctypes as C, sys
sys.path.insert(0, {root!r})
import numpy as np
from coolpuppy_amd import _ffi
from coolpuppy_amd.engine import PileupEngine
import synth
rccl = _ffi.rccl()
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId()
assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p()
rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
eng = PileupEngine(0)
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
clr = synth.make_cooler({{"chrA": 8_000_000}}, lam=60, seed=3)
eng.load_pixels(*clr.pixel_table()); eng.load_bins(clr.bins()["weight"][:].values, None)
eng.reset(1, 10)
eng.accumulate(np.arange(100, dtype=np.int32), np.arange(100, dtype=np.int32) + 30, np.array([0, 100]))
eng.allreduce(comm); eng.sync()
print("PATH", _ffi.rccl_path())
{tail}
print("FRESH OK")
"""


@pytest.mark.parametrize("tidy", [True, False])
def test_fresh_interpreter_with_rccl_exits_cleanly(hip_lib, tidy):
    """GPUTEST_r02 died AFTER its last test: the interpreter aborted at exit (rc 134) with the system ROCm's librccl
    loaded into a process running on torch's libamdhip64, and engines / communicators only released from __del__.  A
    fresh interpreter that does the single-rank RCCL case must exit 0 — whether the script tidies up itself or leaves
    everything to the package's atexit hook — and the librccl it used must sit beside the mapped HIP runtime."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    tail = ("rccl.ncclCommDestroy.argtypes = [C.c_void_p]; rccl.ncclCommDestroy(comm); eng.close()" if tidy
            else "import coolpuppy_amd.dist as D; D._NATIVE_COMMS[(0, 1)] = (comm.value, rccl)")
    code = FRESH.format(root=root, tail=tail)
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert res.returncode == 0, (res.returncode, res.stdout[-2000:], res.stderr[-2000:])
    assert "FRESH OK" in res.stdout
    path = [ln for ln in res.stdout.splitlines() if ln.startswith("PATH ")][0][5:]
    # the child is gone; this interpreter resolves the same way, and its own mappings can be inspected
    from coolpuppy_amd import _ffi
    assert path == _ffi.rccl_path()
    with open("/proc/self/maps") as f:
        hip = [ln.split()[-1] for ln in f if "libamdhip64" in ln]
    assert hip, "no HIP runtime mapped?"
    assert os.path.dirname(os.path.realpath(path)) == os.path.dirname(os.path.realpath(hip[0])) or not os.path.isabs(path)
