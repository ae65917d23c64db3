"""The libhdf5/ctypes .cool reader against a file written by an INDEPENDENT writer (h5py of the image's conda
Python, in a subprocess) with the cooler schema; skipped when that interpreter is not available."""
import os
import subprocess

import numpy as np
import pytest

import synth

CONDA_PY = "/opt/conda/bin/python3.9"

WRITER = r'''
import sys, numpy as np, h5py
z = np.load(sys.argv[1], allow_pickle=False)
with h5py.File(sys.argv[2], "w") as f:
    g = f.create_group(sys.argv[3]) if sys.argv[3] != "/" else f
    g.attrs["bin-size"] = int(z["binsize"]); g.attrs["format"] = "HDF5::Cooler"; g.attrs["nbins"] = int(len(z["weight"]))
    g.create_dataset("chroms/name", data=np.array([s.encode() for s in z["names"]], dtype="S32"))
    g.create_dataset("chroms/length", data=z["lengths"].astype(np.int32))
    enum = h5py.special_dtype(enum=("i4", {n: i for i, n in enumerate(z["names"])}))
    g.create_dataset("bins/chrom", data=z["chrom_id"].astype("i4"), dtype=enum)
    g.create_dataset("bins/start", data=z["start"].astype(np.int32)); g.create_dataset("bins/end", data=z["end"].astype(np.int32))
    g.create_dataset("bins/weight", data=z["weight"], compression="gzip")
    g.create_dataset("bins/cov_tot_raw", data=z["cov"])
    g.create_dataset("pixels/bin1_id", data=z["bin1_id"], compression="gzip", chunks=(4096,))
    g.create_dataset("pixels/bin2_id", data=z["bin2_id"].astype(np.int64), compression="gzip", chunks=(4096,))
    g.create_dataset("pixels/count", data=z["count"].astype(np.int32), compression="gzip", chunks=(4096,))
    g.create_dataset("indexes/bin1_offset", data=z["bin1_offset"]); g.create_dataset("indexes/chrom_offset", data=z["chrom_offset"])
'''


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="no independent HDF5 writer in this image")
@pytest.mark.parametrize("group", ["/", "resolutions/10000"])
def test_read_cool_roundtrip(tmp_path, group, monkeypatch):
    try:
        subprocess.run([CONDA_PY, "-c", "import h5py"], check=True, capture_output=True)
    except Exception:
        pytest.skip("conda python has no h5py")
    from coolpuppy_amd import cool_io
    clr = synth.make_cooler({"chr1": 9_000_000, "chr2": 6_500_000, "chrX": 3_000_000}, lam=30, seed=5, trans_nnz=2000)
    indptr, col, cnt = clr.pixel_table()
    npz = tmp_path / "in.npz"
    nb = np.diff(clr.chrom_offset)
    np.savez(npz, names=np.array(clr.chromnames), lengths=clr.chromsizes.values, binsize=clr.binsize,
             chrom_id=np.repeat(np.arange(3), nb), start=clr.bins()["start"][:].values, end=clr.bins()["end"][:].values,
             weight=clr.bins()["weight"][:].values, cov=clr.bins()["cov_tot_raw"][:].values,
             bin1_id=np.repeat(np.arange(clr.nbins), np.diff(indptr)), bin2_id=col, count=cnt, bin1_offset=indptr,
             chrom_offset=clr.chrom_offset)
    path = tmp_path / "test.cool"
    script = tmp_path / "w.py"
    script.write_text(WRITER)
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    subprocess.run([CONDA_PY, str(script), str(npz), str(path), group], check=True, env=env)
    import builtins
    real_import = builtins.__import__

    def no_h5py(name, *a, **k):                 # force the ctypes path even if h5py were importable
        if name == "h5py":
            raise ImportError("forced")
        return real_import(name, *a, **k)
    monkeypatch.setattr(builtins, "__import__", no_h5py)
    got = cool_io.read_cool(str(path), group=group)
    assert got.binsize == clr.binsize and got.chromnames == clr.chromnames
    np.testing.assert_array_equal(got.chromsizes.values, clr.chromsizes.values)
    np.testing.assert_array_equal(got.bin1_offset, indptr)
    np.testing.assert_array_equal(got.bin2_id, col)
    np.testing.assert_array_equal(got.count, cnt)
    np.testing.assert_array_equal(got.bins()["weight"][:].values, clr.bins()["weight"][:].values)
    np.testing.assert_array_equal(got.bins()["cov_tot_raw"][:].values, clr.bins()["cov_tot_raw"][:].values)
    np.testing.assert_array_equal(got.chrom_offset, clr.chrom_offset)
    assert got.extent(("chr2", 0, 6_500_000)) == clr.extent(("chr2", 0, 6_500_000))


H5PY_CLPY_READER = r'''
import sys, json, numpy as np, h5py
with h5py.File(sys.argv[1], "r") as f:
    d = f["data"]
    rep = {"dtype": str(d.dtype), "shape": list(d.shape), "chunks": list(d.chunks), "compression": d.compression,
           "attrs": {k: (v if isinstance(v, str) else np.asarray(v).tolist()) for k, v in f["attrs"].attrs.items()},
           "groups": sorted(f.keys()),
           "manifest": [{"name": x.decode()} for x in f["annotation/axis0"][...]],
           "hs_format": f["horizontal_stripe_0"].attrs["h5sparse_format"],
           "hs_shape": np.asarray(f["horizontal_stripe_0"].attrs["h5sparse_shape"]).tolist(),
           "coord0": [x.decode() if isinstance(x, bytes) else x for x in f["coordinates_0"][0]]}
    np.savez(sys.argv[2], data=d[...], hs_data=f["horizontal_stripe_0/data"][...], hs_indices=f["horizontal_stripe_0/indices"][...],
             hs_indptr=f["horizontal_stripe_0/indptr"][...])
    print(json.dumps(rep))
'''


def _stripes_frame(monkeypatch):
    import golden_util as gu
    from coolpuppy_amd import coolpup
    monkeypatch.setattr(coolpup.PileUpper, "run_plan", gu.oracle_run_plan)
    return gu.run("G11_stripes_raw", coolpup.pileup)[1]


def test_clpy_roundtrip(tmp_path, monkeypatch):
    """.clpy written through libhdf5/ctypes and read back by the own reader: every column of the frame survives —
    data as float32, stripes with their NaNs and zeros, coordinates, annotation columns of every kind, metadata."""
    from coolpuppy_amd.lib import io as pio
    df = _stripes_frame(monkeypatch)
    df["score"] = np.arange(len(df)) * 0.5
    df["flag"] = [bool(i % 2) for i in range(len(df))]
    df["pair"] = [(i, "x") for i in range(len(df))]
    path = str(tmp_path / "out.clpy")
    pio.save_pileup_df(path, df, metadata={"features": "x.bed", "view_file": None, "nshifts": 0, "expected": False, "run_id": 40000})
    back = pio.load_pileup_df(path)
    assert len(back) == len(df)
    assert back["features"].iloc[0] == "x.bed" and not back["view_file"].iloc[0] and back["run_id"].iloc[0] == 40000
    for i in range(len(df)):
        np.testing.assert_array_equal(back["data"].iloc[i], df["data"].iloc[i].astype(np.float32))
        np.testing.assert_array_equal(back["horizontal_stripe"].iloc[i], np.asarray(df["horizontal_stripe"].iloc[i], float))
        np.testing.assert_array_equal(back["vertical_stripe"].iloc[i], np.asarray(df["vertical_stripe"].iloc[i], float))
        np.testing.assert_array_equal(back["coordinates"].iloc[i], np.asarray(df["coordinates"].iloc[i]).astype("U13"))
        np.testing.assert_array_equal(back["num"].iloc[i], df["num"].iloc[i])
    for col in df.columns:
        if col not in ("data", "horizontal_stripe", "vertical_stripe", "coordinates", "num"):
            assert list(back[col]) == list(df[col]) or np.array_equal(back[col].values, df[col].values, equal_nan=True), col
    assert back["pair"].iloc[0] == (0, "x") and back["flag"].dtype == bool
    lst = pio.load_pileup_df_list([path, path], skipstripes=True)
    assert len(lst) == 2 * len(df) and set(lst["norm"]) == {"none"} and "horizontal_stripe" not in lst.columns
    with pytest.raises(ValueError):
        pio.save_pileup_df(path, df, compression="lzf")
    # the pickle-free column layout of round 2 stays available and reads back the same
    path2 = str(tmp_path / "cols.clpy")
    pio.save_pileup_df(path2, df, metadata={"nshifts": 0, "expected": False}, layout="columns")
    back2 = pio.load_pileup_df(path2)
    assert list(back2["pair"]) == list(df["pair"]) and back2["flag"].dtype == bool
    for i in range(len(df)):
        np.testing.assert_array_equal(back2["num"].iloc[i], df["num"].iloc[i])
    # mode="a" on a file that does not exist yet creates it; list-valued and None metadata survive
    path3 = str(tmp_path / "new.clpy")
    pio.save_pileup_df(path3, df, metadata={"names": ["a", "b"], "view_file": None}, mode="a")
    back3 = pio.load_pileup_df(path3, skipstripes=True)
    assert back3["view_file"].iloc[0] is np.False_ or back3["view_file"].iloc[0] is False


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="needs the image's conda python with h5py")
def test_clpy_read_by_h5py(tmp_path, monkeypatch):
    """The same file opened by an independent HDF5 stack (h5py in the conda interpreter): /data float32 [(rows*W), W]
    in (W, W) gzip chunks, /attrs with version, h5sparse CSR stripe groups, string coordinates, annotation manifest."""
    import json
    from coolpuppy_amd import __version__
    from coolpuppy_amd.lib import io as pio
    df = _stripes_frame(monkeypatch)
    path, dump = str(tmp_path / "out.clpy"), str(tmp_path / "dump.npz")
    pio.save_pileup_df(path, df, metadata={"features": "x.bed", "view_file": None, "run_id": 40000})
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([CONDA_PY, "-c", H5PY_CLPY_READER, path, dump], capture_output=True, text=True, env=env, timeout=300)
    if r.returncode != 0 and "No module named" in r.stderr:
        pytest.skip("h5py not importable in the conda interpreter")
    assert r.returncode == 0, r.stderr
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    w = df["data"].iloc[0].shape[0]
    assert rep["dtype"] == "float32" and rep["shape"] == [w * len(df), w] and rep["chunks"] == [w, w] and rep["compression"] == "gzip"
    assert rep["attrs"]["version"] == __version__ and rep["attrs"]["features"] == "x.bed" and rep["attrs"]["view_file"] in (0, False)
    assert rep["attrs"]["run_id"] == 40000
    assert {"annotation", "attrs", "data", "coordinates_0", "horizontal_stripe_0", "vertical_stripe_0"} <= set(rep["groups"])
    assert rep["hs_format"] == "csr" and rep["hs_shape"] == list(np.asarray(df["horizontal_stripe"].iloc[0]).shape)
    assert rep["coord0"] == [str(x) for x in np.asarray(df["coordinates"].iloc[0])[0]]
    assert [c["name"] for c in rep["manifest"]] == [c for c in df.columns if c not in ("data", "horizontal_stripe", "vertical_stripe", "coordinates")]
    z = np.load(dump)
    np.testing.assert_array_equal(z["data"][:w], df["data"].iloc[0].astype(np.float32))
    from scipy import sparse
    hs = sparse.csr_matrix((z["hs_data"], z["hs_indices"], z["hs_indptr"]), shape=rep["hs_shape"]).toarray()
    np.testing.assert_array_equal(hs, np.asarray(df["horizontal_stripe"].iloc[0], float))


# the frame both interpreters build: every kind of column the pile-up output holds (ints, floats with NaN, bools, strings,
# tuples, per-row integer arrays)
FRAME_SRC = r"""
import numpy as np, pandas as pd
def frame():
    n = 4
    return pd.DataFrame({
        "group": [("+", "-"), ("-", "+"), "all", (0.0, 50000.0)], "orientation": ["+-", "-+", "all", "+-"],
        "n": np.array([10, 20, 30, 40], np.int64), "flank": np.int64(100000), "resolution": np.int64(10000),
        "score": [1.5, np.nan, 0.25, 4.0], "expected": False, "store_stripes": False, "flag": [False, True, False, False],
        "num": [np.arange(9).reshape(3, 3) * (i + 1) for i in range(n)], "name": ["a", "bb", "ccc", ""],
        "cov_start": [np.linspace(0, 1, 3) for _ in range(n)],
    })
"""

PANDAS_COMPARE = r"""
import sys, pickle, json, warnings
warnings.simplefilter("ignore")
import numpy
numpy.typeDict = numpy.sctypeDict            # PyTables 3.6 of this conda env predates numpy 1.24 (import-time alias only)
import tables, pandas, h5py
import pandas.compat._optional as opt
opt.VERSIONS["tables"] = tables.__version__   # pandas' minimum-version gate; the writer calls are the same
exec(open(sys.argv[3]).read())
ref = sys.argv[2]
frame().to_hdf(ref, key="annotation", mode="w")
def dump(path):
    out = {}
    with h5py.File(path, "r") as f:
        def node(name, obj):
            at = {}
            for k in obj.attrs:
                aid = h5py.h5a.open(obj.id, k.encode())
                t, sp = aid.get_type(), aid.get_space()
                v = obj.attrs[k]
                at[k] = [int(t.get_class()), int(t.get_size()), int(sp.get_simple_extent_type()),
                         None if isinstance(v, h5py.Empty) else numpy.asarray(v).tolist()]
            rec = {"attrs": at}
            if isinstance(obj, h5py.Dataset):
                t = obj.id.get_type()
                rec.update(shape=list(obj.shape), tclass=int(t.get_class()), tsize=int(t.get_size()), maxshape=[m for m in obj.maxshape])
                if obj.dtype == object:
                    arr = pickle.loads(obj[0].tobytes())
                    rec["pickled"] = [[repr(x) for x in row] for row in arr.tolist()] if arr.dtype == object else None
                    rec["pickled_shape"] = list(arr.shape)
                else:
                    rec["data"] = [x.decode() if isinstance(x, bytes) else x for x in numpy.asarray(obj[...]).ravel().tolist()]
            out[name] = rec
        node("annotation", f["annotation"])
        f["annotation"].visititems(lambda n, o: node("annotation/" + n, o))
        out["/"] = {"attrs": {k: (None if isinstance(f.attrs[k], h5py.Empty) else numpy.asarray(f.attrs[k]).tolist())
                              for k in ("CLASS", "VERSION", "TITLE", "PYTABLES_FORMAT_VERSION")}}
    return out
print(json.dumps({"mine": dump(sys.argv[1]), "pandas": dump(ref)}, default=str))
"""


@pytest.mark.skipif(not os.path.exists(CONDA_PY), reason="needs the image's conda python (pandas + PyTables + h5py)")
def test_annotation_is_the_store_pandas_writes(tmp_path):
    """/annotation against the real thing: pandas.DataFrame.to_hdf(..., "annotation") — the call the reference makes
    (coolpuppy/lib/io.py:47-53) — run in the image's conda interpreter (pandas 2.3 + PyTables 3.6 + h5py), and both files
    dumped node by node with h5py: same nodes, same shapes and HDF5 type classes / sizes, same PyTables attributes (value,
    type class, size, dataspace kind), same data, and the pickled object blocks unpickle to the same cells.  Then this
    package's reader on pandas' own file."""
    import json
    from coolpuppy_amd.lib import io as pio
    src = tmp_path / "frame_src.py"
    src.write_text(FRAME_SRC)
    ns = {}
    exec(FRAME_SRC, ns)
    df = ns["frame"]()
    mine, ref = str(tmp_path / "mine.clpy"), str(tmp_path / "pandas.h5")
    full = df.copy()
    full["data"] = [np.eye(3) * i for i in range(len(df))]
    pio.save_pileup_df(mine, full, metadata={})
    env = {k: v for k, v in os.environ.items() if not k.startswith("PYTHON")}
    r = subprocess.run([CONDA_PY, "-c", PANDAS_COMPARE, mine, ref, str(src)], capture_output=True, text=True, env=env, timeout=600)
    if r.returncode != 0 and ("No module named" in r.stderr or "ImportError" in r.stderr):
        pytest.skip("pandas + PyTables + h5py not usable in the conda interpreter: " + r.stderr[-300:])
    assert r.returncode == 0, r.stderr[-3000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    a, b = rep["mine"], rep["pandas"]
    assert a["/"] == b["/"]
    assert a["annotation"]["attrs"] == b["annotation"]["attrs"]
    # pandas orders its blocks by its internal manager; match blocks through their item lists
    def blocks(d):
        out = {}
        for k, v in d.items():
            if k.endswith("_items"):
                out[tuple(v["data"])] = (v, d[k.replace("_items", "_values")])
        return out
    ba, bb = blocks(a), blocks(b)
    assert set(ba) == set(bb), (sorted(ba), sorted(bb))
    for key in bb:
        for x, y in zip(ba[key], bb[key]):
            assert x == y, (key, x, y)
    for k in ("annotation/axis0", "annotation/axis1"):
        assert a[k] == b[k], k
    # and the other direction: the file pandas wrote, through this package's reader
    with pio._H5(ref, "r") as h5:
        back = pio._read_annotation_pytables(h5)
    assert list(back.columns) == list(df.columns) and len(back) == len(df)
    for c in df.columns:
        for i in range(len(df)):
            x, y = back[c].iloc[i], df[c].iloc[i]
            if isinstance(y, np.ndarray):
                np.testing.assert_array_equal(x, y)
            else:
                assert x == y or (isinstance(y, float) and np.isnan(y) and np.isnan(x)), (c, i, x, y)
    assert back["n"].dtype == np.int64 and back["flag"].dtype == bool and list(back["flag"]) == [False, True, False, False] and back["score"].dtype == np.float64


def test_clpy_object_blocks_are_read_with_a_restricted_unpickler():
    """The pickled object columns of a PyTables "fixed" store are data, not code: numpy arrays of plain values load, anything
    else in the stream is refused (an untrusted .clpy must not be able to run os.system through pickle.loads)."""
    import pickle
    from coolpuppy_amd.lib import io as pio
    arr = np.empty((2, 2), dtype=object)
    arr[:] = [["chr1", 5], [None, 2.5]]
    back = pio._restricted_loads(pickle.dumps(arr, protocol=2))
    assert back.shape == (2, 2) and back[0, 0] == "chr1" and back[1, 0] is None and back[1, 1] == 2.5

    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("true",))
    with pytest.raises(pickle.UnpicklingError):
        pio._restricted_loads(pickle.dumps(Evil()))


@pytest.mark.parametrize("group,gz", [("/", None), ("resolutions/10000", 4)])
def test_streamed_cooler_reads_row_range_chunks(tmp_path, group, gz):
    """read_cool(..., stream_pixels=True): everything but the pixel table is in memory, pixel chunks come out of the file by
    hyperslab reads straight into caller-provided (page-locked, on the GPU path) arrays — at any offset, for contiguous and
    for chunked + gzip datasets, and the lazily loaded whole arrays equal the eager reader's."""
    from coolpuppy_amd import cool_io
    clr = synth.make_cooler({"chr1": 9_000_000, "chr2": 6_500_000, "chrX": 3_000_000}, lam=30, seed=5, trans_nnz=2000)
    path = str(tmp_path / "t.cool")
    cool_io.write_cool(path, clr, group=group, chunks=None if gz is None else 4096, gzip=gz)
    eager = cool_io.read_cool(path, group=group)
    np.testing.assert_array_equal(eager.bin2_id, clr.pixel_table()[1])
    lazy = cool_io.read_cool(path, group=group, stream_pixels=True)
    assert isinstance(lazy, cool_io.StreamedCooler) and not lazy.pixels_in_memory
    assert lazy.nnz == clr.nnz and lazy.nbins == clr.nbins and lazy.chromnames == clr.chromnames
    np.testing.assert_array_equal(lazy.bin1_offset, clr.pixel_table()[0])
    np.testing.assert_array_equal(lazy.bins()["weight"][:].values, clr.bins()["weight"][:].values)
    f = cool_io._File(path)
    g = "" if group == "/" else "/" + group
    assert clr.nnz > 20000
    for first, m in ((0, 1), (0, 5000), (4095, 3), (1234, clr.nnz - 5000), (clr.nnz - 7, 7)):
        col = np.empty(m, np.int64); cnt = np.empty(m, np.int32)
        f.read_into(f"{g}/pixels/bin2_id", first, col)
        f.read_into(f"{g}/pixels/count", first, cnt)
        np.testing.assert_array_equal(col, clr.pixel_table()[1][first:first + m])
        np.testing.assert_array_equal(cnt, clr.pixel_table()[2][first:first + m])
    f.close()
    assert not lazy.pixels_in_memory
    np.testing.assert_array_equal(lazy.pixel_table()[1], clr.pixel_table()[1])      # whoever asks for the arrays gets them
    np.testing.assert_array_equal(lazy.pixel_table()[2], clr.pixel_table()[2])
    assert lazy.pixels_in_memory


@pytest.mark.parametrize("chunks,gz,shuffle", [(None, None, False), (4096, None, False), (4096, 4, False), (5000, 6, True)])
def test_direct_reader_equals_the_hyperslab_reads(tmp_path, chunks, gz, shuffle):
    """cool_io._DirectReader (the streamed loader's data path: pread / inflate / un-shuffle on several threads, no libhdf5 call per
    read) against the arrays themselves, for contiguous datasets and chunked ones with cooler's filters (gzip, shuffle + gzip), at
    offsets inside / across chunks, into arrays of the file's width and narrower ones; a range-violating narrow copy raises."""
    from coolpuppy_amd import cool_io
    clr = synth.make_cooler({"chr1": 9_000_000, "chr2": 6_500_000}, lam=30, seed=5)
    path = str(tmp_path / "d.cool")
    cool_io.write_cool(path, clr, chunks=chunks, gzip=gz, shuffle=shuffle)
    col, cnt = clr.pixel_table()[1], clr.pixel_table()[2]
    f = cool_io._File(path)
    lay = f.layout("/pixels/bin2_id")
    assert lay is not None and lay["kind"] == ("contiguous" if chunks is None else "chunked") and lay["n"] == clr.nnz
    if chunks is not None:
        assert lay["filters"] == ([2] if shuffle else []) + ([1] if gz else [])
    rc = cool_io._DirectReader(path, f, "/pixels/bin2_id", threads=4)
    rn = cool_io._DirectReader(path, f, "/pixels/count", threads=4)
    rc.PIECE = rn.PIECE = 4096                               # (several tasks per read even on this small table)
    try:
        for first, m in ((0, 1), (0, 5000), (4095, 3), (1234, clr.nnz - 5000), (clr.nnz - 7, 7), (0, clr.nnz)):
            a64, a32, c32 = np.empty(m, np.int64), np.empty(m, np.int32), np.empty(m, np.int32)
            rc.read_into(first, a64); rc.read_into(first, a32); rn.read_into(first, c32)
            np.testing.assert_array_equal(a64, col[first:first + m])
            np.testing.assert_array_equal(a32, col[first:first + m])
            np.testing.assert_array_equal(c32, cnt[first:first + m])
        with pytest.raises(OverflowError):
            rc.read_into(0, np.empty(clr.nnz, np.int8))
    finally:
        rc.close(); rn.close(); f.close()


@pytest.mark.gpu
def test_streamed_cooler_upload_matches_the_eager_one(tmp_path, hip_lib):
    """The pixel table streamed file -> page-locked slabs -> HBM (pup_load_pixels_stream, several slabs) gives the same engine
    state as the whole-array upload: identical pile-ups through pileup(), and the copy statistics are reported."""
    from coolpuppy_amd import cool_io, coolpup
    from coolpuppy_amd.engine import PileupEngine
    clr = synth.make_cooler({"chr1": 90_000_000, "chr2": 65_000_000}, lam=150, seed=8)
    path = str(tmp_path / "s.cool")
    cool_io.write_cool(path, clr, chunks=65536, gzip=1, shuffle=True)
    lazy = cool_io.read_cool(path, stream_pixels=True)
    pairs = synth.random_cis_pairs(clr, 20_000, seed=2)
    kw = dict(features_format="bedpe", flank=100_000, nshifts=2, seed=1)
    a = coolpup.pileup(clr, pairs, **kw)
    b = coolpup.pileup(lazy, pairs, **kw)
    assert not lazy.pixels_in_memory and lazy._last_stream_stats["h2d_bytes"] == clr.nnz * 12
    np.testing.assert_array_equal(np.asarray(a["num"].iloc[0]), np.asarray(b["num"].iloc[0]))
    np.testing.assert_array_equal(np.asarray(a["data"].iloc[0]), np.asarray(b["data"].iloc[0]))
    # small slabs: many trips through the two page-locked buffers
    eng = PileupEngine(0)
    st = cool_io.stream_pixels_into(eng, lazy, slab_pixels=100_003)
    assert st["h2d_bytes"] == clr.nnz * 12 and st["h2d_ms"] > 0
    eng.build_index(clr.chrom_offset)
    eng.load_bins(clr.bins()["weight"][:].values, None)
    ref = PileupEngine(0)
    ref.load_pixels(*clr.pixel_table()); ref.build_index(clr.chrom_offset); ref.load_bins(clr.bins()["weight"][:].values, None)
    r0 = np.arange(100, 8000, 3, dtype=np.int32); c0 = (r0 + 40).astype(np.int32)
    for e in (eng, ref):
        e.reset(1, 10); e.accumulate(r0, c0, np.array([0, len(r0)], np.int64))
    x, y = eng.fetch(), ref.fetch()
    np.testing.assert_array_equal(x["num"], y["num"]); np.testing.assert_array_equal(x["sum"], y["sum"])
    eng.close(); ref.close()


@pytest.mark.parametrize("chunks", [None, 4096])
def test_direct_reader_declines_a_file_with_a_user_block(tmp_path, chunks):
    """behind an HDF5 user block the chunk addresses of older libhdf5 builds are relative to the base address — a pread
    at them would fetch other bytes, silently for unfiltered chunks.  layout() says None for such a file and the reader takes
    libhdf5's own path: same arrays."""
    from coolpuppy_amd import cool_io
    clr = synth.make_cooler({"chr1": 9_000_000, "chr2": 6_500_000}, lam=30, seed=5)
    path = str(tmp_path / "u.cool")
    cool_io.write_cool(path, clr, chunks=chunks, userblock=1024)
    with open(path, "rb") as fh:
        assert fh.read(8) != b"\x89HDF\r\n\x1a\n" and fh.seek(1024) == 1024 and fh.read(8) == b"\x89HDF\r\n\x1a\n"
    col, cnt = clr.pixel_table()[1], clr.pixel_table()[2]
    f = cool_io._File(path)
    assert f._userblock() == 1024 and f.layout("/pixels/bin2_id") is None
    rc = cool_io._DirectReader(path, f, "/pixels/bin2_id", threads=2)
    rn = cool_io._DirectReader(path, f, "/pixels/count", threads=2)
    try:
        for first, m in ((0, 5000), (4095, 3), (0, clr.nnz)):
            a64, c32 = np.empty(m, np.int64), np.empty(m, np.int32)
            rc.read_into(first, a64); rn.read_into(first, c32)
            np.testing.assert_array_equal(a64, col[first:first + m])
            np.testing.assert_array_equal(c32, cnt[first:first + m])
    finally:
        rc.close(); rn.close(); f.close()
    back = cool_io.read_cool(path)
    np.testing.assert_array_equal(back.pixel_table()[1], col)
