"""The libhdf5/ctypes .cool reader against a file written by an INDEPENDENT writer (h5py of the image's conda
Python, in a subprocess) with the cooler schema; skipped when that interpreter is not available."""
import os
import subprocess

import numpy as np
import pytest

from coolpuppy_amd import synth

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


def test_clpy_roundtrip(tmp_path, monkeypatch):
    """.clpy writer / reader in the reference's layout — runs only where h5sparse and PyTables are installed."""
    pytest.importorskip("h5sparse")
    pytest.importorskip("tables")
    import golden_util as gu
    from coolpuppy_amd import coolpup
    from coolpuppy_amd.lib import io as pio
    monkeypatch.setattr(coolpup.PileUpper, "run_plan", gu.oracle_run_plan)
    z, df = gu.run("G11_stripes_raw", coolpup.pileup)
    path = str(tmp_path / "out.clpy")
    pio.save_pileup_df(path, df, metadata={"features": "x.bed", "view_file": None})
    back = pio.load_pileup_df(path)
    assert len(back) == len(df) and back["features"].iloc[0] == "x.bed"
    np.testing.assert_allclose(back["data"].iloc[0], df["data"].iloc[0].astype(np.float32), rtol=0, atol=0, equal_nan=True)
    np.testing.assert_allclose(back["horizontal_stripe"].iloc[0], np.nan_to_num(df["horizontal_stripe"].iloc[0]))
