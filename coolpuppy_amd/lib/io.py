"""`.clpy` files: the reference's HDF5 output layout (reference coolpuppy/lib/io.py:18-190), kept as it is so that
files written here open with the reference's tools (plotpuppy, load_pileup_df) and the other way round:

    /annotation            pandas "fixed" HDF store (PyTables) of every column except the arrays below
    /data                  float32 [(rows * W), W], chunks (W, W), compressed (default lzf): one W x W pile-up per row
    /vertical_stripe_<i>, /horizontal_stripe_<i>   h5sparse CSR of the i-th row's stripes   (when store_stripes)
    /coordinates_<i>       object array [n, 6]
    /attrs                 group whose attributes hold the metadata dict (None -> False) and the writer's version

Host-side and off the pile-up path; needs the packages the reference needs for it — h5sparse (h5py) and PyTables.
They are not part of this image, so this module is exercised only where they are installed
(tests/test_cool_io.py::test_clpy_roundtrip skips otherwise).
"""
import os
import re

import numpy as np
import pandas as pd

from .. import __version__

_ARRAY_COLUMNS = ["data", "vertical_stripe", "horizontal_stripe", "coordinates"]


def _deps():
    try:
        import h5sparse
        import tables  # noqa: F401  (pandas.to_hdf / read_hdf)
        from scipy import sparse
    except ImportError as e:       # loud: there is no alternative writer
        raise ImportError(f"reading / writing .clpy files needs h5sparse and PyTables ({e})") from e
    return h5sparse, sparse


def save_pileup_df(filename, df, metadata=None, mode="w", compression="lzf"):
    """Write a pile-up DataFrame (the output of pileup()) plus a metadata dict to `filename` (:18-95)."""
    h5sparse, sparse = _deps()
    metadata = {} if metadata is None else metadata
    df[[c for c in df.columns if c not in _ARRAY_COLUMNS]].to_hdf(filename, "annotation", mode=mode)
    with h5sparse.File(filename, "a") as f:
        width = df["data"].iloc[0].shape[0]
        ds = f.create_dataset("data", compression=compression, chunks=(width, width), shape=(width * df["data"].shape[0], width))
        for i, arr in df["data"].reset_index(drop=True).items():
            ds[i * width:(i + 1) * width, :] = arr
        if df["store_stripes"].any():
            for name in ("vertical_stripe", "horizontal_stripe"):
                for i, arr in df[name].reset_index(drop=True).items():
                    f.create_dataset(f"{name}_{i}", compression=compression, shape=(len(arr), width),
                                     data=sparse.csr_matrix(arr))
            for i, arr in df["coordinates"].reset_index(drop=True).items():
                f.create_dataset(f"coordinates_{i}", compression=compression, shape=(len(arr), 6), data=arr.astype(object))
        group = f.create_group("attrs")
        for key, val in metadata.items():
            group.attrs[key] = False if val is None else val
        group.attrs["version"] = __version__


def load_pileup_df(filename, quaich=False, skipstripes=False):
    """Read a file written by save_pileup_df (here or by the reference) back into a DataFrame (:98-155)."""
    h5sparse, _ = _deps()
    with h5sparse.File(filename, "r", libver="latest") as f:
        metadata = dict(zip(f["attrs"].attrs.keys(), f["attrs"].attrs.values()))
        dstore = f["data"]
        data = [dstore[chunk] for chunk in dstore.iter_chunks()]
        annotation = pd.read_hdf(filename, "annotation")
        annotation["data"] = data
        if not skipstripes:
            try:
                cols = {"vertical_stripe": [], "horizontal_stripe": [], "coordinates": []}
                for i in range(len(data)):
                    cols["vertical_stripe"].append(f[f"vertical_stripe_{i}"][:].toarray())
                    cols["horizontal_stripe"].append(f[f"horizontal_stripe_{i}"][:].toarray())
                    cols["coordinates"].append(f[f"coordinates_{i}"][:].astype("U13"))
                for k, v in cols.items():
                    annotation[k] = v
            except KeyError:
                pass
    for key, val in metadata.items():
        if key != "version":
            annotation[key] = val
    if quaich:
        sample, bedname = re.search(r"^(.*)-(?:[0-9]+)_over_(.*)_(?:[0-9]+-shifts|expected).*\.clpy",
                                    os.path.basename(filename)).groups()
        annotation["sample"] = sample
        annotation["bedname"] = bedname
    return annotation


def load_pileup_df_list(files, quaich=False, nice_metadata=True, skipstripes=False):
    """Concatenate several .clpy files; nice_metadata adds a 'norm' column: expected / shifts / none (:158-190)."""
    pups = pd.concat([load_pileup_df(p, quaich=quaich, skipstripes=skipstripes) for p in files]).reset_index(drop=True)
    if nice_metadata:
        pups["norm"] = np.where(pups["expected"], ["expected"] * pups.shape[0], ["shifts"] * pups.shape[0]).astype(str)
        pups.loc[np.logical_not(np.logical_or(pups["nshifts"] > 0, pups["expected"])), "norm"] = "none"
    return pups
