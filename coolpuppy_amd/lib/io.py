"""`.clpy` files written and read straight through libhdf5 (ctypes) — no h5py, h5sparse or PyTables needed.

What the reference stores (coolpuppy/lib/io.py:18-95) and where it goes here:

    /data                  float32 [(rows * W), W], chunks (W, W), compressed: one W x W pile-up per output row — same
                           dataset, same dtype, same chunking; gzip instead of lzf (lzf is a filter h5py brings along, plain
                           libhdf5 has deflate; h5py reads either)
    /attrs                 group whose attributes hold the metadata dict (None -> False) and the writer's version — same
    /vertical_stripe_<i>, /horizontal_stripe_<i>
                           the i-th row's stripes as CSR in h5sparse's on-disk form: a group {data, indices, indptr} with
                           attributes h5sparse_format = "csr", h5sparse_shape = (n, W) — same
    /coordinates_<i>       variable-length strings [n, 6] — same
    /annotation            every other column.  The reference hands this to pandas.to_hdf (a PyTables "fixed" store, object
                           columns pickled).  PyTables is not part of this image, so that encoding cannot be produced or
                           checked here; the columns are stored instead as plain HDF5 datasets any reader opens:
                           /annotation/c<k> per column (numbers as they are, strings as variable-length UTF-8, everything
                           else — tuples, lists, arrays such as `num` — as one JSON text per row) and a JSON manifest of
                           names and kinds in the attribute `columns`.  `load_pileup_df` reads this layout; files whose
                           /annotation is a PyTables store (written by the reference) are read through pandas when
                           PyTables is importable.

Host-side, off the pile-up path.
"""
import ctypes as C
import json
import os
import re

import numpy as np
import pandas as pd

from .. import __version__
from ..cool_io import _hdf5, _native

_ARRAY_COLUMNS = ["data", "vertical_stripe", "horizontal_stripe", "coordinates"]
_H5F_ACC_TRUNC, _H5F_ACC_RDWR, _H5F_ACC_RDONLY = 2, 1, 0
_H5T_CSET_UTF8, _H5T_VARIABLE = 1, C.c_size_t(-1).value
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING = 0, 1, 3
_hid = C.c_int64


def _lib():
    lib = _hdf5()
    if getattr(lib, "_clpy_ready", False):
        return lib
    lib.H5Fcreate.restype = _hid; lib.H5Fcreate.argtypes = [C.c_char_p, C.c_uint, _hid, _hid]
    lib.H5Gcreate2.restype = _hid; lib.H5Gcreate2.argtypes = [_hid, C.c_char_p, _hid, _hid, _hid]
    lib.H5Screate_simple.restype = _hid; lib.H5Screate_simple.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
    lib.H5Screate.restype = _hid; lib.H5Screate.argtypes = [C.c_int]
    lib.H5Sget_simple_extent_ndims.restype = C.c_int; lib.H5Sget_simple_extent_ndims.argtypes = [_hid]
    lib.H5Sget_simple_extent_dims.restype = C.c_int; lib.H5Sget_simple_extent_dims.argtypes = [_hid, C.c_void_p, C.c_void_p]
    lib.H5Pcreate.restype = _hid; lib.H5Pcreate.argtypes = [_hid]
    lib.H5Pset_chunk.argtypes = [_hid, C.c_int, C.c_void_p]
    lib.H5Pset_deflate.argtypes = [_hid, C.c_uint]
    lib.H5Pclose.argtypes = [_hid]
    lib.H5Dcreate2.restype = _hid; lib.H5Dcreate2.argtypes = [_hid, C.c_char_p, _hid, _hid, _hid, _hid, _hid]
    lib.H5Dwrite.restype = C.c_int; lib.H5Dwrite.argtypes = [_hid, _hid, _hid, _hid, _hid, C.c_void_p]
    lib.H5Dvlen_reclaim.argtypes = [_hid, _hid, _hid, C.c_void_p]
    lib.H5Acreate2.restype = _hid; lib.H5Acreate2.argtypes = [_hid, C.c_char_p, _hid, _hid, _hid, _hid]
    lib.H5Awrite.restype = C.c_int; lib.H5Awrite.argtypes = [_hid, _hid, C.c_void_p]
    lib.H5Aget_type.restype = _hid; lib.H5Aget_type.argtypes = [_hid]
    lib.H5Aget_space.restype = _hid; lib.H5Aget_space.argtypes = [_hid]
    lib.H5Aget_num_attrs.restype = C.c_int; lib.H5Aget_num_attrs.argtypes = [_hid]
    lib.H5Aopen_by_idx.restype = _hid
    lib.H5Aopen_by_idx.argtypes = [_hid, C.c_char_p, C.c_int, C.c_int, C.c_uint64, _hid, _hid]
    lib.H5Aget_name.restype = C.c_ssize_t; lib.H5Aget_name.argtypes = [_hid, C.c_size_t, C.c_char_p]
    lib.H5Tset_size.argtypes = [_hid, C.c_size_t]
    lib.H5Tset_cset.argtypes = [_hid, C.c_int]
    lib.H5Oopen.restype = _hid; lib.H5Oopen.argtypes = [_hid, C.c_char_p, _hid]
    lib.H5Oclose.argtypes = [_hid]
    lib._clpy_ready = True
    return lib


_NP2H5 = {"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int64": "H5T_NATIVE_INT64_g",
          "int32": "H5T_NATIVE_INT32_g", "int8": "H5T_NATIVE_INT8_g", "uint8": "H5T_NATIVE_UINT8_g"}


class _H5:
    """The dozen libhdf5 calls a .clpy needs: groups, n-d numeric / string datasets, scalar and 1-d attributes."""

    def __init__(self, path, mode):
        self.lib = lib = _lib()
        p = os.fsencode(path)
        if mode == "w":
            self.fid = lib.H5Fcreate(p, _H5F_ACC_TRUNC, 0, 0)
        else:
            self.fid = lib.H5Fopen(p, _H5F_ACC_RDWR if mode == "a" else _H5F_ACC_RDONLY, 0)
        if self.fid < 0:
            raise OSError(f"cannot open {path!r} (mode {mode})")

    def close(self):
        if self.fid >= 0:
            self.lib.H5Fclose(self.fid)
            self.fid = -1

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- types -----------------------------------------------------------------------------------------------------
    def _vstr(self):
        t = self.lib.H5Tcopy(_native(self.lib, "H5T_C_S1_g"))
        self.lib.H5Tset_size(t, _H5T_VARIABLE)
        self.lib.H5Tset_cset(t, _H5T_CSET_UTF8)
        return t

    def _space(self, shape):
        if len(shape) == 0:
            return self.lib.H5Screate(0)                 # H5S_SCALAR
        dims = (C.c_uint64 * len(shape))(*shape)
        return self.lib.H5Screate_simple(len(shape), dims, None)

    # -- writing -----------------------------------------------------------------------------------------------------
    def group(self, name):
        g = self.lib.H5Gcreate2(self.fid, name.encode(), 0, 0, 0)
        if g < 0:
            raise OSError(f"cannot create group {name!r}")
        self.lib.H5Gclose(g)

    def write(self, name, arr, chunks=None, gzip=None):
        """Numeric array (dtype kept) or array of str (variable-length UTF-8), any rank."""
        lib = self.lib
        arr = np.asarray(arr)
        strings = arr.dtype.kind in "OUS"
        space = self._space(arr.shape)                       # before ascontiguousarray, which makes scalars 1-d
        dcpl = 0
        if chunks is not None and arr.size:
            dcpl = lib.H5Pcreate(_native(lib, "H5P_CLS_DATASET_CREATE_ID_g"))
            lib.H5Pset_chunk(dcpl, len(chunks), (C.c_uint64 * len(chunks))(*chunks))
            if gzip:
                lib.H5Pset_deflate(dcpl, int(gzip))
        if strings:
            t = self._vstr()
            enc = [str(x).encode("utf-8") for x in arr.ravel()]
            buf = (C.c_char_p * max(len(enc), 1))(*enc)
            did = lib.H5Dcreate2(self.fid, name.encode(), t, space, 0, dcpl, 0)
            ok = did >= 0 and (not enc or lib.H5Dwrite(did, t, 0, 0, 0, buf) >= 0)
            lib.H5Tclose(t)
        else:
            arr = np.ascontiguousarray(arr)
            if arr.dtype == bool:
                arr = arr.astype(np.int8)
            t = _native(lib, _NP2H5[arr.dtype.name])
            did = lib.H5Dcreate2(self.fid, name.encode(), t, space, 0, dcpl, 0)
            ok = did >= 0 and (arr.size == 0 or lib.H5Dwrite(did, t, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) >= 0)
        if did >= 0:
            lib.H5Dclose(did)
        lib.H5Sclose(space)
        if dcpl:
            lib.H5Pclose(dcpl)
        if not ok:
            raise OSError(f"cannot write dataset {name!r}")

    def set_attr(self, obj, name, value):
        """bool / int / float / str scalar, or a 1-d integer sequence, as an attribute of group or dataset `obj`."""
        lib = self.lib
        oid = lib.H5Oopen(self.fid, obj.encode(), 0)
        if oid < 0:
            raise KeyError(obj)
        try:
            if isinstance(value, (str, bytes, os.PathLike)):
                t = self._vstr()
                space = self._space(())
                p = C.c_char_p(os.fsdecode(value).encode("utf-8") if not isinstance(value, bytes) else value)
                aid = lib.H5Acreate2(oid, name.encode(), t, space, 0, 0)
                ok = aid >= 0 and lib.H5Awrite(aid, t, C.byref(p)) >= 0
                lib.H5Tclose(t)
            else:
                v = np.asarray(value)
                if v.dtype == bool:
                    v = v.astype(np.int8)
                elif v.dtype.kind in "iu":
                    v = v.astype(np.int64)
                elif v.dtype.kind == "f":
                    v = v.astype(np.float64)
                else:
                    raise TypeError(f"attribute {name!r}: unsupported value {value!r}")
                shape = v.shape                              # ascontiguousarray would turn a scalar into [1]
                v = np.ascontiguousarray(v)
                t = _native(lib, _NP2H5[v.dtype.name])
                space = self._space(shape)
                aid = lib.H5Acreate2(oid, name.encode(), t, space, 0, 0)
                ok = aid >= 0 and lib.H5Awrite(aid, t, v.ctypes.data_as(C.c_void_p)) >= 0
            if aid >= 0:
                lib.H5Aclose(aid)
            lib.H5Sclose(space)
            if not ok:
                raise OSError(f"cannot write attribute {name!r} of {obj!r}")
        finally:
            lib.H5Oclose(oid)

    # -- reading -----------------------------------------------------------------------------------------------------
    def exists(self, name):
        cur = ""
        for part in name.strip("/").split("/"):
            cur += "/" + part
            if self.lib.H5Lexists(self.fid, cur.encode(), 0) <= 0:
                return False
        return True

    def _shape(self, sid):
        nd = self.lib.H5Sget_simple_extent_ndims(sid)
        dims = (C.c_uint64 * max(nd, 1))()
        if nd > 0:
            self.lib.H5Sget_simple_extent_dims(sid, dims, None)
        return tuple(int(dims[i]) for i in range(nd))

    def _read_typed(self, reader, tid, shape, reclaim):
        lib = self.lib
        n = int(np.prod(shape)) if shape else 1
        cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
        if cls == _H5T_STRING:
            if lib.H5Tis_variable_str(tid) > 0:
                mem = self._vstr()
                buf = (C.c_char_p * max(n, 1))()
                if n and reader(mem, buf) < 0:
                    raise OSError("HDF5 read failed")
                out = np.array([(buf[i] or b"").decode("utf-8") for i in range(n)], dtype=object).reshape(shape)
                if n:
                    reclaim(mem, buf)
                lib.H5Tclose(mem)
                return out
            out = np.empty(n, dtype=f"S{size}")
            if n and reader(tid, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("HDF5 read failed")
            return np.array([x.rstrip(b"\x00").decode("utf-8") for x in out], dtype=object).reshape(shape)
        if cls == _H5T_INTEGER:
            signed = lib.H5Tget_sign(tid) != 0
            dtype = np.dtype(f"<{'i' if signed else 'u'}{size}")
            mem = _native(lib, f"H5T_NATIVE_{'' if signed else 'U'}INT{8 * size}_g")
        elif cls == _H5T_FLOAT:
            dtype = np.dtype(f"<f{size}")
            mem = _native(lib, "H5T_NATIVE_DOUBLE_g" if size == 8 else "H5T_NATIVE_FLOAT_g")
        else:
            raise NotImplementedError(f"HDF5 type class {cls}")
        out = np.empty(n, dtype=dtype)
        if n and reader(mem, out.ctypes.data_as(C.c_void_p)) < 0:
            raise OSError("HDF5 read failed")
        return out.reshape(shape)

    def read(self, name):
        lib = self.lib
        did = lib.H5Dopen2(self.fid, name.encode(), 0)
        if did < 0:
            raise KeyError(name)
        try:
            sid, tid = lib.H5Dget_space(did), lib.H5Dget_type(did)
            shape = self._shape(sid)
            out = self._read_typed(lambda mem, buf: lib.H5Dread(did, mem, 0, 0, 0, buf), tid, shape,
                                   lambda mem, buf: lib.H5Dvlen_reclaim(mem, sid, 0, buf))
            lib.H5Tclose(tid); lib.H5Sclose(sid)
            return out
        finally:
            lib.H5Dclose(did)

    def attrs(self, obj):
        """{name: value} of every attribute of `obj`."""
        lib = self.lib
        oid = lib.H5Oopen(self.fid, obj.encode(), 0)
        if oid < 0:
            raise KeyError(obj)
        out = {}
        try:
            for i in range(lib.H5Aget_num_attrs(oid)):
                aid = lib.H5Aopen_by_idx(oid, b".", 0, 0, i, 0, 0)       # H5_INDEX_NAME, H5_ITER_INC
                nm = C.create_string_buffer(256)
                lib.H5Aget_name(aid, 256, nm)
                sid, tid = lib.H5Aget_space(aid), lib.H5Aget_type(aid)
                shape = self._shape(sid)
                val = self._read_typed(lambda mem, buf: lib.H5Aread(aid, mem, buf), tid, shape,
                                       lambda mem, buf: lib.H5Dvlen_reclaim(mem, sid, 0, buf))
                out[nm.value.decode()] = val.reshape(-1)[0] if shape == () else val
                lib.H5Tclose(tid); lib.H5Sclose(sid); lib.H5Aclose(aid)
        finally:
            lib.H5Oclose(oid)
        return out


# ---------------------------------------------------------------------------------------------------------------------
def _jsonable(x):
    if isinstance(x, np.ndarray):
        return {"__ndarray__": x.tolist(), "dtype": str(x.dtype)}
    if isinstance(x, tuple):
        return {"__tuple__": [_jsonable(v) for v in x]}
    if isinstance(x, (list,)):
        return [_jsonable(v) for v in x]
    if isinstance(x, (np.integer,)):
        return int(x)
    if isinstance(x, (np.floating,)):
        return float(x)
    if isinstance(x, (np.bool_,)):
        return bool(x)
    return x


def _unjson(x):
    if isinstance(x, dict) and "__ndarray__" in x:
        return np.array(x["__ndarray__"], dtype=x["dtype"])
    if isinstance(x, dict) and "__tuple__" in x:
        return tuple(_unjson(v) for v in x["__tuple__"])
    if isinstance(x, list):
        return [_unjson(v) for v in x]
    return x


def _write_annotation(h5, frame):
    h5.group("annotation")
    manifest = []
    for k, name in enumerate(frame.columns):
        col = frame[name]
        vals = col.to_numpy()
        if col.dtype == bool:
            kind, payload = "bool", vals.astype(np.int8)
        elif col.dtype.kind in "iu":
            kind, payload = "int", vals.astype(np.int64)
        elif col.dtype.kind == "f":
            kind, payload = "float", vals.astype(np.float64)
        elif all(isinstance(v, str) for v in vals):
            kind, payload = "str", vals
        else:
            kind, payload = "json", np.array([json.dumps(_jsonable(v)) for v in vals], dtype=object)
        h5.write(f"annotation/c{k}", payload)
        manifest.append({"name": str(name), "kind": kind})
    h5.set_attr("annotation", "columns", json.dumps(manifest))
    h5.set_attr("annotation", "layout", "coolpuppy_amd-columns-1")


def _read_annotation(h5):
    manifest = json.loads(h5.attrs("annotation")["columns"])
    cols = {}
    for k, c in enumerate(manifest):
        raw = h5.read(f"annotation/c{k}")
        if c["kind"] == "bool":
            raw = raw.astype(bool)
        elif c["kind"] == "json":
            out = np.empty(len(raw), dtype=object)
            for i, text in enumerate(raw):
                out[i] = _unjson(json.loads(text))
            raw = out
        cols[c["name"]] = raw
    return pd.DataFrame(cols, columns=[c["name"] for c in manifest])


def save_pileup_df(filename, df, metadata=None, mode="w", compression="gzip"):
    """Write a pile-up DataFrame (the output of pileup()) plus a metadata dict to `filename` — the reference's
    save_pileup_df (coolpuppy/lib/io.py:18-95); see the module docstring for the layout.  compression: "gzip"
    (level 4), an integer gzip level, or None."""
    if compression == "lzf":
        raise ValueError('compression="lzf" needs h5py\'s filter plug-in; this writer offers "gzip" (h5py reads it natively)')
    level = 4 if compression == "gzip" else (int(compression) if compression else 0)
    metadata = {} if metadata is None else metadata
    rows = df.reset_index(drop=True)
    with _H5(filename, "a" if mode == "a" else "w") as h5:
        _write_annotation(h5, rows[[c for c in rows.columns if c not in _ARRAY_COLUMNS]])
        width = int(rows["data"].iloc[0].shape[0])
        stack = np.concatenate([np.asarray(a, dtype=np.float32).reshape(width, width) for a in rows["data"]], axis=0)
        h5.write("data", stack, chunks=(width, width), gzip=level)
        if "store_stripes" in rows.columns and rows["store_stripes"].any():
            for name in ("vertical_stripe", "horizontal_stripe"):
                for i, arr in rows[name].items():
                    dense = np.asarray(arr, dtype=np.float64).reshape(-1, width)
                    nz = dense != 0                                   # implicit zeros, explicit NaN: what scipy's CSR keeps
                    h5.group(f"{name}_{i}")
                    h5.write(f"{name}_{i}/data", dense[nz])
                    h5.write(f"{name}_{i}/indices", np.nonzero(nz)[1].astype(np.int32))
                    h5.write(f"{name}_{i}/indptr", np.concatenate([[0], np.cumsum(nz.sum(axis=1))]).astype(np.int32))
                    h5.set_attr(f"{name}_{i}", "h5sparse_format", "csr")
                    h5.set_attr(f"{name}_{i}", "h5sparse_shape", np.array(dense.shape, np.int64))
            for i, arr in rows["coordinates"].items():
                h5.write(f"coordinates_{i}", np.asarray(arr).astype(object).reshape(-1, 6))
        h5.group("attrs")
        for key, val in metadata.items():
            h5.set_attr("attrs", str(key), False if val is None else val)
        h5.set_attr("attrs", "version", __version__)


def load_pileup_df(filename, quaich=False, skipstripes=False):
    """Read a .clpy back into a DataFrame (coolpuppy/lib/io.py:98-155): annotation columns, `data` (one W x W float
    array per row), stripes and coordinates when stored, the metadata attributes as constant columns; quaich=True also
    parses sample / bedname out of a quaich-style file name."""
    with _H5(filename, "r") as h5:
        if "columns" in h5.attrs("annotation"):
            annotation = _read_annotation(h5)
        else:                                            # a PyTables store written by the reference
            annotation = pd.read_hdf(filename, "annotation")
        stack = h5.read("data")
        width = stack.shape[1]
        annotation["data"] = [stack[i * width:(i + 1) * width] for i in range(stack.shape[0] // width)]
        if not skipstripes and h5.exists("vertical_stripe_0"):
            cols = {"vertical_stripe": [], "horizontal_stripe": [], "coordinates": []}
            for i in range(len(annotation)):
                for name in ("vertical_stripe", "horizontal_stripe"):
                    shape = tuple(int(x) for x in h5.attrs(f"{name}_{i}")["h5sparse_shape"])
                    dense = np.zeros(shape)
                    indptr = h5.read(f"{name}_{i}/indptr")
                    rows_of = np.repeat(np.arange(shape[0]), np.diff(indptr))
                    dense[rows_of, h5.read(f"{name}_{i}/indices")] = h5.read(f"{name}_{i}/data")
                    cols[name].append(dense)
                cols["coordinates"].append(h5.read(f"coordinates_{i}").astype("U13"))
            for k, v in cols.items():
                annotation[k] = v
        metadata = h5.attrs("attrs")
    for key, val in metadata.items():
        if key != "version":
            annotation[key] = val
    if quaich:
        m = re.search(r"^(.*)-(?:[0-9]+)_over_(.*)_(?:[0-9]+-shifts|expected).*\.clpy", os.path.basename(filename))
        annotation["sample"], annotation["bedname"] = m.groups()
    return annotation


def load_pileup_df_list(files, quaich=False, nice_metadata=True, skipstripes=False):
    """Several .clpy files as one frame; nice_metadata adds `norm`: "expected", "shifts" or "none" (:158-190)."""
    pups = pd.concat([load_pileup_df(p, quaich=quaich, skipstripes=skipstripes) for p in files], ignore_index=True)
    if nice_metadata:
        expected = pups["expected"].astype(bool).to_numpy()
        shifted = (pups["nshifts"].to_numpy() > 0)
        pups["norm"] = np.select([expected, shifted], ["expected", "shifts"], default="none")
    return pups
