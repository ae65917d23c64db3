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
    /annotation            every other column, as the PyTables "fixed" store pandas.to_hdf(filename, "annotation") writes
                           (what the reference does, :47-53, and what its load_pileup_df / plotpuppy read back with
                           pandas.read_hdf, :123): group attributes pandas_type = "frame", ndim, nblocks, ...; `axis0` (column
                           names) and `axis1` (row index) arrays; per dtype block `block<k>_items` (its column names) and
                           `block<k>_values` — float64 / int64 as (rows, columns) arrays, bool as 8-bit bitfields, everything
                           else (strings, tuples, arrays such as `num`) as ONE pickled object ndarray in a variable-length
                           uint8 array (PyTables' ObjectAtom) — every node carrying PyTables' CLASS / VERSION / FLAVOR /
                           TITLE attributes.  The layout was taken from a file pandas 2.3 + PyTables 3.6 wrote in this
                           image's conda environment and is compared with such a file node by node in
                           tests/test_cool_io.py; it is written and read here through libhdf5 alone.
                           (layout="columns" keeps round 2's pickle-free alternative: /annotation/c<k> per column plus a
                           JSON manifest — readable without unpickling, NOT readable by the reference.)

Host-side, off the pile-up path.
"""
import ctypes as C
import json
import os
import pickle
import re

import numpy as np
import pandas as pd

from .. import __version__
from ..cool_io import _hdf5, _native

_ARRAY_COLUMNS = ["data", "vertical_stripe", "horizontal_stripe", "coordinates"]
_H5F_ACC_TRUNC, _H5F_ACC_RDWR, _H5F_ACC_RDONLY = 2, 1, 0
_H5T_CSET_UTF8, _H5T_VARIABLE = 1, C.c_size_t(-1).value
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING, _H5T_BITFIELD, _H5T_ENUM, _H5T_VLEN = 0, 1, 3, 4, 8, 9
_H5S_SCALAR, _H5S_NULL = 0, 2
_H5S_UNLIMITED = C.c_uint64(-1).value


class _hvl(C.Structure):                # hvl_t
    _fields_ = [("len", C.c_size_t), ("p", C.c_void_p)]

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
    lib.H5Pset_shuffle.argtypes = [_hid]
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
    lib.H5Tvlen_create.restype = _hid; lib.H5Tvlen_create.argtypes = [_hid]
    lib.H5Tget_super.restype = _hid; lib.H5Tget_super.argtypes = [_hid]
    lib.H5Tenum_create.restype = _hid; lib.H5Tenum_create.argtypes = [_hid]
    lib.H5Tenum_insert.restype = C.c_int; lib.H5Tenum_insert.argtypes = [_hid, C.c_char_p, C.c_void_p]
    lib.H5Tset_strpad.argtypes = [_hid, C.c_int]
    lib.H5Oopen.restype = _hid; lib.H5Oopen.argtypes = [_hid, C.c_char_p, _hid]
    lib.H5Oclose.argtypes = [_hid]
    lib.H5Sget_simple_extent_type.restype = C.c_int; lib.H5Sget_simple_extent_type.argtypes = [_hid]
    lib._clpy_ready = True
    return lib


_NP2H5 = {"float32": "H5T_NATIVE_FLOAT_g", "float64": "H5T_NATIVE_DOUBLE_g", "int64": "H5T_NATIVE_INT64_g",
          "int32": "H5T_NATIVE_INT32_g", "int8": "H5T_NATIVE_INT8_g", "uint8": "H5T_NATIVE_UINT8_g"}


class _H5:
    """The dozen libhdf5 calls a .clpy needs: groups, n-d numeric / string datasets, scalar and 1-d attributes."""

    def __init__(self, path, mode, userblock=0):
        self.lib = lib = _lib()
        p = os.fsencode(path)
        if mode == "w" or (mode == "a" and not os.path.exists(path)):
            fcpl = 0
            if userblock:                                     # (tests: a file whose HDF5 data starts behind a user block)
                lib.H5Pset_userblock.argtypes = [_hid, C.c_uint64]
                fcpl = lib.H5Pcreate(_native(lib, "H5P_CLS_FILE_CREATE_ID_g"))
                if fcpl < 0 or lib.H5Pset_userblock(fcpl, int(userblock)) < 0:
                    raise OSError(f"cannot set a user block of {userblock} bytes (a power of two >= 512)")
            self.fid = lib.H5Fcreate(p, _H5F_ACC_TRUNC, fcpl, 0)
            if fcpl:
                lib.H5Pclose(fcpl)
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

    def _fixed_str(self, size, utf8=False):
        t = self.lib.H5Tcopy(_native(self.lib, "H5T_C_S1_g"))
        self.lib.H5Tset_size(t, max(int(size), 1))
        if utf8:
            self.lib.H5Tset_cset(t, _H5T_CSET_UTF8)
        return t

    def write_fixed_strings(self, name, values):
        """1-d array of byte strings as a fixed-width ASCII string dataset (how PyTables stores numpy 'S' arrays)."""
        lib = self.lib
        arr = np.asarray([v if isinstance(v, bytes) else str(v).encode("utf-8") for v in values] or [b""])
        arr = arr.astype(f"S{max(arr.dtype.itemsize, 1)}")[: len(values)]
        t = self._fixed_str(arr.dtype.itemsize)
        space = self._space((len(values),))
        did = lib.H5Dcreate2(self.fid, name.encode(), t, space, 0, 0, 0)
        ok = did >= 0 and (len(values) == 0 or lib.H5Dwrite(did, t, 0, 0, 0, np.ascontiguousarray(arr).ctypes.data_as(C.c_void_p)) >= 0)
        if did >= 0:
            lib.H5Dclose(did)
        lib.H5Sclose(space); lib.H5Tclose(t)
        if not ok:
            raise OSError(f"cannot write dataset {name!r}")

    def write_bitfield(self, name, arr):
        """bool array as 8-bit bitfields (PyTables' BoolAtom)."""
        lib = self.lib
        arr = np.ascontiguousarray(np.asarray(arr, dtype=bool).astype(np.uint8))
        t = _native(lib, "H5T_NATIVE_B8_g")
        space = self._space(arr.shape)
        did = lib.H5Dcreate2(self.fid, name.encode(), t, space, 0, 0, 0)
        ok = did >= 0 and (arr.size == 0 or lib.H5Dwrite(did, t, 0, 0, 0, arr.ctypes.data_as(C.c_void_p)) >= 0)
        if did >= 0:
            lib.H5Dclose(did)
        lib.H5Sclose(space)
        if not ok:
            raise OSError(f"cannot write dataset {name!r}")

    def write_vlen_bytes(self, name, blobs):
        """A list of byte strings as an extendible 1-d array of variable-length uint8 rows (PyTables' VLArray)."""
        lib = self.lib
        t = lib.H5Tvlen_create(_native(lib, "H5T_NATIVE_UINT8_g"))
        n = len(blobs)
        dims, maxd = (C.c_uint64 * 1)(n), (C.c_uint64 * 1)(_H5S_UNLIMITED)
        space = lib.H5Screate_simple(1, dims, maxd)
        dcpl = lib.H5Pcreate(_native(lib, "H5P_CLS_DATASET_CREATE_ID_g"))
        lib.H5Pset_chunk(dcpl, 1, (C.c_uint64 * 1)(65536))
        keep = [np.frombuffer(b, dtype=np.uint8) for b in blobs]
        buf = (_hvl * max(n, 1))()
        for i, a in enumerate(keep):
            buf[i].len, buf[i].p = a.size, a.ctypes.data
        did = lib.H5Dcreate2(self.fid, name.encode(), t, space, 0, dcpl, 0)
        ok = did >= 0 and (n == 0 or lib.H5Dwrite(did, t, 0, 0, 0, buf) >= 0)
        if did >= 0:
            lib.H5Dclose(did)
        lib.H5Sclose(space); lib.H5Pclose(dcpl); lib.H5Tclose(t)
        if not ok:
            raise OSError(f"cannot write dataset {name!r}")

    def read_vlen_bytes(self, name):
        lib = self.lib
        did = lib.H5Dopen2(self.fid, name.encode(), 0)
        if did < 0:
            raise KeyError(name)
        try:
            sid = lib.H5Dget_space(did)
            n = self._shape(sid)[0]
            t = lib.H5Tvlen_create(_native(lib, "H5T_NATIVE_UINT8_g"))
            buf = (_hvl * max(n, 1))()
            if n and lib.H5Dread(did, t, 0, 0, 0, buf) < 0:
                raise OSError("HDF5 read failed")
            out = [C.string_at(buf[i].p, buf[i].len) if buf[i].len else b"" for i in range(n)]
            if n:
                lib.H5Dvlen_reclaim(t, sid, 0, buf)
            lib.H5Tclose(t); lib.H5Sclose(sid)
            return out
        finally:
            lib.H5Dclose(did)

    def set_pytables_attr(self, obj, name, value):
        """Attributes the way PyTables writes them: str -> fixed-length UTF-8 string scalar ('' -> a null-dataspace S1),
        bool -> 8-bit bitfield scalar, int -> int64 scalar."""
        lib = self.lib
        oid = lib.H5Oopen(self.fid, obj.encode(), 0)
        if oid < 0:
            raise KeyError(obj)
        try:
            if isinstance(value, str):
                raw = value.encode("utf-8")
                t = self._fixed_str(len(raw), utf8=True)
                space = lib.H5Screate(_H5S_NULL if not raw else _H5S_SCALAR)
                aid = lib.H5Acreate2(oid, name.encode(), t, space, 0, 0)
                ok = aid >= 0 and (not raw or lib.H5Awrite(aid, t, C.c_char_p(raw)) >= 0)
                lib.H5Tclose(t)
            else:
                v = np.array(value, dtype=np.uint8 if isinstance(value, (bool, np.bool_)) else np.int64)
                t = _native(lib, "H5T_NATIVE_B8_g" if v.dtype == np.uint8 else "H5T_NATIVE_INT64_g")
                space = lib.H5Screate(_H5S_SCALAR)
                aid = lib.H5Acreate2(oid, name.encode(), t, space, 0, 0)
                ok = aid >= 0 and lib.H5Awrite(aid, t, v.ctypes.data_as(C.c_void_p)) >= 0
            if aid >= 0:
                lib.H5Aclose(aid)
            lib.H5Sclose(space)
            if not ok:
                raise OSError(f"cannot write attribute {name!r} of {obj!r}")
        finally:
            lib.H5Oclose(oid)

    def write(self, name, arr, chunks=None, gzip=None, shuffle=False):
        """Numeric array (dtype kept) or array of str (variable-length UTF-8), any rank."""
        lib = self.lib
        arr = np.asarray(arr)
        strings = arr.dtype.kind in "OUS"
        space = self._space(arr.shape)                       # before ascontiguousarray, which makes scalars 1-d
        dcpl = 0
        if chunks is not None and arr.size:
            dcpl = lib.H5Pcreate(_native(lib, "H5P_CLS_DATASET_CREATE_ID_g"))
            lib.H5Pset_chunk(dcpl, len(chunks), (C.c_uint64 * len(chunks))(*chunks))
            if shuffle:                                      # (cooler's default pipeline: shuffle, then gzip)
                lib.H5Pset_shuffle(dcpl)
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
            if isinstance(value, (list, tuple)) and not all(isinstance(x, (bool, int, float, np.number)) for x in value):
                value = json.dumps(_jsonable(list(value)))      # e.g. a list of names: stored as JSON text
            if isinstance(value, (str, bytes, os.PathLike)):
                t = self._vstr()
                space = self._space(())
                p = C.c_char_p(os.fsdecode(value).encode("utf-8") if not isinstance(value, bytes) else value)
                aid = lib.H5Acreate2(oid, name.encode(), t, space, 0, 0)
                ok = aid >= 0 and lib.H5Awrite(aid, t, C.byref(p)) >= 0
                lib.H5Tclose(t)
            elif isinstance(value, (bool, np.bool_)):
                # h5py's bool: an int8 enum {FALSE = 0, TRUE = 1} (the reference writes `attrs[key] = False` for None)
                t = lib.H5Tenum_create(_native(lib, "H5T_NATIVE_INT8_g"))
                for nm, iv in ((b"FALSE", 0), (b"TRUE", 1)):
                    lib.H5Tenum_insert(t, nm, C.byref(C.c_int8(iv)))
                space = self._space(())
                aid = lib.H5Acreate2(oid, name.encode(), t, space, 0, 0)
                ok = aid >= 0 and lib.H5Awrite(aid, t, C.byref(C.c_int8(1 if value else 0))) >= 0
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
        elif cls == _H5T_BITFIELD and size == 1:         # PyTables' bool
            out = np.empty(n, dtype=np.uint8)
            if n and reader(_native(lib, "H5T_NATIVE_B8_g"), out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("HDF5 read failed")
            return out.astype(bool).reshape(shape)
        elif cls == _H5T_ENUM:                           # h5py's bool: an int8 enum {FALSE = 0, TRUE = 1}
            base = lib.H5Tget_super(tid)
            bsize = lib.H5Tget_size(base)
            lib.H5Tclose(base)
            out = np.empty(n, dtype=np.dtype(f"<i{bsize}"))
            if n and reader(tid, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError("HDF5 read failed")
            return out.astype(bool).reshape(shape)
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
                if lib.H5Sget_simple_extent_type(sid) == _H5S_NULL:      # PyTables' empty string (TITLE)
                    out[nm.value.decode()] = ""
                    lib.H5Tclose(tid); lib.H5Sclose(sid); lib.H5Aclose(aid)
                    continue
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


# ---- /annotation as pandas' PyTables "fixed" store ------------------------------------------------------------------------
def _pt_node(h5, name, cls, version, extra=()):
    for k, v in (("CLASS", cls), ("VERSION", version), ("TITLE", "")) + tuple(extra):
        h5.set_pytables_attr(name, k, v)


def _pt_array_attrs(h5, name, kind=None):
    extra = (("FLAVOR", "numpy"),) + ((("kind", kind), ("name", "N.")) if kind else ()) + (("transposed", True),)
    _pt_node(h5, name, "ARRAY", "2.4", extra)


def _write_annotation_pytables(h5, frame):
    """pandas.DataFrame.to_hdf(..., "annotation") in its default "fixed" format, through libhdf5 (module docstring)."""
    _pt_node(h5, "/", "GROUP", "1.0", (("PYTABLES_FORMAT_VERSION", "2.1"),))
    h5.group("annotation")
    names = [str(c) for c in frame.columns]
    blocks = {"bool": [], "float": [], "int": [], "object": []}     # pandas consolidates columns by dtype; so do we
    for c in frame.columns:
        dt = frame[c].dtype
        blocks["bool" if dt == bool else "float" if dt.kind == "f" else "int" if dt.kind in "iu" else "object"].append(c)
    blocks = [(k, cols) for k, cols in blocks.items() if cols]
    _pt_node(h5, "annotation", "GROUP", "1.0",
             (("pandas_type", "frame"), ("pandas_version", "0.15.2"), ("encoding", "UTF-8"), ("errors", "strict"),
              ("ndim", 2), ("nblocks", len(blocks)), ("axis0_variety", "regular"), ("axis1_variety", "regular"))
             + tuple((f"block{i}_items_variety", "regular") for i in range(len(blocks))))
    h5.write_fixed_strings("annotation/axis0", names)
    _pt_array_attrs(h5, "annotation/axis0", "string")
    h5.write("annotation/axis1", np.arange(len(frame), dtype=np.int64))
    _pt_array_attrs(h5, "annotation/axis1", "integer")
    for i, (kind, cols) in enumerate(blocks):
        h5.write_fixed_strings(f"annotation/block{i}_items", [str(c) for c in cols])
        _pt_array_attrs(h5, f"annotation/block{i}_items", "string")
        key = f"annotation/block{i}_values"
        if kind == "object":
            vals = np.empty((len(frame), len(cols)), dtype=object)          # (rows, columns): pandas writes values.T
            for j, c in enumerate(cols):
                col = frame[c].to_numpy(dtype=object)
                for r in range(len(frame)):
                    vals[r, j] = col[r]
            h5.write_vlen_bytes(key, [pickle.dumps(vals, protocol=4)])
            _pt_node(h5, key, "VLARRAY", "1.4", (("PSEUDOATOM", "object"), ("transposed", True)))
            continue
        vals = np.stack([frame[c].to_numpy() for c in cols], axis=1)
        if kind == "bool":
            h5.write_bitfield(key, vals)
        else:
            h5.write(key, vals.astype(np.float64 if kind == "float" else np.int64))
        _pt_array_attrs(h5, key)


class _ArrayUnpickler(pickle.Unpickler):
    """Unpickler for the object blocks of a pandas "fixed" store: a numpy object ndarray of plain Python values.  Only the
    numpy array / dtype / scalar reconstructors and the builtin value types resolve; anything else in the stream (a crafted
    file could name os.system) raises instead of being imported — pd.read_hdf, which the reference uses, offers no such guard."""
    # exact (module, name) pairs — "any numpy submodule" would let a crafted file import numpy.distutils / numpy.f2py and friends
    _NUMPY = {("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"),
              ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
              ("numpy", "ndarray"), ("numpy", "dtype"),
              ("numpy.core.numeric", "_frombuffer"), ("numpy._core.numeric", "_frombuffer")}
    _BUILTINS = {"list", "tuple", "dict", "set", "frozenset", "str", "bytes", "bytearray", "int", "float", "complex", "bool",
                 "slice", "range", "NoneType"}

    def find_class(self, module, name):
        if (module, name) in self._NUMPY:
            if module.startswith("numpy.core."):             # (numpy 2 renamed the package; old pickles still name the old one)
                module = "numpy._core." + module[len("numpy.core."):] if hasattr(np, "_core") else module
            return super().find_class(module, name)
        if module == "builtins" and name in self._BUILTINS:
            return super().find_class(module, name)
        if (module, name) == ("_codecs", "encode"):      # how protocol <= 2 spells the array's byte string (str -> bytes, nothing else)
            return super().find_class(module, name)
        raise pickle.UnpicklingError(f".clpy annotation: refusing to unpickle {module}.{name} (only numpy arrays of plain values are read)")


def _restricted_loads(data):
    """The object ndarray pickled in a block of a "fixed" store — and nothing else: the unpickled value must BE an ndarray."""
    import io as _io
    val = _ArrayUnpickler(_io.BytesIO(data)).load()
    if not isinstance(val, np.ndarray):
        raise pickle.UnpicklingError(f".clpy annotation: an object block unpickled to {type(val).__name__}, not to an array")
    return val


def _read_annotation_pytables(h5):
    """A PyTables "fixed" frame (written by the reference through pandas, or by _write_annotation_pytables) -> DataFrame."""
    at = h5.attrs("annotation")
    if str(at.get("pandas_type", "")) != "frame":
        raise ValueError("/annotation is neither a pandas 'fixed' frame nor the column layout of this package")
    names = list(h5.read("annotation/axis0"))
    index = h5.read("annotation/axis1")
    cols = {}
    for i in range(int(at["nblocks"])):
        items = list(h5.read(f"annotation/block{i}_items"))
        key = f"annotation/block{i}_values"
        battrs = h5.attrs(key)
        if str(battrs.get("CLASS", "")) == "VLARRAY":
            vals = _restricted_loads(h5.read_vlen_bytes(key)[0])  # the file's own pickled object ndarray, (rows, columns)
        else:
            vals = h5.read(key)
        vals = np.asarray(vals).reshape(len(index), len(items)) if len(items) else vals
        for j, c in enumerate(items):
            col = vals[:, j]
            if col.dtype == object:
                keep = np.empty(len(col), dtype=object)
                for r in range(len(col)):
                    keep[r] = col[r]
                col = keep
            cols[c] = col
    out = pd.DataFrame({c: pd.Series(list(cols[c]) if cols[c].dtype == object else cols[c], index=index) for c in names},
                       columns=names)
    return out


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


def save_pileup_df(filename, df, metadata=None, mode="w", compression="gzip", layout="pytables"):
    """Write a pile-up DataFrame (the output of pileup()) plus a metadata dict to `filename` — the reference's
    save_pileup_df (coolpuppy/lib/io.py:18-95); see the module docstring for the layout.  compression: "gzip"
    (level 4), an integer gzip level, or None.  layout: "pytables" (what the reference writes and reads: /annotation is
    pandas' fixed store, object columns pickled) or "columns" (pickle-free /annotation of this package only)."""
    if layout not in ("pytables", "columns"):
        raise ValueError('layout must be "pytables" or "columns"')
    if compression == "lzf":
        raise ValueError('compression="lzf" needs h5py\'s filter plug-in; this writer offers "gzip" (h5py reads it natively)')
    level = 4 if compression == "gzip" else (int(compression) if compression else 0)
    metadata = {} if metadata is None else metadata
    rows = df.reset_index(drop=True)
    with _H5(filename, "a" if mode == "a" else "w") as h5:
        (_write_annotation_pytables if layout == "pytables" else _write_annotation)(
            h5, rows[[c for c in rows.columns if c not in _ARRAY_COLUMNS]])
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
        else:                                            # a PyTables "fixed" store: the reference's files, and ours
            annotation = _read_annotation_pytables(h5)
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
