"""Read a ``.cool`` file into an :class:`~coolpuppy_amd.cooler_lite.ArrayCooler` (host I/O).

The reference opens coolers through the ``cooler`` package (``cooler.Cooler(path)``, coolpuppy/CLI.py:406), which
reads HDF5 through h5py.  Neither is guaranteed here, so this reader goes to libhdf5 directly via ctypes (h5py is
used when importable).  It reads exactly the datasets the pile-up path consumes — the cooler schema's

    chroms/{name,length}   bins/{chrom,start,end,weight,...}   pixels/{bin1_id,bin2_id,count}
    indexes/{bin1_offset,chrom_offset}                          attrs: bin-size

— and nothing is transformed: ``indexes/bin1_offset`` IS the CSR row pointer of the upper-triangular pixel table
the GPU engine loads (``pixels/bin1_id`` is not even read).  Multi-resolution files: pass ``group="resolutions/10000"``.
"""
import ctypes as C
import os

import numpy as np
import pandas as pd

from .cooler_lite import ArrayCooler

_H5F_ACC_RDONLY = 0
_H5P_DEFAULT = 0
_H5S_ALL = 0
_H5T_INTEGER, _H5T_FLOAT, _H5T_STRING, _H5T_ENUM = 0, 1, 3, 8

_lib = None


def _hdf5():
    global _lib
    if _lib is not None:
        return _lib
    cands = [os.environ.get("COOLPUPPY_AMD_LIBHDF5", ""), "libhdf5.so", "libhdf5_serial.so", "/opt/conda/lib/libhdf5.so",
             "/usr/lib/x86_64-linux-gnu/libhdf5_serial.so", "/usr/lib/x86_64-linux-gnu/hdf5/serial/libhdf5.so"]
    last = None
    for c in cands:
        if not c:
            continue
        try:
            lib = C.CDLL(c)
            break
        except OSError as e:
            last = e
    else:
        raise RuntimeError(f"libhdf5 not found (set COOLPUPPY_AMD_LIBHDF5 to its path): {last}")
    hid = C.c_int64
    lib.H5open.restype = C.c_int
    lib.H5Fopen.restype = hid; lib.H5Fopen.argtypes = [C.c_char_p, C.c_uint, hid]
    lib.H5Fclose.argtypes = [hid]
    lib.H5Dopen2.restype = hid; lib.H5Dopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Dclose.argtypes = [hid]
    lib.H5Dget_space.restype = hid; lib.H5Dget_space.argtypes = [hid]
    lib.H5Dget_type.restype = hid; lib.H5Dget_type.argtypes = [hid]
    lib.H5Sget_simple_extent_npoints.restype = C.c_int64; lib.H5Sget_simple_extent_npoints.argtypes = [hid]
    lib.H5Sclose.argtypes = [hid]
    lib.H5Screate_simple.restype = hid; lib.H5Screate_simple.argtypes = [C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.H5Sselect_hyperslab.restype = C.c_int
    lib.H5Sselect_hyperslab.argtypes = [hid, C.c_int, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.H5Tget_class.restype = C.c_int; lib.H5Tget_class.argtypes = [hid]
    lib.H5Tget_size.restype = C.c_size_t; lib.H5Tget_size.argtypes = [hid]
    lib.H5Tget_sign.restype = C.c_int; lib.H5Tget_sign.argtypes = [hid]
    lib.H5Tget_super.restype = hid; lib.H5Tget_super.argtypes = [hid]
    lib.H5Tget_nmembers.restype = C.c_int; lib.H5Tget_nmembers.argtypes = [hid]
    lib.H5Tget_member_name.restype = C.c_void_p; lib.H5Tget_member_name.argtypes = [hid, C.c_uint]
    lib.H5Tget_member_value.argtypes = [hid, C.c_uint, C.c_void_p]
    lib.H5Tis_variable_str.restype = C.c_int; lib.H5Tis_variable_str.argtypes = [hid]
    lib.H5Tcopy.restype = hid; lib.H5Tcopy.argtypes = [hid]
    lib.H5Tclose.argtypes = [hid]
    lib.H5free_memory.argtypes = [C.c_void_p]
    lib.H5Dread.restype = C.c_int; lib.H5Dread.argtypes = [hid, hid, hid, hid, hid, C.c_void_p]
    lib.H5Lexists.restype = C.c_int; lib.H5Lexists.argtypes = [hid, C.c_char_p, hid]
    lib.H5Aopen_by_name.restype = hid; lib.H5Aopen_by_name.argtypes = [hid, C.c_char_p, C.c_char_p, hid, hid]
    lib.H5Aread.restype = C.c_int; lib.H5Aread.argtypes = [hid, hid, C.c_void_p]
    lib.H5Aclose.argtypes = [hid]
    lib.H5Gopen2.restype = hid; lib.H5Gopen2.argtypes = [hid, C.c_char_p, hid]
    lib.H5Gclose.argtypes = [hid]
    lib.H5Lget_name_by_idx.restype = C.c_ssize_t
    lib.H5Lget_name_by_idx.argtypes = [hid, C.c_char_p, C.c_int, C.c_int, C.c_uint64, C.c_char_p, C.c_size_t, hid]
    lib.H5Eset_auto2.argtypes = [hid, C.c_void_p, C.c_void_p]
    # where a dataset's bytes sit in the file (_File.layout: the streamed loader reads them itself, on several threads)
    lib.H5Dget_offset.restype = C.c_uint64; lib.H5Dget_offset.argtypes = [hid]
    lib.H5Dget_create_plist.restype = hid; lib.H5Dget_create_plist.argtypes = [hid]
    lib.H5Pget_layout.restype = C.c_int; lib.H5Pget_layout.argtypes = [hid]
    lib.H5Pget_chunk.restype = C.c_int; lib.H5Pget_chunk.argtypes = [hid, C.c_int, C.POINTER(C.c_uint64)]
    lib.H5Pget_nfilters.restype = C.c_int; lib.H5Pget_nfilters.argtypes = [hid]
    lib.H5Pget_filter2.restype = C.c_int
    lib.H5Pget_filter2.argtypes = [hid, C.c_uint, C.POINTER(C.c_uint), C.POINTER(C.c_size_t), C.POINTER(C.c_uint), C.c_size_t, C.c_char_p, C.POINTER(C.c_uint)]
    lib.H5Pclose.argtypes = [hid]
    lib.H5Fget_create_plist.restype = hid; lib.H5Fget_create_plist.argtypes = [hid]
    lib.H5Pget_userblock.restype = C.c_int; lib.H5Pget_userblock.argtypes = [hid, C.POINTER(C.c_uint64)]
    lib.H5Tget_order.restype = C.c_int; lib.H5Tget_order.argtypes = [hid]
    if hasattr(lib, "H5Dget_num_chunks"):
        lib.H5Dget_num_chunks.restype = C.c_int; lib.H5Dget_num_chunks.argtypes = [hid, hid, C.POINTER(C.c_uint64)]
        lib.H5Dget_chunk_info.restype = C.c_int
        lib.H5Dget_chunk_info.argtypes = [hid, hid, C.c_uint64, C.POINTER(C.c_uint64), C.POINTER(C.c_uint), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    lib.H5open()
    lib.H5Eset_auto2(0, None, None)          # errors are reported through return codes, not stderr spam
    _lib = lib
    return lib


def _native(lib, name):
    return C.c_int64.in_dll(lib, name).value


class _File:
    def __init__(self, path):
        self.lib = _hdf5()
        self.fid = self.lib.H5Fopen(os.fsencode(path), _H5F_ACC_RDONLY, _H5P_DEFAULT)
        if self.fid < 0:
            raise OSError(f"cannot open {path!r} as HDF5")

    def close(self):
        if self.fid >= 0:
            self.lib.H5Fclose(self.fid)
            self.fid = -1

    def children(self, group):
        """Names of the links in a group (alphabetical)."""
        out = []
        g = (group or "/").encode()
        while True:
            size = self.lib.H5Lget_name_by_idx(self.fid, g, 0, 0, len(out), None, 0, _H5P_DEFAULT)
            if size < 0:
                return out
            buf = C.create_string_buffer(size + 1)
            self.lib.H5Lget_name_by_idx(self.fid, g, 0, 0, len(out), buf, size + 1, _H5P_DEFAULT)
            out.append(buf.value.decode())

    def exists(self, name):
        parts = name.strip("/").split("/")
        cur = ""
        for p in parts:
            cur += "/" + p
            if self.lib.H5Lexists(self.fid, cur.encode(), _H5P_DEFAULT) <= 0:
                return False
        return True

    def read(self, name):
        """Dataset -> numpy array (integers, floats, fixed-length strings, integer-backed enums)."""
        lib = self.lib
        did = lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        if did < 0:
            raise KeyError(f"dataset {name!r} not found")
        try:
            sid = lib.H5Dget_space(did)
            n = lib.H5Sget_simple_extent_npoints(sid)
            lib.H5Sclose(sid)
            tid = lib.H5Dget_type(did)
            cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
            labels = None
            if cls == _H5T_ENUM:
                nm = lib.H5Tget_nmembers(tid)
                sup = lib.H5Tget_super(tid)
                bsize = lib.H5Tget_size(sup)
                lib.H5Tclose(sup)
                labels = {}
                for i in range(nm):
                    p = lib.H5Tget_member_name(tid, i)
                    nm_s = C.string_at(p).decode()
                    lib.H5free_memory(p)
                    val = C.c_int64(0)
                    lib.H5Tget_member_value(tid, i, C.byref(val))
                    labels[int(np.frombuffer(bytes(val)[:bsize], dtype=f"<i{bsize}")[0])] = nm_s
                dtype, mem, size = np.dtype(f"<i{bsize}"), tid, bsize
            elif cls == _H5T_INTEGER:
                signed = lib.H5Tget_sign(tid) != 0
                dtype = np.dtype(f"<{'i' if signed else 'u'}{size}")
                mem = _native(lib, {(1, True): "H5T_NATIVE_INT8_g", (2, True): "H5T_NATIVE_INT16_g",
                                    (4, True): "H5T_NATIVE_INT32_g", (8, True): "H5T_NATIVE_INT64_g",
                                    (1, False): "H5T_NATIVE_UINT8_g", (2, False): "H5T_NATIVE_UINT16_g",
                                    (4, False): "H5T_NATIVE_UINT32_g", (8, False): "H5T_NATIVE_UINT64_g"}[(size, signed)])
            elif cls == _H5T_FLOAT:
                dtype = np.dtype(f"<f{size}")
                mem = _native(lib, "H5T_NATIVE_DOUBLE_g" if size == 8 else "H5T_NATIVE_FLOAT_g")
            elif cls == _H5T_STRING:
                if lib.H5Tis_variable_str(tid) > 0:
                    raise NotImplementedError(f"{name}: variable-length strings are not supported")
                dtype, mem = np.dtype(f"S{size}"), tid
            else:
                raise NotImplementedError(f"{name}: HDF5 type class {cls} is not supported")
            out = np.empty(int(n), dtype=dtype)
            if n > 0 and lib.H5Dread(did, mem, _H5S_ALL, _H5S_ALL, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) < 0:
                raise OSError(f"H5Dread failed for {name!r}")
            lib.H5Tclose(tid)
            if labels is not None:
                lut = np.array([labels[i] for i in range(max(labels) + 1)], dtype=object)
                return lut[out]
            return out
        finally:
            lib.H5Dclose(did)

    def dataset_info(self, name):
        """(number of elements, numpy dtype) of an integer / float dataset, without reading it."""
        lib = self.lib
        did = lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        if did < 0:
            raise KeyError(f"dataset {name!r} not found")
        try:
            sid = lib.H5Dget_space(did)
            n = lib.H5Sget_simple_extent_npoints(sid)
            lib.H5Sclose(sid)
            tid = lib.H5Dget_type(did)
            cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
            signed = lib.H5Tget_sign(tid) != 0 if cls == _H5T_INTEGER else True
            lib.H5Tclose(tid)
            if cls == _H5T_INTEGER:
                return int(n), np.dtype(f"<{'i' if signed else 'u'}{size}")
            if cls == _H5T_FLOAT:
                return int(n), np.dtype(f"<f{size}")
            raise NotImplementedError(f"{name}: HDF5 type class {cls} is not a number")
        finally:
            lib.H5Dclose(did)

    def read_into(self, name, first, out):
        """Elements [first, first + len(out)) of a 1-D numeric dataset straight into the (contiguous) array `out` — a hyperslab
        read with the array's own dtype as the memory type: no intermediate copy (`out` may be a view of page-locked memory)."""
        lib = self.lib
        did = lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        if did < 0:
            raise KeyError(f"dataset {name!r} not found")
        try:
            kind, size = out.dtype.kind, out.dtype.itemsize
            mem = _native(lib, {("i", 4): "H5T_NATIVE_INT32_g", ("i", 8): "H5T_NATIVE_INT64_g", ("u", 4): "H5T_NATIVE_UINT32_g",
                                ("u", 8): "H5T_NATIVE_UINT64_g", ("f", 8): "H5T_NATIVE_DOUBLE_g", ("f", 4): "H5T_NATIVE_FLOAT_g"}[(kind, size)])
            fsp = lib.H5Dget_space(did)
            start, count = (C.c_uint64 * 1)(int(first)), (C.c_uint64 * 1)(int(out.shape[0]))
            msp = lib.H5Screate_simple(1, count, None)
            ok = lib.H5Sselect_hyperslab(fsp, 0, start, None, count, None) >= 0 and \
                lib.H5Dread(did, mem, msp, fsp, _H5P_DEFAULT, out.ctypes.data_as(C.c_void_p)) >= 0
            lib.H5Sclose(msp); lib.H5Sclose(fsp)
            if not ok:
                raise OSError(f"hyperslab read of {name!r} [{first}, +{out.shape[0]}) failed")
        finally:
            lib.H5Dclose(did)

    def _userblock(self):
        """Size of the file's user block (H5Pget_userblock on the file creation property list); -1 when it cannot be asked."""
        lib = self.lib
        pl = lib.H5Fget_create_plist(self.fid)
        if pl < 0:
            return -1
        try:
            size = C.c_uint64(0)
            return int(size.value) if lib.H5Pget_userblock(pl, C.byref(size)) >= 0 else -1
        finally:
            lib.H5Pclose(pl)

    def layout(self, name):
        """Where the bytes of a 1-D numeric dataset sit in the file, for readers that fetch them without libhdf5 (whose calls are
        serialised by its global lock): {"dtype", "n", "kind": "contiguous", "offset"} or {"kind": "chunked", "chunk": elements per
        chunk, "chunks": [(first element, file offset, stored bytes, filter mask)], "filters": [1 = deflate | 2 = shuffle, in pipeline order]}.  None for
        anything else (other filters, big-endian or non-native types, no storage allocated yet): the caller uses H5Dread."""
        lib = self.lib
        if self._userblock() != 0:
            #  chunk addresses of H5Dget_chunk_info are relative to the file's base address in libhdf5 builds before
            # 1.14.4 — behind a user block a pread at them fetches other bytes, and an unfiltered chunk would not even fail to
            # inflate.  Such files (rare: cooler never writes one) go through H5Dread.
            return None
        did = lib.H5Dopen2(self.fid, name.encode(), _H5P_DEFAULT)
        if did < 0:
            raise KeyError(f"dataset {name!r} not found")
        tid = pl = -1
        try:
            sid = lib.H5Dget_space(did)
            n = lib.H5Sget_simple_extent_npoints(sid)
            lib.H5Sclose(sid)
            tid = lib.H5Dget_type(did)
            cls, size = lib.H5Tget_class(tid), lib.H5Tget_size(tid)
            if cls not in (_H5T_INTEGER, _H5T_FLOAT) or size not in (4, 8) or lib.H5Tget_order(tid) != 0:      # (0 = little-endian)
                return None
            if cls == _H5T_INTEGER:
                dtype = np.dtype(("i" if lib.H5Tget_sign(tid) == 1 else "u") + str(size))
            else:
                dtype = np.dtype("f" + str(size))
            out = {"dtype": dtype, "n": int(n)}
            pl = lib.H5Dget_create_plist(did)
            kind = lib.H5Pget_layout(pl)
            nf = lib.H5Pget_nfilters(pl)
            if kind == 1 and nf == 0:                       # H5D_CONTIGUOUS
                off = lib.H5Dget_offset(did)
                if off == 0xFFFFFFFFFFFFFFFF:                # HADDR_UNDEF
                    return None
                out.update(kind="contiguous", offset=int(off))
                return out
            if kind != 2 or not hasattr(lib, "H5Dget_num_chunks"):
                return None
            filters = []                                     # pipeline order (as applied on write): 1 = deflate, 2 = shuffle
            for i in range(nf):
                flags, ncd, cfg = C.c_uint(0), C.c_size_t(0), C.c_uint(0)
                fid = lib.H5Pget_filter2(pl, i, C.byref(flags), C.byref(ncd), None, 0, None, C.byref(cfg))
                if fid not in (1, 2):
                    return None                              # (lzf, szip, ...: libhdf5's own pipeline)
                filters.append(int(fid))
            dims = (C.c_uint64 * 1)(0)
            if lib.H5Pget_chunk(pl, 1, dims) != 1:
                return None
            nchunks = C.c_uint64(0)
            fsp = lib.H5Dget_space(did)
            try:
                if lib.H5Dget_num_chunks(did, fsp, C.byref(nchunks)) < 0:
                    return None
                chunks = []
                off, mask, addr, sz = (C.c_uint64 * 1)(0), C.c_uint(0), C.c_uint64(0), C.c_uint64(0)
                for k in range(int(nchunks.value)):
                    if lib.H5Dget_chunk_info(did, fsp, k, off, C.byref(mask), C.byref(addr), C.byref(sz)) < 0:
                        return None
                    chunks.append((int(off[0]), int(addr.value), int(sz.value), int(mask.value)))
            finally:
                lib.H5Sclose(fsp)
            chunks.sort()
            out.update(kind="chunked", chunk=int(dims[0]), chunks=chunks, filters=filters)
            return out
        finally:
            if pl >= 0:
                lib.H5Pclose(pl)
            if tid >= 0:
                lib.H5Tclose(tid)
            lib.H5Dclose(did)

    def attr_int(self, obj, name):
        lib = self.lib
        aid = lib.H5Aopen_by_name(self.fid, obj.encode(), name.encode(), _H5P_DEFAULT, _H5P_DEFAULT)
        if aid < 0:
            raise KeyError(f"attribute {name!r} not found on {obj!r}")
        v = C.c_int64(0)
        rc = lib.H5Aread(aid, _native(lib, "H5T_NATIVE_INT64_g"), C.byref(v))
        lib.H5Aclose(aid)
        if rc < 0:
            raise OSError(f"cannot read attribute {name!r}")
        return int(v.value)


class StreamedCooler(ArrayCooler):
    """An ArrayCooler whose pixel table stays in the FILE: bins, chromosomes and the row pointer are in memory, pixels/bin2_id
    and pixels/count are streamed to the GPU in row-range chunks through page-locked slabs when an engine is first needed
    (stream_pixels_into) — the table never sits in pageable host memory.  Whoever asks for the arrays themselves (multi-rank
    row subsets, the test oracle) gets them read in full, once."""

    def __init__(self, chromsizes, binsize, bin1_offset, source, bins=None, filename="in_memory.cool"):
        self._pixel_source = source                      # {"path", "group", "nnz", "bin2_dtype", "count_dtype"}
        self._b2 = self._ct = None
        super().__init__(chromsizes, binsize, bin1_offset, np.empty(0, np.int32), np.empty(0, np.int32), bins=bins, filename=filename)
        self._b2 = self._ct = None

    def _load_pixels(self):
        if self._b2 is None:
            src = self._pixel_source
            f = _File(src["path"])
            try:
                self._b2 = f.read(f"{src['group']}/pixels/bin2_id")
                self._ct = _counts32(f.read(f"{src['group']}/pixels/count"))
            finally:
                f.close()

    @property
    def pixels_in_memory(self):
        return self._b2 is not None

    @property
    def bin2_id(self):
        self._load_pixels()
        return self._b2

    @bin2_id.setter
    def bin2_id(self, v):
        self._b2 = v

    @property
    def count(self):
        self._load_pixels()
        return self._ct

    @count.setter
    def count(self, v):
        self._ct = v

    @property
    def nnz(self):
        return int(self._pixel_source["nnz"])


class _DirectReader:
    """Elements [first, first + m) of a 1-D dataset into a numpy array WITHOUT libhdf5 on the data path: its calls hold a global
    lock, so one thread reading hyperslabs (~10 GB/s from the page cache) was 0.37 of the 0.51 s a first pile-up from a .cool file
    took — the H2D copies themselves 0.07 s.  Contiguous datasets are pread() straight into the destination in pieces on several
    threads; chunked ones (cooler's default: gzip + shuffle) have their stored chunks pread, inflated (zlib releases the interpreter
    lock), un-shuffled and copied, a chunk per task.  Anything _File.layout does not describe goes through H5Dread as before."""
    PIECE = 8 << 20                                      # bytes per pread task of a contiguous dataset

    def __init__(self, path, h5file, name, threads=8):
        self.name, self.h5 = name, h5file
        self.lay = h5file.layout(name)
        self.fd = os.open(path, os.O_RDONLY) if self.lay is not None else -1
        self.pool = None
        if self.lay is not None:
            from concurrent.futures import ThreadPoolExecutor
            self.pool = ThreadPoolExecutor(max_workers=threads, thread_name_prefix="coolpuppy_amd-read")

    def close(self):
        if self.pool is not None:
            self.pool.shutdown(wait=True)
            self.pool = None
        if self.fd >= 0:
            os.close(self.fd)
            self.fd = -1

    def _contiguous(self, first, out):
        lay = self.lay
        if out.dtype == lay["dtype"]:
            raw = out
        else:
            raw = np.empty(out.shape[0], lay["dtype"])
        item = lay["dtype"].itemsize
        mv = memoryview(raw).cast("B")
        total = raw.shape[0] * item
        base = lay["offset"] + int(first) * item

        def task(a):
            b = min(a + self.PIECE, total)
            while a < b:
                got = os.preadv(self.fd, [mv[a:b]], base + a)
                if got <= 0:
                    raise OSError(f"short read of {self.name!r}")
                a += got
        list(self.pool.map(task, range(0, total, self.PIECE)))
        if raw is not out:
            _narrow_into(out, raw)

    def _chunked(self, first, out):
        import zlib
        lay = self.lay
        ce, item = lay["chunk"], lay["dtype"].itemsize
        last = int(first) + out.shape[0]
        k0, k1 = int(first) // ce, (last + ce - 1) // ce
        starts = [c[0] for c in lay["chunks"]]
        import bisect
        todo = lay["chunks"][bisect.bisect_left(starts, k0 * ce):bisect.bisect_left(starts, k1 * ce)]

        def task(c):
            start, addr, nbytes, mask = c
            buf = os.pread(self.fd, nbytes, addr)
            if len(buf) != nbytes:
                raise OSError(f"short read of a chunk of {self.name!r}")
            n_el = min(ce, lay["n"] - start)
            # the pipeline backwards; bit i of the chunk's mask = filter i was skipped when the chunk was written
            for idx in range(len(lay["filters"]) - 1, -1, -1):
                if mask & (1 << idx):
                    continue
                if lay["filters"][idx] == 1:
                    buf = zlib.decompress(buf)
                else:
                    buf = np.frombuffer(buf, np.uint8, count=ce * item).reshape(item, ce).T.tobytes()     # byte planes back into elements
            if len(buf) != ce * item:                               # (a chunk always holds `chunk` elements, edge chunks too)
                raise OSError(f"{self.name!r}: a chunk unpacked to {len(buf)} bytes, expected {ce * item}")
            vals = np.frombuffer(buf, lay["dtype"], count=ce)[:n_el]
            a, b = max(start, int(first)), min(start + n_el, last)
            if b > a:
                _narrow_into(out[a - int(first):b - int(first)], vals[a - start:b - start])
        list(self.pool.map(task, todo))
        have = sum(min(c[0] + ce, last) - max(c[0], int(first)) for c in todo if min(c[0] + ce, last) > max(c[0], int(first)))
        if have != out.shape[0]:                                     # unallocated chunks read as the fill value (zeros)
            raise OSError(f"{self.name!r}: chunks missing under [{first}, {last})")

    def read_into(self, first, out):
        if self.lay is None:
            return self.h5.read_into(self.name, first, out)
        if self.lay["kind"] == "contiguous":
            return self._contiguous(first, out)
        return self._chunked(first, out)


def _narrow_into(out, vals):
    """out[:] = vals for arrays of different integer width: range-checked, never wrapped."""
    if out.dtype == vals.dtype:
        out[:] = vals
        return
    if out.dtype.kind in "iu" and vals.dtype.kind in "iu" and vals.size:
        info = np.iinfo(out.dtype)
        if int(vals.max()) > info.max or int(vals.min()) < info.min:
            raise OverflowError("pixel counts outside 0 .. 2^31-1 do not fit the engine's int32 pixel table")
    out[:] = vals


def stream_pixels_into(eng, clr, slab_pixels=0):
    """Upload the pixel table of a StreamedCooler from its file: the library's page-locked slabs are filled by hyperslab reads of
    pixels/bin2_id and pixels/count (pup_load_pixels_stream), each slab copied asynchronously while the next is read.
    Returns the copy statistics of PileupEngine.load_pixels_stream."""
    src = clr._pixel_source
    f = _File(src["path"])
    g = src["group"]
    cdt = np.dtype(src["count_dtype"])
    if cdt.kind not in "iu":
        # a float pixels/count column: the whole table through host memory (PileupEngine.load_pixels takes float values — as int32
        # when they are whole numbers, else as float64 beside the table: pup_load_pixel_values)
        try:
            bin2 = f.read(f"{g}/pixels/bin2_id")
            vals = f.read(f"{g}/pixels/count")
        finally:
            f.close()
        eng.load_pixels(clr.bin1_offset, bin2, vals)
        return {"h2d_ms": None, "h2d_bytes": None, "h2d_GBps": None}
    b2dt = np.dtype(np.int64) if np.dtype(src["bin2_dtype"]).itemsize == 8 else np.dtype(np.int32)
    direct = os.environ.get("COOLPUPPY_AMD_DIRECT_READS", "1") != "0"
    threads = int(os.environ.get("COOLPUPPY_AMD_READ_THREADS", "8"))
    rd_col = _DirectReader(src["path"], f, f"{g}/pixels/bin2_id", threads) if direct else None
    rd_cnt = _DirectReader(src["path"], f, f"{g}/pixels/count", threads) if direct else None

    def fill(first, m, colv, cntv):
        if rd_col is not None and rd_col.lay is not None and rd_cnt.lay is not None and rd_cnt.lay["dtype"].kind in "iu":
            # both columns of the slab at once: their pieces share the readers' threads
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=2) as two:
                a = two.submit(rd_col.read_into, first, colv)
                b = two.submit(rd_cnt.read_into, first, cntv)
                a.result(); b.result()
            return
        f.read_into(f"{g}/pixels/bin2_id", first, colv)
        if cdt == np.dtype(np.int32):
            f.read_into(f"{g}/pixels/count", first, cntv)
        else:                                            # wider / unsigned counts: range-checked, then narrowed (never wrapped)
            tmp = np.empty(m, np.int64)
            f.read_into(f"{g}/pixels/count", first, tmp)
            cntv[:] = _counts32(tmp)
    try:
        return eng.load_pixels_stream(clr.bin1_offset, src["nnz"], b2dt, fill, slab_pixels=slab_pixels)
    finally:
        for r in (rd_col, rd_cnt):
            if r is not None:
                r.close()
        f.close()


def read_cool(path, group="/", extra_bins=None, stream_pixels=False):
    """Open ``path`` (optionally a ``group`` inside a multi-resolution file) and return an ArrayCooler.

    Every numeric column of ``bins/`` (or just those listed in ``extra_bins``) is loaded.  stream_pixels=False: the whole pixel
    table is read into host memory.  stream_pixels=True: a StreamedCooler — the pixel table goes from the file to the GPU in
    chunks through page-locked memory when the first pile-up needs it (the engine keeps it in HBM afterwards)."""
    if stream_pixels:
        g = "/" + group.strip("/")
        g = "" if g == "/" else g
        f = _File(path)
        try:
            names = [x.decode() if isinstance(x, bytes) else str(x) for x in f.read(f"{g}/chroms/name")]
            names = [x.rstrip("\x00") for x in names]
            lengths = f.read(f"{g}/chroms/length").astype(np.int64)
            binsize = f.attr_int(g or "/", "bin-size")
            bin1_offset = f.read(f"{g}/indexes/bin1_offset").astype(np.int64)
            nnz, b2dt = f.dataset_info(f"{g}/pixels/bin2_id")
            _, cdt = f.dataset_info(f"{g}/pixels/count")
            cols = {}
            for c in _bins_columns(f.children(f"{g}/bins"), extra_bins):
                v = f.read(f"{g}/bins/{c}")
                if v.dtype.kind in "iuf":
                    cols[c] = v
        finally:
            f.close()
        return StreamedCooler(pd.Series(lengths, index=names), binsize, bin1_offset,
                              {"path": path, "group": g, "nnz": nnz, "bin2_dtype": str(b2dt), "count_dtype": str(cdt)}, bins=cols, filename=path)
    try:
        import h5py  # noqa: F401
        return _read_cool_h5py(path, group, extra_bins)
    except ImportError:
        pass
    g = "/" + group.strip("/")
    g = "" if g == "/" else g
    f = _File(path)
    try:
        names = [x.decode() if isinstance(x, bytes) else str(x) for x in f.read(f"{g}/chroms/name")]
        names = [x.rstrip("\x00") for x in names]
        lengths = f.read(f"{g}/chroms/length").astype(np.int64)
        binsize = f.attr_int(g or "/", "bin-size")
        bin1_offset = f.read(f"{g}/indexes/bin1_offset").astype(np.int64)
        bin2_id = f.read(f"{g}/pixels/bin2_id")
        count = f.read(f"{g}/pixels/count")
        cols = {}
        for c in _bins_columns(f.children(f"{g}/bins"), extra_bins):
            v = f.read(f"{g}/bins/{c}")
            if v.dtype.kind in "iuf":
                cols[c] = v
    finally:
        f.close()
    return ArrayCooler(pd.Series(lengths, index=names), binsize, bin1_offset, bin2_id, _counts32(count), bins=cols,
                       filename=path)


def _counts32(count):
    """pixels/count as the engine's int32 — refusing, not wrapping, what does not fit.  A float column is kept as it is
    (PileupEngine.load_pixels uploads it as float64 pixel values)."""
    if count.dtype.kind == "f":
        return count
    if count.dtype.itemsize > 4 and count.size and (int(count.max()) > 2**31 - 1 or int(count.min()) < 0):
        raise OverflowError("pixel counts outside 0 .. 2^31-1 do not fit the engine's int32 pixel table")
    return count.astype(np.int32)


def _bins_columns(names_in_file, extra_bins):
    """Which bins/ columns to load: the caller's list, else every one that is not part of the bin table itself (so that
    coverage_norm= / clr_weight_name= can name any numeric column of the file, as with cooler)."""
    if extra_bins is not None:
        return [c for c in extra_bins if c in names_in_file]
    return [c for c in names_in_file if c not in ("chrom", "start", "end")]


def _read_cool_h5py(path, group, extra_bins):
    import h5py
    with h5py.File(path, "r") as h5:
        g = h5[group]
        names = [x.decode() if isinstance(x, bytes) else str(x) for x in g["chroms/name"][:]]
        lengths = g["chroms/length"][:].astype(np.int64)
        binsize = int(g.attrs["bin-size"])
        cols = {c: g["bins"][c][:] for c in _bins_columns(list(g["bins"].keys()), extra_bins)}
        cols = {c: v for c, v in cols.items() if v.dtype.kind in "iuf"}
        return ArrayCooler(pd.Series(lengths, index=names), binsize, g["indexes/bin1_offset"][:].astype(np.int64),
                           g["pixels/bin2_id"][:], _counts32(g["pixels/count"][:]), bins=cols, filename=path)


def write_cool(path, clr, group="/", chunks=None, gzip=None, shuffle=False, userblock=0):
    """Write an ArrayCooler as a single-resolution ``.cool`` (the datasets read_cool consumes, cooler's schema: chroms/{name,length},
    bins/{chrom,start,end,<columns>}, pixels/{bin1_id,bin2_id,count}, indexes/{bin1_offset,chrom_offset}, attrs bin-size /
    format / nbins / nnz).  A utility for tests and benchmarks (the reference only ever reads coolers; `cooler` writes them):
    chunks / gzip as cooler does when given, else contiguous datasets; userblock: bytes of HDF5 user block in front (tests)."""
    from .lib.io import _H5
    g = "/" + group.strip("/")
    g = "" if g == "/" else g
    indptr, col, cnt = clr.pixel_table()
    with _H5(path, "w", userblock=userblock) as h5:
        cur = ""
        for part in [p for p in g.split("/") if p]:
            cur += "/" + part
            h5.group(cur)
        for sub in ("chroms", "bins", "pixels", "indexes"):
            h5.group(f"{g}/{sub}")
        h5.write_fixed_strings(f"{g}/chroms/name", [str(c) for c in clr.chromnames])
        h5.write(f"{g}/chroms/length", clr.chromsizes.values.astype(np.int32))
        nb = np.diff(clr.chrom_offset)
        b = clr.bins()
        h5.write(f"{g}/bins/chrom", np.repeat(np.arange(len(nb), dtype=np.int32), nb))
        h5.write(f"{g}/bins/start", b["start"][:].values.astype(np.int32))
        h5.write(f"{g}/bins/end", b["end"][:].values.astype(np.int32))
        for c in b.columns:
            if c not in ("chrom", "start", "end"):
                h5.write(f"{g}/bins/{c}", np.asarray(b[c][:].values))
        ck = None if chunks is None else (int(chunks),)
        h5.write(f"{g}/pixels/bin1_id", np.repeat(np.arange(clr.nbins, dtype=np.int64), np.diff(indptr)), chunks=ck, gzip=gzip, shuffle=shuffle)
        h5.write(f"{g}/pixels/bin2_id", np.asarray(col, np.int64), chunks=ck, gzip=gzip, shuffle=shuffle)
        h5.write(f"{g}/pixels/count", np.asarray(cnt) if np.asarray(cnt).dtype.kind == "f" else np.asarray(cnt, np.int32), chunks=ck, gzip=gzip, shuffle=shuffle)
        h5.write(f"{g}/indexes/bin1_offset", np.asarray(indptr, np.int64))
        h5.write(f"{g}/indexes/chrom_offset", np.asarray(clr.chrom_offset, np.int64))
        root = g or "/"
        h5.set_attr(root, "bin-size", int(clr.binsize))
        h5.set_attr(root, "format", "HDF5::Cooler")
        h5.set_attr(root, "nbins", int(clr.nbins))
        h5.set_attr(root, "nnz", int(len(col)))
