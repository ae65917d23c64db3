"""Python face of the HIP pile-up engine: one :class:`PileupEngine` per GPU.

Thin, numpy-in / numpy-out wrapper over the C ABI (``include/pup_hip.h``).  It owns no algorithm:
everything that touches pixels runs in ``libpup_hip.so``.  What it replaces in the reference is the
inner part of ``PileUpper.pileup_region`` — ``get_data`` + ``_stream_snips`` + ``accumulate_stream``
(reference coolpuppy/coolpup.py:1024-1057, 1059-1191, 1236-1283).
"""
import ctypes as C
import sys
import threading
import os
import weakref

import numpy as np

from . import _ffi
from ._ffi import MODE_COV, MODE_DEVPTR, MODE_EXPECTED, MODE_OOE, MODE_TRANSPOSE, PupError  # noqa: F401


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _as(a, dtype):
    return np.ascontiguousarray(a, dtype=dtype)


_LIVE = weakref.WeakSet()      # engines not closed yet: coolpuppy_amd.shutdown() closes them before the HIP runtime goes


def close_all():
    for eng in list(_LIVE):
        try:
            eng.close()
        except Exception:       # noqa: BLE001 - shutting down
            pass


class PileupEngine:
    """Device-resident pixel table + running (kind, group) accumulators on one MI355X."""

    def __init__(self, device_id=0):
        self._lib = _ffi.lib()
        h = C.c_void_p()
        rc = self._lib.pup_create(int(device_id), C.byref(h))
        if rc != 0:
            raise PupError(rc, self._lib.pup_last_error(None).decode())
        self._h = h
        _LIVE.add(self)
        self.device_id = int(device_id)
        self.n_tiles = 0
        self.pad = 0
        self.nbins = 0
        self.nnz = 0

    # -- plumbing -------------------------------------------------------------------------------------
    def _check(self, rc):
        if rc != 0:
            raise PupError(rc, self._lib.pup_last_error(self._h).decode())

    def close(self):
        if getattr(self, "_h", None):
            self._lib.pup_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    @property
    def W(self):
        return 2 * self.pad + 1

    # -- inputs ---------------------------------------------------------------------------------------
    def load_pixels(self, bin1_offset, bin2_id, count):
        """Upper-triangular pixel table as stored in a .cool (indexes/bin1_offset, pixels/bin2_id, pixels/count)."""
        bin1_offset = _as(bin1_offset, np.int64)
        bin2_id = np.ascontiguousarray(bin2_id)
        if bin2_id.dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
            bin2_id = bin2_id.astype(np.int64)
        count = np.ascontiguousarray(count)
        values = None
        if count.dtype.kind == "f":
            # a float pixels/count column (cooler allows it; the reference multiplies it through, coolpup.py:1053-1057): whole
            # numbers in range ARE counts — every kernel serves them; anything else goes up as float64 pixel values beside the
            # table (pup_load_pixel_values) and is piled up by the kernels on the balanced value table
            whole = count.size == 0 or (bool(np.all(np.isfinite(count))) and bool(np.all(count == np.floor(count)))
                                        and 0 <= float(count.min()) and float(count.max()) <= 2**31 - 1)
            if whole:
                count = count.astype(np.int32)
            else:
                values = np.ascontiguousarray(count, np.float64)
                count = np.zeros(count.shape[0], np.int32)
        if count.dtype != np.int32:
            if not np.issubdtype(count.dtype, np.integer):
                raise PupError(-6, f"pixel counts of dtype {count.dtype} are not supported (integers or floats expected)")
            if count.size and (int(count.max()) > 2**31 - 1 or int(count.min()) < 0):
                raise PupError(-5, "pixel counts outside 0 .. 2^31-1 do not fit the engine's int32 pixel table")
            count = count.astype(np.int32)
        nbins = bin1_offset.shape[0] - 1
        nnz = bin2_id.shape[0]
        if count.shape[0] != nnz:
            raise ValueError("bin2_id and count differ in length")
        self._check(self._lib.pup_load_pixels(self._h, _ptr(bin1_offset), _ptr(bin2_id), bin2_id.dtype.itemsize,
                                              _ptr(count), nbins, nnz))
        if values is not None:
            self._check(self._lib.pup_load_pixel_values(self._h, _ptr(values), nnz))
        self.nbins, self.nnz = nbins, nnz
        self.float_values = values is not None

    def load_pixels_stream(self, bin1_offset, nnz, bin2_dtype, fill, slab_pixels=0):
        """The pixel table streamed in (pup_load_pixels_stream): ``fill(first, m, bin2_view, count_view)`` writes pixels
        [first, first + m) into two numpy views of a PAGE-LOCKED slab of the library (bin2_view: dtype bin2_dtype, count_view:
        int32) — e.g. straight from hyperslabs of a .cool file — while the previous slab is on its way to the GPU.
        Returns {"h2d_ms", "h2d_bytes", "h2d_GBps"} of the copies alone."""
        bin1_offset = _as(bin1_offset, np.int64)
        bin2_dtype = np.dtype(bin2_dtype)
        if bin2_dtype not in (np.dtype(np.int32), np.dtype(np.int64)):
            raise ValueError("bin2_id must be streamed as int32 or int64")
        nbins = bin1_offset.shape[0] - 1
        err = []

        def _cb(user, first, m, p_col, p_cnt):
            try:
                colv = np.ctypeslib.as_array(C.cast(p_col, C.POINTER(C.c_int64 if bin2_dtype.itemsize == 8 else C.c_int32)), shape=(m,))
                cntv = np.ctypeslib.as_array(C.cast(p_cnt, C.POINTER(C.c_int32)), shape=(m,))
                fill(int(first), int(m), colv, cntv)
                return 0
            except Exception as e:      # noqa: BLE001 - must not propagate through the C frame
                err.append(e)
                return 1
        cb = _ffi.FILL_FN(_cb)
        ms, nbytes = C.c_double(0.0), C.c_int64(0)
        rc = self._lib.pup_load_pixels_stream(self._h, _ptr(bin1_offset), nbins, int(nnz), bin2_dtype.itemsize, int(slab_pixels),
                                              C.cast(cb, C.c_void_p), None, C.byref(ms), C.byref(nbytes))
        if err:
            raise err[0]
        self._check(rc)
        self.nbins, self.nnz = nbins, int(nnz)
        self.float_values = False             # (the streamed table is a table of counts: the library has dropped any float values too)
        return {"h2d_ms": ms.value, "h2d_bytes": nbytes.value,
                "h2d_GBps": (nbytes.value / (ms.value * 1e-3) / 1e9) if ms.value > 0 else None}

    def build_index(self, chrom_offset, max_bytes=0):
        """Rank-bitmap index over the cis part of the table (chrom_offset = the cooler's indexes/chrom_offset).
        Returns True when built, False when it does not fit (the engine then keeps using binary search)."""
        co = _as(chrom_offset, np.int64)
        rc = self._lib.pup_build_index(self._h, _ptr(co), co.shape[0] - 1, int(max_bytes))
        if rc == -2:          # PUP_ENOMEM: optional structure, not an error for the caller
            return False
        self._check(rc)
        return True

    def coverage(self, chrom_offset, ignore_diags=2):
        """(cov_cis_raw, cov_tot_raw) of the loaded table — cooltools.coverage semantics, see pup_coverage."""
        co = _as(chrom_offset, np.int64)
        cis = np.empty(self.nbins, np.float64)
        tot = np.empty(self.nbins, np.float64)
        self._check(self._lib.pup_coverage(self._h, _ptr(co), co.shape[0] - 1, int(ignore_diags), _ptr(cis), _ptr(tot)))
        return cis, tot

    def load_bins(self, weight=None, cov=None):
        """Per-bin float64 vectors: balancing weights (NaN = masked; None = raw) and coverage (None = unused)."""
        w = None if weight is None else _as(weight, np.float64)
        c = None if cov is None else _as(cov, np.float64)
        for v in (w, c):
            if v is not None and v.shape[0] != self.nbins:
                raise ValueError(f"bin vector has {v.shape[0]} entries, table has {self.nbins} bins")
        self._check(self._lib.pup_load_bins(self._h, _ptr(w), _ptr(c)))

    def set_expected(self, expected=None):
        """By-diagonal expected vector (cis), a scalar (trans) or None."""
        if expected is None:
            self._check(self._lib.pup_set_expected(self._h, None, 0))
            return
        e = np.atleast_1d(_as(expected, np.float64))
        self._check(self._lib.pup_set_expected(self._h, _ptr(e), e.shape[0]))

    def set_expected_table(self, start, end, vectors=None, pair=None):
        """Expected of many regions at once. start/end: global bin ranges (sorted, disjoint).
        vectors: list of by-diagonal vectors, one per region (cis);  pair: [n, n] matrix of scalars (trans)."""
        start, end = _as(start, np.int32), _as(end, np.int32)
        n = start.shape[0]
        if pair is not None:
            pm = _as(pair, np.float64).reshape(n, n)
            self._check(self._lib.pup_set_expected_table(self._h, _ptr(start), _ptr(end), None, None, n, None, 0, _ptr(pm)))
            return
        lens = np.array([len(v) for v in vectors], np.int64)
        offs = np.concatenate([[0], np.cumsum(lens)[:-1]]).astype(np.int64)
        vals = _as(np.concatenate([np.asarray(v, np.float64) for v in vectors]) if n else np.zeros(1), np.float64)
        self._check(self._lib.pup_set_expected_table(self._h, _ptr(start), _ptr(end), _ptr(offs), _ptr(lens), n,
                                                     _ptr(vals), vals.shape[0], None))

    # -- accumulators -----------------------------------------------------------------------------------
    def reset(self, n_tiles, pad):
        self._check(self._lib.pup_reset(self._h, int(n_tiles), int(pad)))
        self.n_tiles, self.pad = int(n_tiles), int(pad)

    def accumulate(self, r0, c0, tile_ptr, *, flip_from=None, ignore_diags=2, mode=0):
        """Accumulate tile-grouped snippets given by their top-left GLOBAL bins (host arrays).
        flip_from[t] (optional): first anti-transposed snippet of tile t (flipped ones come last in a tile)."""
        r0 = _as(r0, np.int32)
        c0 = _as(c0, np.int32)
        tile_ptr = _as(tile_ptr, np.int64)
        if tile_ptr.shape[0] != self.n_tiles + 1:
            raise ValueError(f"tile_ptr needs {self.n_tiles + 1} entries")
        ff = None if flip_from is None else _as(flip_from, np.int64)
        if ff is not None and ff.shape[0] != self.n_tiles:
            raise ValueError(f"flip_from needs {self.n_tiles} entries")
        self._check(self._lib.pup_accumulate(self._h, _ptr(r0), _ptr(c0), r0.shape[0], _ptr(tile_ptr), _ptr(ff),
                                             int(ignore_diags), int(mode) & ~MODE_DEVPTR))

    def accumulate_rescaled(self, r0, c0, height, width, tile_ptr, *, flip_from=None, ignore_diags=2, mode=0):
        """Rescaled pile-up: variable height x width windows zoomed to the W x W tile (W = 2*pad+1 = rescale_size)."""
        r0, c0 = _as(r0, np.int32), _as(c0, np.int32)
        hh, ww = _as(height, np.int32), _as(width, np.int32)
        tile_ptr = _as(tile_ptr, np.int64)
        if tile_ptr.shape[0] != self.n_tiles + 1:
            raise ValueError(f"tile_ptr needs {self.n_tiles + 1} entries")
        ff = None if flip_from is None else _as(flip_from, np.int64)
        self._check(self._lib.pup_accumulate_rescaled(self._h, _ptr(r0), _ptr(c0), _ptr(hh), _ptr(ww), r0.shape[0],
                                                      _ptr(tile_ptr), _ptr(ff), int(ignore_diags),
                                                      int(mode) & ~MODE_DEVPTR))

    def accumulate_device(self, r0_ptr, c0_ptr, n, tile_ptr, *, flip_from=None, ignore_diags=2, mode=0):
        """Same, with r0/c0 already resident in HBM (raw device addresses, e.g. tensor.data_ptr())."""
        tile_ptr = _as(tile_ptr, np.int64)
        ff = None if flip_from is None else _as(flip_from, np.int64)
        self._check(self._lib.pup_accumulate(self._h, C.c_void_p(r0_ptr), C.c_void_p(c0_ptr), int(n),
                                             _ptr(tile_ptr), _ptr(ff), int(ignore_diags), int(mode) | MODE_DEVPTR))

    def stripes(self, r0, c0, pad, *, ignore_diags=2, mode=0):
        """(horizontal [n,W], vertical [n,W]) centre row / reversed centre column of every snippet (store_stripes)."""
        r0 = _as(r0, np.int32)
        c0 = _as(c0, np.int32)
        W = 2 * int(pad) + 1
        h = np.empty((r0.shape[0], W), np.float64)
        v = np.empty((r0.shape[0], W), np.float64)
        self._check(self._lib.pup_stripes(self._h, _ptr(r0), _ptr(c0), r0.shape[0], int(pad), int(ignore_diags),
                                          int(mode), _ptr(h), _ptr(v)))
        return h, v

    def extract(self, r0, c0, pad, *, height=None, width=None, ignore_diags=2, mode=0, coverage=False):
        """Per-snippet windows for host callbacks: data [n,W,W] as _stream_snips yields them (NaN-masked, /expected with
        MODE_OOE, the expected window with MODE_EXPECTED; rescaled to W x W when height/width are given).  With
        coverage=True also returns (cov_start [n,W], cov_end [n,W]).  Nothing is accumulated."""
        r0, c0 = _as(r0, np.int32), _as(c0, np.int32)
        hh = None if height is None else _as(height, np.int32)
        ww = None if width is None else _as(width, np.int32)
        n, W = r0.shape[0], 2 * int(pad) + 1
        data = np.empty((n, W, W), np.float64)
        cs = np.empty((n, W), np.float64) if coverage else None
        ce = np.empty((n, W), np.float64) if coverage else None
        self._check(self._lib.pup_extract(self._h, _ptr(r0), _ptr(c0), _ptr(hh), _ptr(ww), n, int(pad), int(ignore_diags),
                                          int(mode) & ~MODE_DEVPTR, _ptr(data), _ptr(cs), _ptr(ce)))
        return (data, cs, ce) if coverage else data

    def sync(self):
        self._check(self._lib.pup_sync(self._h))

    def fetch(self):
        """dict(sum [T,W,W] f64, num [T,W,W] i64, n [T] i64, cov_start [T,W], cov_end [T,W])."""
        T, W = self.n_tiles, self.W
        out = {
            "sum": np.empty((T, W, W), np.float64),
            "num": np.empty((T, W, W), np.int64),
            "n": np.empty((T,), np.int64),
            "cov_start": np.empty((T, W), np.float64),
            "cov_end": np.empty((T, W), np.float64),
        }
        self._check(self._lib.pup_fetch(self._h, _ptr(out["sum"]), _ptr(out["num"]), _ptr(out["n"]),
                                        _ptr(out["cov_start"]), _ptr(out["cov_end"])))
        return out

    # -- cross-GPU ----------------------------------------------------------------------------------------
    def packed_sizes(self):
        a, b = C.c_int64(), C.c_int64()
        self._check(self._lib.pup_packed_sizes(self._h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def export_to(self, dev_f64_ptr, dev_i64_ptr):
        self._check(self._lib.pup_export(self._h, C.c_void_p(dev_f64_ptr), C.c_void_p(dev_i64_ptr)))

    def import_from(self, dev_f64_ptr, dev_i64_ptr):
        self._check(self._lib.pup_import(self._h, C.c_void_p(dev_f64_ptr), C.c_void_p(dev_i64_ptr)))

    # -- measurement --------------------------------------------------------------------------------------
    def allreduce(self, rccl_comm):
        """In-place RCCL all-reduce of the packed accumulators over an ncclComm_t (int address or c_void_p), asynchronous
        on the engine's stream — the torch-free form of export_to / all_reduce / import_from."""
        addr = rccl_comm.value if isinstance(rccl_comm, C.c_void_p) else int(rccl_comm)
        self._check(self._lib.pup_allreduce(self._h, C.c_void_p(addr)))

    def tile_block_sizes(self, n):
        """(f64 elements, i64 elements) of a block of n packed tiles (pup_pack_tiles layout)."""
        W2 = self.W * self.W
        return n * (W2 + 2 * self.W), n * (W2 + 1)

    def pack_tiles(self, tile_ids, dev_f64_ptr, dev_i64_ptr):
        ids = _as(tile_ids, np.int32)
        self._check(self._lib.pup_pack_tiles(self._h, _ptr(ids), ids.shape[0], C.c_void_p(dev_f64_ptr), C.c_void_p(dev_i64_ptr)))

    def unpack_tiles(self, tile_ids, dev_f64_ptr=None, dev_i64_ptr=None, mode=1):
        """mode 0: overwrite the tiles from the block, 1: add the block to them, 2: clear them."""
        ids = _as(tile_ids, np.int32)
        self._check(self._lib.pup_unpack_tiles(self._h, _ptr(ids), ids.shape[0], C.c_void_p(dev_f64_ptr or 0), C.c_void_p(dev_i64_ptr or 0), int(mode)))

    def allgather_tiles(self, rccl_comm, tile_ids, rank_ptr, my_rank):
        """pup_allgather_tiles: tile_ids[rank_ptr[r]:rank_ptr[r+1]] = the tiles rank r holds; afterwards every listed tile is the
        sum over the ranks that list it (asynchronous on the engine's stream)."""
        addr = rccl_comm.value if isinstance(rccl_comm, C.c_void_p) else int(rccl_comm)
        ids, ptr = _as(tile_ids, np.int32), _as(rank_ptr, np.int64)
        self._check(self._lib.pup_allgather_tiles(self._h, C.c_void_p(addr), _ptr(ids), _ptr(ptr), ptr.shape[0] - 1, int(my_rank)))

    def set_profiling(self, enabled=True):
        """True / 1: HIP-event timing of the kernels + pixel statistics; 3: timing only; False / 0: off."""
        self._check(self._lib.pup_set_profiling(self._h, int(enabled)))

    def set_tuning(self, chunk_snippets=0, variant=0):
        """chunk_snippets: snippets per chunk of the per-window kernels.  variant bits: 1 ignore the index, 2 LDS-tile
        kernel, 4 no factorised num, 8 force / 16 forbid the workgroup-staged kernel, 32 no sparse trans kernel, 64 no tile
        pairing (128: no effect since round 3) — see pup_set_tuning in pup_hip.h."""
        self._check(self._lib.pup_set_tuning(self._h, int(chunk_snippets), int(variant)))

    @staticmethod
    def staged_region(pad, ooe=False, extra=False):
        """(rows, columns) of the region the workgroup-staged kernel keeps in LDS — staged_geometry() of pup_engine.hip /
        StagedGeom of pup_staged.hpp: 128 x 128 bins for windows up to 21 bins, 64 x 128 otherwise (coverage / statistics riding along,
        wider windows: their register budget)."""
        big = 2 * int(pad) + 1 <= 21 and not extra
        return (128 if big else 64), 128

    @staticmethod
    def wide_geometry(W):
        """Sub-window grid of the wide staged kernel for window width W — wide_geometry() of csrc/pup_wide.hpp."""
        best = None
        for k in range(9):                                   # the lane shapes (cells per lane, column chunks per row): wide_shape_ch / _nch
            ch, nch = (7 + k, 4) if k < 7 else ((11, 5) if k == 7 else (7, 6))
            max_rows, max_cols = min(256 // nch, 64), ch * nch
            ngr, ngc = -(-W // max_rows), -(-W // max_cols)
            cost = ngr * ngc * 4 * ch
            ng_best = None if best is None else best[1]["NGr"] * best[1]["NGc"]
            if best is None or ngr * ngc < ng_best or (ngr * ngc == ng_best and cost < best[0]):      # fewest groups, then fewest LDS instructions
                best = (cost, {"NGr": ngr, "NGc": ngc, "SH": -(-W // ngr), "SW": -(-W // ngc), "NPC": 4, "CH": ch, "NCH": nch, "shape": k})
        return best[1]

    @staticmethod
    def block_order(r0, c0, chrom_offset, tile=None, block=None, pad=10, ooe=False, extra=False):
        """Permutation that puts snippets in the order the workgroup-staged kernel walks them inside every tile:
        (tile, block row, block column, r0, c0), blocks of (rows - W + 1) x (columns - W + 1) top-left corners
        (W = 2*pad+1, region as staged_region says) anchored at the chromosome start.  Host mirror of the device-side
        block sort (tests, and bench.py's "preblocked" measurement)."""
        W = 2 * int(pad) + 1
        rows, cols = PileupEngine.staged_region(pad, ooe, extra)
        br_size, bc_size = block or (rows - W + 1, cols - W + 1)
        r0 = np.asarray(r0, np.int64)
        c0 = np.asarray(c0, np.int64)
        co = np.asarray(chrom_offset, np.int64)
        start = co[np.clip(np.searchsorted(co, r0, side="right") - 1, 0, len(co) - 2)]
        br = start + (r0 - start) // br_size
        bc = (c0 - start) // bc_size
        keys = (c0, r0, bc, br) if tile is None else (c0, r0, bc, br, np.asarray(tile))
        return np.lexsort(keys)

    def stats(self):
        s = _ffi.PupStats()
        self._check(self._lib.pup_get_stats(self._h, C.byref(s)))
        return {k: getattr(s, k) for k, _ in _ffi.PupStats._fields_}

    def debug_timing(self):
        """Per-wave phase clocks of the staged kernel's last launch, [workgroups, 16, 8] (set_tuning variant bit 26), or None."""
        buf = np.zeros(4096 * 16 * 8, np.int64)
        g = self._lib.pup_debug_timing(self._h, _ptr(buf), buf.shape[0])
        if g < 0:
            self._check(g)
        return None if g == 0 else buf[:g * 16 * 8].reshape(g, 16, 8)

    def last_kernel(self):
        """Kernel family that served the last accumulate call (pup_last_kernel): 'staged', 'wide', 'wide_fact', 'regtile', 'band', ..."""
        return self._lib.pup_last_kernel(self._h).decode()

    def last_prepass(self):
        """How the last staged call got its block order (pup_last_prepass): 'binning', 'library_sort', or '' (not staged)."""
        return self._lib.pup_last_prepass(self._h).decode()

    def clear_stats(self):
        self._check(self._lib.pup_clear_stats(self._h))

    def event_record(self, slot):
        self._check(self._lib.pup_event_record(self._h, int(slot)))

    def event_elapsed_ms(self, a, b):
        ms = C.c_float()
        self._check(self._lib.pup_event_elapsed_ms(self._h, int(a), int(b), C.byref(ms)))
        return ms.value


def device_count():
    n = _ffi.lib().pup_device_count()
    if n < 0:
        raise PupError(n, _ffi.lib().pup_last_error(None).decode())
    return n


# ---- host-side helpers of the library (no GPU work) ---------------------------------------------------------------------
class _PinnedPool:
    """Page-locked host blocks from pup_host_alloc, recycled: pinning memory costs about as much as copying it, so freed
    blocks are kept (up to `keep` bytes) for the next arrays of the same size class (powers of two from 1 MiB)."""

    def __init__(self, keep=2 << 30):
        self.free, self.kept, self.keep, self.broken = {}, 0, keep, False

    def take(self, nbytes):
        size = 1 << 20
        while size < nbytes:
            size <<= 1
        lst = self.free.get(size)
        if lst:
            self.kept -= size
            return lst.pop(), size
        if self.broken:
            return None, size
        p = C.c_void_p()
        try:
            rc = _ffi.lib().pup_host_alloc(C.byref(p), size)
        except Exception:
            rc = -1
        if rc != 0 or not p.value:
            self.broken = True              # no HIP runtime / no device here: plain numpy arrays from now on
            return None, size
        return p.value, size

    def give(self, ptr, size):
        if self.kept + size <= self.keep:
            self.free.setdefault(size, []).append(ptr)
            self.kept += size
        else:
            _ffi.lib().pup_host_free(C.c_void_p(ptr))


    def drain(self):
        """hipHostFree every kept block (interpreter exit: while the HIP runtime is still loaded)."""
        for size, lst in list(self.free.items()):
            while lst:
                try:
                    _ffi.lib().pup_host_free(C.c_void_p(lst.pop()))
                except Exception:       # noqa: BLE001 - shutting down
                    pass
        self.free, self.kept = {}, 0
        self.keep = 0                   # blocks still held by live arrays are freed, not pooled, when they go


_POOL = _PinnedPool(keep=int(os.environ.get("COOLPUPPY_AMD_PINNED_POOL_MB", "1024")) << 20)


class _PinnedBlock:
    def __init__(self, ptr, size, nbytes):
        self.ptr, self.size = ptr, size
        self.__array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 3}

    def __del__(self):
        try:
            _POOL.give(self.ptr, self.size)
        except Exception:
            pass


def pinned_empty(n, dtype=np.int32):
    """Uninitialised 1-D array in page-locked memory (pup_host_alloc) — pup_accumulate copies such arrays to the device by
    asynchronous DMA.  The block returns to a pool when the array (and every view of it) is gone.  Falls back to an ordinary
    numpy array where the HIP runtime cannot pin memory (no device)."""
    dtype = np.dtype(dtype)
    nbytes = max(int(n) * dtype.itemsize, 1)
    ptr, size = _POOL.take(nbytes)
    if ptr is None:
        return np.empty(int(n), dtype)
    return np.asarray(_PinnedBlock(ptr, size, nbytes)).view(dtype)[:int(n)]


class _WindowArena(threading.local):
    """Per-thread scratch for the per-region window arrays of ONE pile-up (r0 / c0 / group codes of every region, alive until the plan
    has gathered them): carved out of one buffer that is kept between pile-ups — three fresh arrays per region were 130 MB of new pages
    (and their faults, inside the window pass) per 10^7 windows.  reset() at the start of a pile-up reuses the buffer only when no array
    of an earlier pile-up is still alive."""

    def __init__(self):
        self.buf, self.at = np.empty(0, np.int32), 0

    def reset(self):
        # (every array handed out is a view whose .base is the buffer: while any of them is alive the buffer is left to them and a new
        # one of the same size takes over — an earlier plan's windows are never overwritten under its feet)
        if sys.getrefcount(self.buf) > 2:
            self.buf = np.empty(self.buf.shape[0], np.int32)
        self.at = 0

    def take(self, n):
        n = int(n)
        if self.at + n > self.buf.shape[0]:
            self.buf, self.at = np.empty(max(2 * self.buf.shape[0], self.at + n, 1 << 20), np.int32), 0
        out = self.buf[self.at:self.at + n]
        self.at += n
        return out


_ARENA = _WindowArena()


def host_windows(st1, st2, code, shift, sign, nshifts, resolution, off1, off2, lo1, hi1, lo2, hi2, h, w, arena=False):
    """pup_host_windows: ROI windows + shifted control copies of one region, bounds-tested, as (r0, c0, code, n_roi_kept).  st1 / st2 / code: int32 per ROI row (code may be None); shift / sign: int32, n*nshifts each."""
    st1, st2 = _as(st1, np.int32), _as(st2, np.int32)
    n = st1.shape[0]
    cap = n * (1 + int(nshifts))
    code = None if code is None else _as(code, np.int32)
    if shift is not None and np.asarray(shift).dtype.itemsize > 4 and len(shift) and \
            (int(np.max(shift)) > 2**31 - 1 or int(np.min(shift)) < -2**31):
        # (CoordCreator._draw_raw draws int64 shifts when |minshift| / |maxshift| >= 2^31: they must not be truncated silently)
        raise ValueError("control shifts beyond +-2^31 bp do not fit the int32 shifts of pup_host_windows")
    shift = None if shift is None else _as(shift, np.int32)
    sign = None if sign is None else _as(sign, np.int32)
    alloc = _ARENA.take if arena else (lambda m: np.empty(m, np.int32))
    r0, c0 = alloc(cap), alloc(cap)                                 # per-region intermediates: group_tiles makes the DMA source
    code_out = alloc(cap) if code is not None else None
    n_roi = C.c_int64(0)
    kept = _ffi.lib().pup_host_windows(_ptr(st1), _ptr(st2), _ptr(code), n, _ptr(shift), _ptr(sign), int(nshifts),
                                       float(resolution), int(off1), int(off2), int(lo1), int(hi1), int(lo2), int(hi2),
                                       int(h), int(w), _ptr(r0), _ptr(c0), _ptr(code_out), C.byref(n_roi))
    if kept < 0:
        raise ValueError("pup_host_windows: bad arguments")
    return r0[:kept], c0[:kept], (None if code_out is None else code_out[:kept]), int(n_roi.value)


def host_windows_into(r0, c0, at, st1, st2, shift, sign, nshifts, resolution, off1, off2, lo1, hi1, lo2, hi2, h, w, controls_only=False):
    """pup_host_windows / pup_host_control_windows writing at position `at` of the caller's (page-locked) r0 / c0 arrays: the ROI
    windows of a region (shift / sign None, nshifts 0) or its shifted control copies alone.  Returns the number of windows written."""
    st1, st2 = _as(st1, np.int32), _as(st2, np.int32)
    n = st1.shape[0]
    cap = n * int(nshifts) if controls_only else n * (1 + int(nshifts))
    if at + cap > r0.shape[0] or r0.dtype != np.int32 or c0.dtype != np.int32:
        raise ValueError("host_windows_into: output arrays too small")
    if shift is not None and np.asarray(shift).dtype.itemsize > 4 and len(shift) and \
            (int(np.max(shift)) > 2**31 - 1 or int(np.min(shift)) < -2**31):
        raise ValueError("control shifts beyond +-2^31 bp do not fit the int32 shifts of pup_host_windows")
    shift = None if shift is None else _as(shift, np.int32)
    sign = None if sign is None else _as(sign, np.int32)
    pr, pc = C.c_void_p(r0.ctypes.data + 4 * int(at)), C.c_void_p(c0.ctypes.data + 4 * int(at))
    args = (_ptr(st1), _ptr(st2), None, n, _ptr(shift), _ptr(sign), int(nshifts), float(resolution), int(off1), int(off2),
            int(lo1), int(hi1), int(lo2), int(hi2), int(h), int(w), pr, pc, None)
    kept = _ffi.lib().pup_host_control_windows(*args) if controls_only else _ffi.lib().pup_host_windows(*args, None)
    if kept < 0:
        raise ValueError("pup_host_windows: bad arguments" if kept == -1 else "pup_host_windows: out of memory")
    return int(kept)


def legacy_randint(low, high, m, scale=1, offset=0, discard=False, dtype=np.int64, out=None):
    """offset + scale * np.random.randint(low, high, m) of numpy's LEGACY global generator, drawn by the library
    (pup_host_mt_randint: same numbers, same generator state afterwards, several threads instead of one call per number).
    discard=True only advances the generator.  Falls back to numpy itself for tiny requests, ranges wider than 2^32 or a
    global generator that is not MT19937."""
    m = int(m)
    st = np.random.get_state(legacy=True) if m >= 2048 and 0 < int(high) - int(low) <= (1 << 32) else None
    if out is not None and (out.shape != (m,) or not out.flags.c_contiguous or out.dtype.itemsize not in (4, 8)):
        raise ValueError("legacy_randint: out must be a contiguous int32 / int64 array of m entries")
    if st is None or st[0] != "MT19937":
        d = np.random.randint(low, high, m)
        if discard:
            return None
        d = (d if (scale == 1 and offset == 0) else offset + scale * d)
        if out is not None:
            out[:] = d
            return out
        return d.astype(dtype, copy=False)
    key = np.array(st[1], dtype=np.uint32, copy=True)
    pos = C.c_int32(int(st[2]))
    if out is None:
        out = None if discard else np.empty(m, dtype)
    rc = _ffi.lib().pup_host_mt_randint(_ptr(key), C.byref(pos), int(low), int(high), m, int(scale), int(offset), _ptr(out),
                                        0 if out is None else out.dtype.itemsize)
    if rc != 0:
        raise ValueError("pup_host_mt_randint: bad arguments")
    np.random.set_state(("MT19937", key, int(pos.value), st[3], st[4]))
    return out


def lut_codes(lut, codes, add_from=None, add=0):
    """lut[codes] (+ add from entry add_from on) as int32, by the library's multi-threaded pass for long arrays (pup_host_lut_i32)."""
    lut = _as(lut, np.int32)
    n = len(codes)
    add_from = n if add_from is None else int(add_from)
    if n >= 200_000 and isinstance(codes, np.ndarray) and codes.dtype == np.int32 and codes.flags.c_contiguous and len(lut):
        out = np.empty(n, np.int32)
        if _ffi.lib().pup_host_lut_i32(_ptr(lut), len(lut), _ptr(codes), n, add_from, int(add), _ptr(out)) == 0:
            return out
    out = lut[codes]
    if add and add_from < n:
        out[add_from:] += np.int32(add)
    return out


def count_le(edges, values):
    """np.searchsorted(edges, values, side="right") for a short sorted edge list and float64 values, by the library for long arrays."""
    values = np.asarray(values)
    e = np.asarray(edges, np.float64)
    if len(values) >= 10_000 and values.dtype == np.float64 and values.flags.c_contiguous and 0 < len(e) <= 64 \
            and np.array_equal(e, np.asarray(edges)) and not np.isnan(e).any():
        out = np.empty(len(values), np.int32)
        if _ffi.lib().pup_host_count_le(_ptr(e), len(e), _ptr(values), len(values), _ptr(out)) == 0:
            return out
    return np.searchsorted(edges, values, side="right")


def legacy_randint_plan(calls):
    """A sequence of legacy_randint calls as ONE library job (pup_host_mt_randint_plan): `calls` = [(low, high, m, scale, offset,
    out)], out a contiguous int32 / int64 array of m entries or None (draw and discard).  Same numbers, same generator state
    afterwards as the calls made one by one.  Returns False — nothing drawn, generator untouched — when the library declines
    (calls rejecting over different ranges, a range wider than 2^32, a global generator that is not MT19937): the caller then
    makes the calls itself."""
    if not calls:
        return True
    st = np.random.get_state(legacy=True)
    if st[0] != "MT19937":
        return False
    n = len(calls)
    low = np.empty(n, np.int64); high = np.empty(n, np.int64); m = np.empty(n, np.int64)
    scale = np.empty(n, np.int64); offset = np.empty(n, np.int64); nbytes = np.zeros(n, np.int32)
    outs = (C.c_void_p * n)()
    for k, (lo, hi, mk, sc, of, out) in enumerate(calls):
        if not 0 < int(hi) - int(lo) <= (1 << 32):
            return False
        if out is not None:
            if out.shape != (int(mk),) or not out.flags.c_contiguous or out.dtype.itemsize not in (4, 8) or out.dtype.kind != "i":
                raise ValueError("legacy_randint_plan: out must be a contiguous int32 / int64 array of m entries")
            outs[k] = out.ctypes.data
            nbytes[k] = out.dtype.itemsize
        low[k], high[k], m[k], scale[k], offset[k] = int(lo), int(hi), int(mk), int(sc), int(of)
    key = np.array(st[1], dtype=np.uint32, copy=True)
    pos = C.c_int32(int(st[2]))
    rc = _ffi.lib().pup_host_mt_randint_plan(_ptr(key), C.byref(pos), n, _ptr(low), _ptr(high), _ptr(m), _ptr(scale), _ptr(offset),
                                             C.cast(outs, C.c_void_p), _ptr(nbytes))
    if rc != 0:
        return False
    np.random.set_state(("MT19937", key, int(pos.value), st[3], st[4]))
    return True


def factorize_objects(a):
    """pd.factorize(a) (codes int64 with -1 for missing values, uniques in order of first appearance) for a 1-D object array whose
    entries point at few distinct objects: the pointers are factorised by identity in the library (pup_host_factorize_ptr), pandas
    only sees one representative per distinct object.  Arrays with more than 65 536 distinct objects, short or non-object arrays go
    to pandas as they are."""
    import pandas as pd
    a = np.asarray(a)
    n = a.shape[0]
    if a.dtype != object or a.ndim != 1 or n < 50_000 or not a.flags.c_contiguous:
        return pd.factorize(a)
    codes32 = np.empty(n, np.int32)
    first = np.empty(65_536, np.int64)
    nu = _ffi.lib().pup_host_factorize_ptr(a.ctypes.data, n, _ptr(codes32), _ptr(first), first.shape[0])
    if nu < 0:
        return pd.factorize(a)
    reps = a[first[:nu]]
    rcodes, uniq = pd.factorize(reps)                           # distinct objects that are equal strings share a code; missing: -1
    if len(rcodes) == len(uniq) and np.array_equal(rcodes, np.arange(len(uniq))):
        return codes32.astype(np.int64), uniq                   # (every distinct object its own value: the numbering stands as it is)
    return rcodes.astype(np.int64)[codes32], uniq


def stable_argsort(keys, bits):
    """np.argsort(keys, kind="stable") for non-negative integer keys below 2**bits, by the library's multi-threaded radix sort."""
    keys = np.ascontiguousarray(keys).view(np.uint64) if np.asarray(keys).dtype.itemsize == 8 else np.ascontiguousarray(keys, np.uint64)
    order = np.empty(keys.shape[0], np.int64)
    rc = _ffi.lib().pup_host_argsort(_ptr(keys), keys.shape[0], int(bits), _ptr(order))
    if rc == -2:                             # PUP_ENOMEM: no scratch memory / no thread to be had — numpy does the same sort
        return np.argsort(keys, kind="stable")
    if rc != 0:
        raise ValueError("pup_host_argsort: bad arguments")
    return order


def pair_region_counts(s1, e1, s2, e2, c1, c2, mindist, maxdist, reg_code, reg_start, reg_end):
    """pup_host_pair_region_counts: kept BEDPE rows per view region, from the unsorted table (None when the library declines)."""
    cols = [_as(v, np.int64) for v in (s1, e1, s2, e2)] + [_as(v, np.int32) for v in (c1, c2)]
    rc_, rs_, re_ = _as(reg_code, np.int32), _as(reg_start, np.int64), _as(reg_end, np.int64)
    out = np.zeros(rc_.shape[0], np.int64)
    rc = _ffi.lib().pup_host_pair_region_counts(*[_ptr(v) for v in cols], cols[0].shape[0], float(mindist), float(maxdist),
                                                _ptr(rc_), _ptr(rs_), _ptr(re_), rc_.shape[0], _ptr(out))
    return out if rc == 0 else None


def sort_pairs(s1, e1, s2, e2, c1, c2, rank, mindist, maxdist):
    """pup_host_sort_pairs: the distance filter and the (chrom1, chrom2, start1, start2) sort of a BEDPE table in one call.
    s1 .. e2: int64 columns, c1 / c2: int32 chromosome codes, rank[code]: the chromosome's place in sort order.  Returns None when
    the library declines (negative starts, keys beyond 63 bits, no scratch memory), else (rows, s1, e1, s2, e2, c1, c2, filtered,
    permuted): rows[i] = source row of sorted row i, the columns in sorted order."""
    n = s1.shape[0]
    rank = _as(rank, np.int64)
    cols = [_as(v, np.int64) for v in (s1, e1, s2, e2)] + [_as(v, np.int32) for v in (c1, c2)]
    rows = np.empty(n, np.int64)
    outs = [np.empty(n, np.int64) for _ in range(4)] + [np.empty(n, np.int32) for _ in range(2)]
    flags = C.c_int32(0)
    kept = _ffi.lib().pup_host_sort_pairs(*[_ptr(v) for v in cols], n, _ptr(rank), int(rank.shape[0]), float(mindist), float(maxdist),
                                          _ptr(rows), *[_ptr(v) for v in outs], C.byref(flags))
    if kept < 0:
        if kept == -1:
            raise ValueError("pup_host_sort_pairs: bad arguments")
        return None
    return (rows[:kept], *[v[:kept] for v in outs], bool(flags.value & 1), bool(flags.value & 2))


def take_rows(columns, order):
    """[col[order] for col in columns] (1-D numeric arrays of one length) through pup_host_take_rows: every column in one
    multi-threaded pass.  Columns the library cannot gather (object dtype, odd element sizes, non-contiguous) go through numpy."""
    order = np.ascontiguousarray(order, np.int64)
    n = order.shape[0]
    out = [None] * len(columns)
    src, dst, es, idx = [], [], [], []
    for j, col in enumerate(columns):
        col = np.asarray(col)
        if col.ndim == 1 and col.dtype.kind in "iufb" and col.dtype.itemsize in (1, 2, 4, 8) and col.flags.c_contiguous and n >= 4096:
            o = np.empty(n, col.dtype)
            src.append(col); dst.append(o); es.append(col.dtype.itemsize); idx.append(j)
        else:
            out[j] = col[order]
    if idx:
        n_src = min(c.shape[0] for c in src)
        ps = (C.c_void_p * len(idx))(*[c.ctypes.data for c in src])
        pd_ = (C.c_void_p * len(idx))(*[c.ctypes.data for c in dst])
        rc = _ffi.lib().pup_host_take_rows(len(idx), ps, pd_, _ptr(np.array(es, np.int32)), _ptr(order), n, n_src)
        if rc == -2:                         # PUP_ENOMEM (no thread to be had): numpy gathers the columns
            dst = [c[order] for c in src]
        elif rc != 0:
            raise IndexError("take_rows: index out of bounds")
        for j, o in zip(idx, dst):
            out[j] = o
    return out


class RunTile:
    """Tile numbers of a region's windows as two runs: the first `split` windows belong to tile `a`, the other `n - split` to
    tile `b` (an ungrouped region with controls: ROI windows, then the shifted copies).  group_tiles hands the runs to the library
    as they are; np.asarray() gives the per-window array for the callers that want one (oracle replays, the flipped-window paths)."""
    __slots__ = ("split", "n", "a", "b")

    def __init__(self, split, n, a, b):
        self.split, self.n, self.a, self.b = int(split), int(n), int(a), int(b)

    def __len__(self):
        return self.n

    def __array__(self, dtype=None, copy=None):
        out = np.empty(self.n, np.int32)
        out[:self.split] = self.a
        out[self.split:] = self.b
        return out if dtype is None else out.astype(dtype, copy=False)


def group_tiles(parts, T):
    """pup_host_group_tiles(_runs): [(r0, c0, tile), ...] (int32 arrays per region; tile may be a RunTile) -> (r0, c0, tile_ptr)
    of one engine call, stably grouped by tile; r0 / c0 page-locked."""
    keep = [(_as(a, np.int32), _as(b, np.int32), t if isinstance(t, RunTile) else _as(t, np.int32)) for a, b, t in parts]
    P = len(keep)
    n = sum(a.shape[0] for a, _, _ in keep)
    arr = lambda k: (C.c_void_p * max(P, 1))(*[x[k].ctypes.data for x in keep])       # noqa: E731
    tiles = (C.c_void_p * max(P, 1))(*[None if isinstance(x[2], RunTile) else x[2].ctypes.data for x in keep])
    lens = np.array([a.shape[0] for a, _, _ in keep], np.int64)
    for a, _, t in keep:
        if len(t) != a.shape[0]:
            raise ValueError("group_tiles: tile numbers and windows differ in length")
    split = np.array([x[2].split if isinstance(x[2], RunTile) else 0 for x in keep], np.int64)
    ta = np.array([x[2].a if isinstance(x[2], RunTile) else 0 for x in keep], np.int32)
    tb = np.array([x[2].b if isinstance(x[2], RunTile) else 0 for x in keep], np.int32)
    r0, c0 = pinned_empty(n), pinned_empty(n)
    tile_ptr = np.empty(int(T) + 1, np.int64)
    rc = _ffi.lib().pup_host_group_tiles_runs(P, arr(0), arr(1), tiles, _ptr(split), _ptr(ta), _ptr(tb), _ptr(lens), int(T),
                                              _ptr(r0), _ptr(c0), _ptr(tile_ptr))
    if rc != 0:
        raise ValueError("pup_host_group_tiles: tile id outside [0, T)")
    return r0, c0, tile_ptr
