"""The processed feature table behind ``CoordCreator.intervals`` as a column store.

The reference keeps its features in one pandas frame and rebuilds it at every step of ``CoordCreator.process``
(coolpuppy/coolpup.py:259-385: ``astype(str)``, centres, distance filter, ``expand2D`` :94-115, ``_control_regions`` :387-453,
``_binnify`` :489-527 — a dozen column inserts, two boolean selections and a sort of a 20-column frame).  Nothing downstream
of it needs a frame: region selection, window generation and grouping read a handful of integer columns.  Here the table is a
set of numpy columns in final (filtered, sorted) row order; every other column — the caller's own, and the reference's derived
ones — is produced on first request from the untouched source frame through (keep mask, sort order).  ``frame()`` assembles
the very frame the reference's steps produce (same columns, order, dtypes, index labels) the first time somebody asks for
``CoordCreator.intervals``; a pile-up never does.

Two implementations share the interface (``n``, ``names``, ``col``, ``dtype``, ``has``, ``frame``, ``chrom_codes``):
:class:`ArrayTable` (built by :func:`build_table` straight from the source columns) and :class:`FrameTable` (a frame somebody
assigned, or one the general pandas path of ``CoordCreator._process_frame`` produced for inputs the array path does not take:
non-integer coordinates, missing chromosome names, negative starts).
"""
import numpy as np
import pandas as pd

BEDPE_DERIVED = ("center1", "center2", "distance", "exp_start1", "exp_end1", "exp_start2", "exp_end2")
BEDPE_BINS = ("stBin1", "endBin1", "stBin2", "endBin2")
BED_DERIVED = ("center", "exp_start", "exp_end")
BED_BINS = ("stBin", "endBin")


def scaled_interval(start, end, scale):
    """bioframe.expand(df, scale=...): grow an interval by 0.5*(scale-1)*length on both sides, rounded (np.round,
    half to even) back to the integer dtype of the input — what the reference calls for rescaled pile-ups (:78-91)."""
    start = np.asarray(start)
    end = np.asarray(end)
    pads = 0.5 * (scale - 1) * (end - start)
    return np.round(start - pads).astype(start.dtype), np.round(end + pads).astype(end.dtype)


class FrameTable:
    """A processed frame (the general pandas path, or one assigned to ``CoordCreator.intervals``) behind the table interface."""
    lazy = False

    def __init__(self, frame, kind):
        self._frame = frame
        self.kind = kind
        self.n = len(frame)
        self.names = list(frame.columns)
        self._have = set(self.names)
        self._cols = {}
        self._codes = None
        self.sorted_pairs = False

    def has(self, name):
        return name in self._have

    def dtype(self, name):
        return self._frame[name].dtype

    def col(self, name):
        v = self._cols.get(name)
        if v is None:
            v = self._cols[name] = self._frame[name].values
        return v

    def bins32(self, name):
        return None

    def frame(self):
        return self._frame

    def chrom_codes(self):
        """(codes of the chromosome column(s) over the rows, names): one dictionary for both sides of a pair."""
        if self._codes is None:
            if self.kind == "bedpe":
                n = self.n
                codes, uniq = pd.factorize(np.concatenate([self.col("chrom1"), self.col("chrom2")]))
                self._codes = ((codes[:n], codes[n:]), [str(u) for u in uniq])
            else:
                codes, uniq = pd.factorize(self.col("chrom"))
                self._codes = ((codes,), [str(u) for u in uniq])
        return self._codes

    def group_codes(self, a, b=None):
        """pd.factorize of column a (or of a and b over one dictionary): ((codes_a[, codes_b]), uniques)."""
        if b is None:
            codes, uniq = pd.factorize(self.col(a))
            return (codes.astype(np.int64),), uniq
        va, vb = self.col(a), self.col(b)
        codes, uniq = pd.factorize(np.concatenate([va, vb]))
        return (codes[:len(va)].astype(np.int64), codes[len(va):].astype(np.int64)), uniq


def _string_codes(series):
    """(codes, names) of a chromosome column as the reference's ``astype(str)`` sees it: distinct values that print alike share
    a code.  None when a value is missing (the general path words those as pandas does)."""
    from .engine import factorize_objects
    a = series.to_numpy()
    if a.dtype != object:
        a = np.asarray(series, dtype=object)
    codes, uniq = factorize_objects(np.ascontiguousarray(a))
    if len(codes) and int(codes.min()) < 0:
        return None
    names, where, remap = [], {}, np.empty(len(uniq), np.int64)
    for j, u in enumerate(uniq):
        s = str(u)
        if s not in where:
            where[s] = len(names)
            names.append(s)
        remap[j] = where[s]
    if len(names) != len(uniq):
        codes = remap[codes]
    return codes, names


def _merge_dictionaries(codes_b, names_a, names_b):
    """Codes of a second column over the first column's dictionary, extended by the names only the second one holds."""
    where = {s: i for i, s in enumerate(names_a)}
    names = list(names_a)
    remap = np.empty(len(names_b), np.int64)
    for j, s in enumerate(names_b):
        if s not in where:
            where[s] = len(names)
            names.append(s)
        remap[j] = where[s]
    if np.array_equal(remap, np.arange(len(names_b))):
        return codes_b, names
    return remap[codes_b], names


def _take(arrays, order):
    from .engine import take_rows
    return take_rows([np.ascontiguousarray(a) for a in arrays], order)


def _stable_order(key, bits):
    """Stable argsort of non-negative integer keys, or None when they are in order already."""
    n = len(key)
    if n < 2 or bool(np.all(key[1:] >= key[:-1])):
        return None
    if n >= 50_000:
        from .engine import stable_argsort
        return stable_argsort(key, bits)
    return np.argsort(key, kind="stable")


class ArrayTable:
    """Columns of the processed table, produced on demand from the source frame (see the module docstring)."""
    lazy = True

    def __init__(self, src, kind, rows, filtered, n, ready, codes, chrom_names, resolution, flank, rescale_flank, tag_kind):
        # rows[i] = source row of table row i (None: the source rows as they are); filtered: the distance filter dropped rows
        self.src, self.kind, self.rows, self.filtered, self.n = src, kind, rows, bool(filtered), int(n)
        self._cols = dict(ready)                 # sorted columns worked out so far: starts / ends (int64)
        self._codes = codes                      # chromosome codes per side, sorted rows
        self.chrom_names = list(chrom_names)
        self.resolution, self.flank, self.rescale_flank = int(resolution), flank, rescale_flank
        self._src_names = list(src.columns)
        derived = (BEDPE_DERIVED if kind == "bedpe" else BED_DERIVED)
        bins = (BEDPE_BINS if kind == "bedpe" else BED_BINS)
        tail = list(derived) + (["kind"] if tag_kind else []) + list(bins)
        self._derived = set(tail)
        self.names = self._src_names + [c for c in tail if c not in set(self._src_names)]
        self._have = set(self.names)
        self._frame = None
        self._bins32 = {}
        self._gc = {}
        self.sorted_pairs = kind == "bedpe"      # rows of one chromosome pair are contiguous

    # -- interface ---------------------------------------------------------------------------------------------------
    def has(self, name):
        return name in self._have

    def dtype(self, name):
        if self._frame is not None:
            return self._frame[name].dtype
        if name in self._derived or name in ("chrom", "chrom1", "chrom2"):
            return self.col(name).dtype
        return self.src[name].dtype

    def chrom_codes(self):
        return self._codes, self.chrom_names

    def _sides(self):
        return ("1", "2") if self.kind == "bedpe" else ("",)

    def _gather(self, v):
        if self.rows is not None:
            v = _take([v], self.rows)[0] if isinstance(v, np.ndarray) else v.take(self.rows)
        return v

    def _cbin(self, s):
        """Bin of the feature's centre, floor(((start + end) / 2) / resolution): the sum is an integer and the resolution too, so
        the integer floor division IS the reference's float expression (its quotient is never within rounding distance of an
        integer unless it is one: multiples of 1 / (2 * resolution) against a spacing of 2^-52 relative)."""
        return (self.col("start" + s) + self.col("end" + s)) // (2 * self.resolution)

    def bins32(self, name):
        """stBin* / endBin* as int32 (what the window passes consume), worked out without an int64 detour; None when a bin does
        not fit or the frame has been materialised."""
        v = self._bins32.get(name)
        if v is None and self._frame is None and name in self._derived and name[:5] in ("stBin", "endBi"):
            s = name[-1] if name[-1] in "12" else ""
            if self.rescale_flank is not None:
                full = self.col(name)
                if len(full) and not (-2**31 < int(full.min()) and int(full.max()) < 2**31 - 2**24):
                    return None
                v = self._bins32[name] = full.astype(np.int32)
                return v
            cb = self._cbin(s)
            if len(cb) and not (-2**30 < int(cb.min()) and int(cb.max()) < 2**30):
                return None
            cb = cb.astype(np.int32)
            pad = np.int32(int(self.flank) // self.resolution)
            self._bins32["stBin" + s], self._bins32["endBin" + s] = cb - pad, cb + (pad + np.int32(1))
            v = self._bins32[name]
        return v

    def col(self, name):
        v = self._cols.get(name)
        if v is not None:
            return v
        if self._frame is not None and name in self._have:
            v = self._frame[name].values
        elif name in self._derived:
            v = self._derive(name)
        elif name in ("chrom", "chrom1", "chrom2") and name in self._have:
            side = {"chrom": 0, "chrom1": 0, "chrom2": 1}[name]
            v = np.asarray(self.chrom_names, dtype=object)[self._codes[side]]
        else:
            s = self.src[name]
            v = s.to_numpy() if isinstance(s.dtype, np.dtype) else s.array
            v = self._gather(v)
        self._cols[name] = v
        return v

    def _derive(self, name):
        res = self.resolution
        if name == "kind":
            v = np.empty(self.n, dtype=object)
            v[:] = "ROI"
            return v
        if name == "distance":
            return self.col("center2") - self.col("center1")
        s = name[-1] if name[-1] in "12" else ""
        stem = name[:len(name) - len(s)]
        if stem == "center":
            return (self.col("start" + s) + self.col("end" + s)) / 2
        if stem in ("stBin", "endBin"):
            if self.rescale_flank is not None:
                es, ee = scaled_interval(self.col("start" + s), self.col("end" + s), 2 * self.rescale_flank + 1)
                st = np.floor(es / res).astype(int)
                en = np.ceil(ee / res).astype(int)
            else:
                pad = int(self.flank) // res
                cb = self._cbin(s).astype(np.int64)
                st, en = cb - pad, cb + (pad + 1)
            self._cols["stBin" + s], self._cols["endBin" + s] = st, en
            return st if stem == "stBin" else en
        if stem == "exp_start":
            return self.col("stBin" + s) * res
        if stem == "exp_end":
            return self.col("endBin" + s) * res
        raise KeyError(name)

    def group_codes(self, a, b=None):
        """Integer codes of a grouping column (a pair of columns over one dictionary) in sorted row order: string-like columns are
        factorised on the SOURCE column by object identity (few distinct objects) and the codes are gathered — a million object
        pointers never move.  The dictionary's order is that of the source rows; callers only look values up in it."""
        from .engine import factorize_objects
        key = (a, b)
        if key in self._gc:
            return self._gc[key]
        out = None
        wanted = [c for c in (a, b) if c is not None]
        plain = all(c in self._src_names and c not in self._derived and c not in ("chrom", "chrom1", "chrom2") for c in wanted)
        if plain and self._frame is None and all(self.src[c].dtype == object for c in wanted):
            ca, ua = factorize_objects(np.ascontiguousarray(self.src[a].to_numpy()))
            if b is None:
                out = (self._gather(ca.astype(np.int64)),), ua
            else:
                cb, ub = factorize_objects(np.ascontiguousarray(self.src[b].to_numpy()))
                if not (pd.isna(ua).any() or pd.isna(ub).any()):
                    where = {u: i for i, u in enumerate(ua)}
                    uniq = list(ua)
                    remap = np.empty(len(ub), np.int64)
                    for j, u in enumerate(ub):
                        if u not in where:
                            where[u] = len(uniq)
                            uniq.append(u)
                        remap[j] = where[u]
                    cb2 = np.where(cb >= 0, remap[np.maximum(cb, 0)], -1) if len(ub) else cb
                    out = (self._gather(ca.astype(np.int64)), self._gather(cb2.astype(np.int64))), np.asarray(uniq, dtype=object)
        if out is None:
            out = FrameTable.group_codes(self, a, b)
        self._gc[key] = out
        return out

    def frame(self):
        """The frame the reference's process() ends with: the caller's columns (chromosome names as str) filtered and sorted, the
        derived columns behind them — built once, in one construction."""
        if self._frame is None:
            # index labels.  BED: the caller's, permuted.  BEDPE: the kept rows renumbered from 0 BEFORE the sort — the reference
            # resets the index after its distance filter whether or not it dropped a row (coolpup.py:321), so a caller's own
            # labels (a subset of a bigger frame, say) never survive
            if self.kind == "bedpe" and not self.filtered:
                index = pd.RangeIndex(self.n) if self.rows is None else pd.Index(np.asarray(self.rows, np.int64))
            elif self.rows is None:
                index = self.src.index
            elif not self.filtered:
                index = self.src.index.take(self.rows)
            else:
                kept = np.zeros(len(self.src), np.int64)
                kept[self.rows] = 1
                index = pd.Index((np.cumsum(kept) - 1)[self.rows])
            cols = {}
            for name in self.names:
                v = self.col(name)
                if name in BEDPE_BINS or name in BED_BINS or name.startswith(("exp_start", "exp_end")):
                    v = np.asarray(v).astype(np.int64, copy=False)
                cols[name] = v
            self._frame = pd.DataFrame(cols, index=index, copy=False)
        return self._frame


def _same_objects(a, b):
    """True when two object columns hold the same object in every row (a cis BEDPE table's chrom1 / chrom2 as pandas' readers
    box them): one memcmp of the pointer arrays."""
    if a.dtype != object or b.dtype != object or a.shape != b.shape or not (a.flags.c_contiguous and b.flags.c_contiguous):
        return False
    import ctypes
    memcmp = ctypes.CDLL(None).memcmp
    memcmp.argtypes, memcmp.restype = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t], ctypes.c_int
    return a.nbytes == 0 or memcmp(a.ctypes.data, b.ctypes.data, a.nbytes) == 0


def build_table(src, kind, resolution, flank, rescale_flank, mindist, maxdist, tag_kind, early=None):
    """The array path of CoordCreator.process for a feature frame `src` (reference :259-385, 489-527).  Returns an ArrayTable, or
    None when the input is not of the plain kind it handles (integer coordinates >= 0, integer resolution, chromosome names
    without missing values, keys that fit 63 bits) or is empty after the distance filter — the caller then takes the pandas path,
    which words warnings and odd dtypes the way the reference does."""
    if not float(resolution).is_integer() or len(src) == 0:
        return None
    res = int(resolution)
    sides = ("1", "2") if kind == "bedpe" else ("",)
    S, E = [], []
    for s in sides:
        for nm, dst in (("start" + s, S), ("end" + s, E)):
            c = src[nm]
            if not isinstance(c.dtype, np.dtype) or c.dtype.kind not in "iu":
                return None
            v = c.to_numpy()
            if v.dtype != np.int64:
                if v.dtype == np.uint64 and len(v) and int(v.max()) >= 2**62:
                    return None
                v = v.astype(np.int64)
            dst.append(np.ascontiguousarray(v))
    if any(len(v) and (int(v.max()) >= 2**52) for v in E):
        return None
    first = _string_codes(src["chrom" + sides[0]])
    if first is None:
        return None
    codes, names = [first[0]], first[1]
    if kind == "bedpe":
        if _same_objects(src["chrom1"].to_numpy(), src["chrom2"].to_numpy()):
            codes.append(codes[0])
        else:
            second = _string_codes(src["chrom2"])
            if second is None:
                return None
            c2, names = _merge_dictionaries(second[0], names, second[1])
            codes.append(c2)
    nu = len(names)
    rank = np.empty(nu, np.int64)
    rank[np.argsort(np.asarray(names, dtype=object), kind="stable")] = np.arange(nu)
    if kind == "bedpe":
        # the reference's filter and sort — (chrom1, chrom2, start1, start2), stable, chromosome names in string order — in one
        # call of the library
        from .engine import sort_pairs
        if early is not None:          # (what only needs the unsorted columns — the control draws' sizes — starts before the sort)
            early(S, E, codes, names)
        got = sort_pairs(S[0], E[0], S[1], E[1], codes[0], codes[1], rank, mindist, maxdist)
        if got is None or len(got[0]) == 0:
            return None
        rows, s1, e1, s2, e2, c1, c2, filtered, permuted = got
        if s1.max() >= 2**52 or s2.max() >= 2**52:
            return None
        ready = {"start1": s1, "end1": e1, "start2": s2, "end2": e2}
        return ArrayTable(src, kind, rows if (filtered or permuted) else None, filtered, len(rows), ready, (c1, c2), names, res, flank,
                          rescale_flank, tag_kind)
    # bed: (chrom, start), stable
    if len(S[0]) and (int(S[0].min()) < 0 or int(S[0].max()) >= 2**52):
        return None
    starts = S[0] // max(int(np.gcd.reduce(S[0])), 1)              # bin-aligned anchors: fewer key bits
    width = [int(v).bit_length() for v in (nu - 1, starts.max())]
    if sum(width) > 63:
        return None
    order = _stable_order(rank[codes[0]] << width[1] | starts, sum(width))
    code = codes[0].astype(np.int32)
    if order is not None:
        S[0], E[0], code = _take([S[0], E[0], code], order)
    return ArrayTable(src, kind, order, False, len(code), {"start": S[0], "end": E[0]}, (code,), names, res, flank, rescale_flank, tag_kind)
