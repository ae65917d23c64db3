"""In-memory stand-in for the slice of ``cooler.Cooler`` that coolpuppy's pile-up path touches.

``cooler`` itself is a third-party dependency of the reference (requirements.txt:3, source not in
the reference tree).  The host layer here only needs the following surface, restated from the call
sites in coolpuppy/coolpup.py (line numbers of the reference):

* ``clr.binsize`` (:838), ``clr.chromsizes`` (:860), ``clr.chromnames`` (:928), ``clr.filename`` (:1652, :2277)
* ``clr.offset(chrom)`` / ``clr.extent(region)`` (:923-925)
* ``clr.bins().columns`` (:950, :957), ``clr.bins()[col].fetch(region)`` (:1083-1098)
* the pixel table behind ``clr.matrix(sparse=True, balance=w).fetch(r1, r2)`` (:1053-1055) — here
  exposed RAW as :meth:`ArrayCooler.pixel_table` (``indexes/bin1_offset``, ``pixels/bin2_id``,
  ``pixels/count``), because the GPU engine consumes the upper-triangular table directly.

A real ``cooler.Cooler`` can be adapted with :func:`from_cooler` when the package is installed.
"""
import os

import numpy as np
import pandas as pd


def _parse_region(region, chromsizes):
    """Accept 'chr1', ('chr1', s, e), a 3-field Series (coolpup.py:1046-1051) -> (chrom, start, end)."""
    if isinstance(region, str):
        return region, 0, int(chromsizes[region])
    if isinstance(region, pd.Series):
        region = tuple(region.iloc[:3])
    chrom, start, end = region[0], region[1], region[2]
    start = 0 if start is None else int(start)
    end = int(chromsizes[chrom]) if end is None else int(end)
    return str(chrom), start, end


class _ColumnSelector:
    def __init__(self, clr, col):
        self._clr, self._col = clr, col

    def fetch(self, region):
        lo, hi = self._clr.extent(region)
        return pd.Series(self._clr._bins[self._col][lo:hi], index=np.arange(lo, hi), name=self._col)

    def __getitem__(self, key):
        return pd.Series(self._clr._bins[self._col][key], name=self._col)


class _BinsSelector:
    def __init__(self, clr):
        self._clr = clr

    @property
    def columns(self):
        return pd.Index(list(self._clr._bins.keys()))

    def __getitem__(self, key):
        if isinstance(key, str):
            return _ColumnSelector(self._clr, key)
        if isinstance(key, (list, tuple)):
            return pd.DataFrame({k: self._clr._bins[k] for k in key})
        return pd.DataFrame({k: v[key] for k, v in self._clr._bins.items()})

    def fetch(self, region):
        lo, hi = self._clr.extent(region)
        return pd.DataFrame({k: v[lo:hi] for k, v in self._clr._bins.items()}, index=np.arange(lo, hi))


class ArrayCooler:
    """A cooler held as numpy arrays (single resolution, upper-triangular, sorted pixels)."""

    def __init__(self, chromsizes, binsize, bin1_offset, bin2_id, count, bins=None, filename="in_memory.cool"):
        self.binsize = int(binsize)
        if not isinstance(chromsizes, pd.Series):
            chromsizes = pd.Series(dict(chromsizes), dtype=np.int64)
        self.chromsizes = chromsizes.astype(np.int64)
        self.chromsizes.name = "length"
        self.chromnames = [str(c) for c in self.chromsizes.index]
        nb = -(-self.chromsizes.values // self.binsize)          # ceil
        self.chrom_offset = np.concatenate([[0], np.cumsum(nb)]).astype(np.int64)
        self._chrom_index = {c: i for i, c in enumerate(self.chromnames)}
        self.nbins = int(self.chrom_offset[-1])
        self.bin1_offset = np.ascontiguousarray(bin1_offset, np.int64)
        self.bin2_id = np.ascontiguousarray(bin2_id)
        self.count = np.ascontiguousarray(count)
        if self.bin1_offset.shape[0] != self.nbins + 1:
            raise ValueError("bin1_offset must have nbins+1 entries")
        self.filename = filename
        starts = np.concatenate([np.arange(n, dtype=np.int64) * self.binsize for n in nb])
        lens = np.repeat(self.chromsizes.values, nb)
        self._bins = {
            "chrom": np.repeat(np.array(self.chromnames, dtype=object), nb),
            "start": starts,
            "end": np.minimum(starts + self.binsize, lens),
        }
        for k, v in (bins or {}).items():
            v = np.asarray(v)
            if v.shape[0] != self.nbins:
                raise ValueError(f"bins column {k!r} has {v.shape[0]} entries, expected {self.nbins}")
            self._bins[k] = v

    # -- cooler-like surface ---------------------------------------------------------------------------
    def offset(self, region):
        """First global bin id of a chromosome, or of the bin holding region start."""
        if isinstance(region, str) and region in self._chrom_index:
            return int(self.chrom_offset[self._chrom_index[region]])
        chrom, start, _ = _parse_region(region, self.chromsizes)
        return int(self.chrom_offset[self._chrom_index[chrom]]) + start // self.binsize

    def extent(self, region):
        """(lo, hi) global bin range covering region: lo = offset + start//binsize, hi = offset + ceil(end/binsize)."""
        chrom, start, end = _parse_region(region, self.chromsizes)
        off = int(self.chrom_offset[self._chrom_index[chrom]])
        return off + start // self.binsize, off + -(-end // self.binsize)

    def bins(self):
        return _BinsSelector(self)

    def set_bins_column(self, name, values):
        values = np.asarray(values)
        if values.shape[0] != self.nbins:
            raise ValueError("wrong length")
        self._bins[name] = values

    # -- what the GPU engine consumes -------------------------------------------------------------------
    def pixel_table(self):
        """(indexes/bin1_offset int64[nbins+1], pixels/bin2_id, pixels/count) — the CSR of the upper triangle."""
        return self.bin1_offset, self.bin2_id, self.count

    @property
    def nnz(self):
        return int(self.bin2_id.shape[0])


def from_cooler(clr):
    """Adapt a real ``cooler.Cooler`` (when that package is available) into an :class:`ArrayCooler`.

    Reads the whole pixel table into host memory once; ``indexes/bin1_offset`` already is the CSR
    row pointer, so no transformation is needed.
    """
    with clr.open("r") as h5:
        bin1_offset = h5["indexes/bin1_offset"][:]
        bin2_id = h5["pixels/bin2_id"][:]
        count = h5["pixels/count"][:]
    table = clr.bins()[:]
    cols = {c: table[c].values for c in table.columns if c not in ("chrom", "start", "end")}
    return ArrayCooler(clr.chromsizes, clr.binsize, bin1_offset, bin2_id, count, bins=cols,
                       filename=clr.filename)


_ADAPTED = {}          # id(cooler object) or path -> (weak reference to the object | mtime, ArrayCooler)
_MAX_PATH_ADAPTERS = 4  # tables read from paths stay alive through this cache only: keep the most recent few (a table is GBs)


def as_array_cooler(clr):
    """Return clr if it already exposes ``pixel_table()``; open it when it is a path / cooler URI
    ("file.cool" or "file.mcool::resolutions/10000"); otherwise adapt a ``cooler.Cooler``.  The adapter of a given
    cooler object (or unchanged file) is made once and kept, so that repeated pile-ups on it find the pixel table
    already resident on the GPU (the engine cache is keyed on the adapter)."""
    import weakref
    if hasattr(clr, "pixel_table"):
        return clr
    if isinstance(clr, (str, os.PathLike)):
        from .cool_io import read_cool
        path, _, group = str(clr).partition("::")
        stamp = (os.path.getmtime(path), os.path.getsize(path))
        hit = _ADAPTED.pop(str(clr), None)
        if hit is None or hit[0] != stamp:
            hit = (stamp, read_cool(path, group=group or "/"))
        _ADAPTED[str(clr)] = hit                            # (re-inserted: dicts keep insertion order = recency)
        paths = [k for k in _ADAPTED if isinstance(k, str)]
        for k in paths[:max(0, len(paths) - _MAX_PATH_ADAPTERS)]:
            del _ADAPTED[k]
        return hit[1]
    hit = _ADAPTED.get(id(clr))
    if hit is not None and hit[0]() is clr:
        return hit[1]
    ac = from_cooler(clr)
    try:
        ref = weakref.ref(clr, lambda _r, k=id(clr): _ADAPTED.pop(k, None))
    except TypeError:                   # not weak-referenceable: adapt every time
        return ac
    _ADAPTED[id(clr)] = (ref, ac)
    return ac
