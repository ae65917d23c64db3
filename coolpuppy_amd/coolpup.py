"""Host-side mirror of coolpuppy's pile-up API on top of the MI355X engine.

Same public names, arguments, defaults, error behaviour and returned DataFrame layout as the
reference's ``coolpuppy/coolpup.py`` (``CoordCreator`` :150-749, ``PileUpper`` :752-1919,
``pileup`` :1922-2279) — so code written against coolpuppy keeps working — but the work is organised
for a GPU instead of a per-snippet Python loop:

* coordinates are generated as numpy column tables per region (not one dict per snippet); the
  random-shift controls issue the SAME ``np.random`` legacy calls in the SAME order as the
  reference (``_control_regions`` :387-453), so control windows are bit-identical for ``nproc=1``;
* every snippet becomes (r0, c0, tile, flip): top-left GLOBAL bins + accumulator index; the whole
  per-snippet extract/mask/normalise/accumulate loop (``_stream_snips`` :1059-1191,
  ``accumulate_stream`` :1236-1283, ``_add_snip`` lib/puputils.py:12-41) runs in ``libpup_hip.so``;
* the tail of ``pileupsWithControl`` (:1511-1654) — merge, coverage normalisation, ROI/control ratio,
  inf→NaN, symmetrisation, annotation — is restated with pandas on the fetched tiles.

By-window pile-ups, stored stripes, coverage computation and rescaled pile-ups are served by the
same engine (K3 / K4 / K5).  Per-snippet Python callbacks (``postprocess_func`` / ``extra_sum_funcs``) get their
windows from the engine (``pup_extract``, K6) and run, with the reference's per-snippet bookkeeping, on the host.
"""
import itertools
import threading
import logging
import os
import re
import warnings
from functools import partial

import numpy as np
import pandas as pd

from .cooler_lite import as_array_cooler
from .lib.puputils import SnipAccumulator, _collapse, finalize_callback_pileups, finalize_pileups, sum_pups

logger = logging.getLogger("coolpuppy")

KIND_ROI, KIND_CONTROL = 0, 1
_DEFAULT_EDGES = "default"


# ------------------------------------------------------------------------------------------------------
# small helpers
# ------------------------------------------------------------------------------------------------------
def natsorted(seq):
    """Natural sort ("chr2" < "chr10"), the ordering the reference gets from ``natsort.natsorted``."""
    def key(s):
        return [(0, int(t), "") if t.isdigit() else (1, 0, t) for t in re.split(r"(\d+)", str(s)) if t != ""]
    return sorted(seq, key=key)


def _default_band_edges():
    return np.append([0], 50000 * 2 ** np.arange(30))


def bin_distance_intervals(intervals, band_edges="default"):
    """Annotate a DataFrame that has a 'distance' column with a 'distance_band' tuple column
    (reference coolpup.py:28-51): band = (edges[i-1], edges[i]) with i = searchsorted(edges, d, 'right')."""
    if isinstance(band_edges, str) and band_edges == "default":
        band_edges = _default_band_edges()
    ids = np.searchsorted(band_edges, intervals["distance"], side="right")
    lut = {i: tuple(band_edges[i - 1:i + 1]) for i in np.unique(ids)}   # i == 0 (negative distance) -> ()
    intervals["distance_band"] = [lut[i] for i in ids]
    return intervals


def assign_groups(intervals, groupby=[]):
    """'group' column: "all", or the list of the groupby values of each row (reference :54-75)."""
    if groupby:
        keys = intervals[groupby].to_numpy()             # one row of values per snippet
        intervals["group"] = [keys[i] for i in range(len(keys))]
    else:
        intervals["group"] = "all"
    return intervals


from .intervals import scaled_interval as _scaled_interval  # noqa: E402


def expand(intervals, flank, resolution, rescale_flank=None):
    """Window of one feature: the bin holding its centre +- flank, or (rescaled pile-ups) the feature itself
    extended by rescale_flank times its length on each side (reference :78-91)."""
    out = intervals.copy()
    if rescale_flank is not None:
        out["exp_start"], out["exp_end"] = _scaled_interval(out["start"].values, out["end"].values,
                                                            2 * rescale_flank + 1)
        return out
    cbin_start = np.floor(out["center"] / resolution) * resolution
    out["exp_start"] = cbin_start - flank
    out["exp_end"] = np.floor(out["center"] / resolution + 1) * resolution + flank
    return out


def expand2D(intervals, flank, resolution, rescale_flank=None):
    """Two-sided version of :func:`expand` (reference :94-115)."""
    if rescale_flank is not None:
        for side in ("1", "2"):
            intervals["exp_start" + side], intervals["exp_end" + side] = _scaled_interval(
                intervals["start" + side].values, intervals["end" + side].values, 2 * rescale_flank + 1)
        return intervals
    for side in ("1", "2"):
        c = intervals["center" + side]
        if (float(resolution).is_integer() and intervals["start" + side].dtype.kind in "iu"
                and intervals["end" + side].dtype.kind in "iu"):
            # centres are multiples of 0.5 and the resolution is an integer: c // res == floor(c / res) exactly (the
            # quotient is never within rounding distance of an integer unless it is one), one division serves both
            q = c.values / resolution
            intervals["exp_start" + side] = np.floor(q) * resolution - flank
            intervals["exp_end" + side] = np.floor(q + 1) * resolution + flank
        else:
            intervals["exp_start" + side] = np.floor(c // resolution) * resolution - flank
            intervals["exp_end" + side] = np.floor(c / resolution + 1) * resolution + flank
    return intervals


def flip_mark_intervals_func(intervals, flipby, flip_negative_strand, extra_func=None):
    """'flip' column: negative strand of side 1, or flipby1 > flipby2 (reference :118-125)."""
    if flip_negative_strand:
        flip = intervals["strand1"].to_numpy() == "-"
    else:
        flip = intervals[flipby + "1"].to_numpy() > intervals[flipby + "2"].to_numpy()
    intervals["flip"] = flip
    return intervals if extra_func is None else extra_func(intervals)


def flip_snip_func(snip, groupby, ignore_group_order, extra_func=None):
    """Per-snippet half of the flip (reference :128-147): anti-transpose the window of a marked snippet; with
    ignore_group_order also swap every paired annotation X1 <-> X2 and rebuild its group.  Only the callback path
    calls it — the engine applies the flip while accumulating."""
    if snip["flip"]:
        snip["data"] = np.rot90(np.flipud(snip["data"]))
        if ignore_group_order:
            names = list(snip)
            stems = sorted({k[:-1] for k in names if f"{k[:-1]}1" in snip and f"{k[:-1]}2" in snip})
            for stem in stems:
                snip[f"{stem}1"], snip[f"{stem}2"] = snip[f"{stem}2"], snip[f"{stem}1"]
            if groupby:
                snip["group"] = np.array([snip[col] for col in groupby], dtype=object)
    if extra_func is not None:
        snip = extra_func(snip)
    return snip


_DRAW_AHEAD_MIN = 200_000      # control draws of a pile-up from which a helper thread draws them ahead of the window passes
_DRAW_BUFFERS = {}      # dtype -> [shift buffer, sign buffer]: the draw-ahead's outputs, kept between pile-ups (80 MB of page faults per 10^7 draws otherwise)


def _draw_signs(m, dtype=np.int64, out=None):
    """np.random.choice([-1, 1], m) — the reference's call (coolpup.py:421) — without its list conversion and fancy
    index: the legacy generator implements a uniform choice as randint(0, len(a), m) followed by a[idx], so this draws the
    same numbers and leaves the generator in the same state (tests/test_host_misc.py pins that for the installed numpy)."""
    from .engine import legacy_randint
    return legacy_randint(0, 2, m, scale=2, offset=-1, dtype=dtype, out=out)


def _draw_ints(low, high, m, discard=False, dtype=np.int64, out=None):
    """np.random.randint(low, high, m) of the legacy generator (coolpup.py:420), drawn by the library."""
    from .engine import legacy_randint
    return legacy_randint(low, high, m, discard=discard, dtype=dtype, out=out)


class _Cols(dict):
    """A column table: name -> 1-D numpy array, all of one length."""

    def __len__(self):
        for v in self.values():
            return int(v.shape[0])
        return 0

    def take(self, sel):
        return _Cols({k: v[sel] for k, v in self.items()})

    def tiled(self, k):
        return _Cols({name: np.tile(v, k) for name, v in self.items()})

    @staticmethod
    def concat(a, b):
        return _Cols({k: np.concatenate([a[k], b[k]]) for k in a})

    def frame(self):
        return pd.DataFrame({k: v for k, v in self.items()})

    @staticmethod
    def from_frame(df):
        return _Cols({c: df[c].values for c in df.columns})


def _first_codes(codes, n_keys, step=1 << 16):
    """pd.unique(codes) — the distinct values in order of first appearance — for group codes below n_keys: stretch by stretch, done
    as soon as every key has turned up (the groups of a by-distance / by-strand pile-up all occur among a region's first few
    thousand windows; hashing its 10^7 codes to learn that was 18 ms of a 1e6-pair call)."""
    n = len(codes)
    if n <= 4 * step:
        return pd.unique(codes)
    seen, order = set(), []
    for a in range(0, n, step):
        for c in pd.unique(codes[a:a + step]).tolist():
            if c not in seen:
                seen.add(c)
                order.append(c)
        if len(order) >= n_keys:
            break
        if a >= 8 * step:            # (a key that never occurs: no early end to be had — the rest in one go)
            for c in pd.unique(codes[a + step:]).tolist():
                if c not in seen:
                    seen.add(c)
                    order.append(c)
            break
    return order


def _nrows(rows):
    """Number of table rows in a selection (a slice of the sorted table or an index array)."""
    return rows.stop - rows.start if isinstance(rows, slice) else len(rows)


def _rows_where(rows, mask):
    """The rows of a selection a boolean mask over it keeps, as an index array."""
    if isinstance(rows, slice):
        return np.flatnonzero(mask) + rows.start
    return rows[mask]


# ------------------------------------------------------------------------------------------------------
# CoordCreator
# ------------------------------------------------------------------------------------------------------
class _DrawAhead:
    """The control shifts of the regions to come, drawn on a helper thread while the main thread turns the regions already
    drawn into windows.  The reference draws region by region as its stream is consumed (coolpup.py:420-436: randint then choice,
    m = rows x nshifts numbers each); the legacy generator is sequential, so the numbers themselves cannot be drawn in parallel
    — but they need not be drawn by the thread that uses them: the draws (pup_host_mt_randint, 2 ns per number, the GIL released)
    were the largest single item of a 1e6-pair pile-up's host time.  The helper owns numpy's global generator from start() to
    close(): it issues exactly the calls, in exactly the order, the main thread would have (`sizes` = the m of every region in
    stream order), at most `depth` regions ahead.  close() joins it (and, after an error in the main thread, lets it finish the
    sequence so that the generator ends where a serial run would have left it)."""

    def __init__(self, cc, sizes, depth=3):
        import queue
        import threading
        self._cc, self._sizes = cc, list(sizes)
        # every region's numbers go to their own stretch of two arrays kept between pile-ups (nobody holds a region's draws beyond
        # its window pass, and a pile-up is over before the next one draws)
        narrow = np.dtype(cc._draw_dtype()) if hasattr(cc, "_draw_dtype") else None
        self._bufs = None
        if narrow is not None:
            total = int(sum(self._sizes))
            bufs = _DRAW_BUFFERS.get(narrow)
            if bufs is None or len(bufs[0]) < total:
                bufs = _DRAW_BUFFERS[narrow] = [np.empty(total, narrow), np.empty(total, narrow)]
            self._bufs, self._at = bufs, 0
        self._q = queue.Queue(maxsize=depth)
        self._err = None
        self._stop = False
        self._th = threading.Thread(target=self._run, name="coolpuppy_amd-draws", daemon=True)
        self._th.start()

    def cancel(self, state):
        """Stop a helper nobody will consume (started early on sizes that turned out not to be the pile-up's, or before an error)
        and put numpy's legacy generator back to `state`, where it stood before the helper's first draw."""
        self._stop = True
        self.close()
        np.random.set_state(state)

    def _run_plan(self):
        """Every region's draws as one library job (engine.legacy_randint_plan: the raw stream produced a buffer ahead of a pool
        of workers that lives for the whole sequence — the call-by-call form starts its threads twice per call and twists on one
        thread while nobody else works).  False: the library declined, nothing was drawn."""
        from .engine import legacy_randint_plan
        cc, calls, parts, a = self._cc, [], [], 0
        lo, hi = int(cc.minshift), int(cc.maxshift)
        for m in self._sizes:
            sh, sg = self._bufs[0][a:a + m], self._bufs[1][a:a + m]
            a += m
            calls.append((lo, hi, m, 1, 0, sh))
            calls.append((0, 2, m, 2, -1, sg))
            if cc.trans:                 # (the second pair of a trans pile-up only moves bp columns: drawn and dropped)
                calls.append((lo, hi, m, 1, 0, None))
                calls.append((0, 2, m, 1, 0, None))
            parts.append((m, (sh, sg)))
        if not legacy_randint_plan(calls):
            return False
        self._at = a
        for part in parts:
            self._q.put(part)
        return True

    def _run(self):
        try:
            if (self._bufs is not None and len(self._sizes) > 1 and not self._stop
                    and not os.environ.get("COOLPUPPY_AMD_NO_DRAW_PLAN") and self._run_plan()):
                return
            for m in self._sizes:
                if self._stop:
                    break
                if self._bufs is not None:
                    a = self._at
                    self._at += m
                    self._q.put((m, self._cc._draw_raw_now(m, out=(self._bufs[0][a:a + m], self._bufs[1][a:a + m]))))
                else:
                    self._q.put((m, self._cc._draw_raw_now(m)))
        except BaseException as e:       # noqa: BLE001 — handed to the consumer
            self._err = e
            self._q.put((None, None))

    def take(self, m):
        got, val = self._q.get()
        if self._err is not None:
            raise self._err
        if got != m:
            raise RuntimeError(f"control draws out of step: region needs {m} numbers, {got} were drawn for it")
        return val

    def close(self):
        while self._th.is_alive():       # (an aborted run: drain, so that the helper is never left blocked on a full queue)
            try:
                self._q.get(timeout=0.05)
            except Exception:            # noqa: BLE001
                pass
        self._th.join()


class CoordCreator:
    """Turns BED / BEDPE features into pile-up windows in bin units (reference coolpup.py:150-749).

    Same constructor and attributes as the reference.  Window streams are exposed as column tables
    (:meth:`region_table`) rather than per-snippet dicts; :attr:`pos_stream` still yields dict rows for
    code that iterates it.
    """

    def __init__(self, features, resolution, *, features_format="auto", flank=100000, rescale_flank=None,
                 chroms="all", minshift=10**5, maxshift=10**6, nshifts=10, mindist="auto", maxdist=None,
                 local=False, subset=0, trans=False, seed=None, _draw_hint=None):
        # _draw_hint (pileup() only): {"regions": [(chrom, start, end), ...] in pile-up order} — the control draws of a plain
        # pile-up start as soon as the rows per region are known, while the table is still being sorted (see _start_early_draws)
        self._draw_hint = _draw_hint
        self._early_ahead = None
        # (the caller's frame is only ever READ: the processed table is a column store of its own, intervals.py, and the frame
        # behind `.intervals` is assembled from it on first access — tests/test_host_misc.py checks the caller's frame stays as it was)
        self._src = features
        self._tbl = None
        self._fc = None
        self.resolution = resolution
        self.features_format = features_format
        self.flank = flank
        self.rescale_flank = rescale_flank
        self.chroms = chroms
        self.minshift = minshift
        self.maxshift = maxshift
        self.nshifts = nshifts
        self.trans = trans
        if mindist == "auto":
            self.mindist = 2 * self.flank + 2 * self.resolution
        else:
            self.mindist = mindist
            if self.trans:
                warnings.warn("Ignoring mindist when using trans", stacklevel=2)
                self.mindist = 0
        if maxdist is None:
            self.maxdist = np.inf
        else:
            self.maxdist = maxdist
            if self.trans:
                warnings.warn("Ignoring maxdist when using trans", stacklevel=2)
                self.maxdist = np.inf
        self.local = local
        self.subset = subset
        if seed is None and (nshifts > 0 or subset > 0):
            from . import dist as _dist
            seed = _dist.shared_seed(seed)   # several ranks must draw the same subset and control shifts
        self.seed = seed
        self.process()

    # -- the processed table -------------------------------------------------------------------------------------
    @property
    def intervals(self):
        """The processed feature frame (reference attribute): filtered, sorted, with the derived columns — assembled from the
        column store on first access (intervals.ArrayTable.frame); a pile-up never asks for it."""
        return self._tbl.frame()

    @intervals.setter
    def intervals(self, frame):
        from .intervals import FrameTable
        kind = getattr(self, "kind", None) or ("bedpe" if "chrom1" in frame.columns else "bed")
        self._tbl = FrameTable(frame, kind)
        self._fc = None

    # -- construction-time processing (reference :259-385) -------------------------------------------
    def process(self):
        bedpe_cols = ["chrom1", "start1", "end1", "chrom2", "start2", "end2"]
        bed_cols = ["chrom", "start", "end"]
        src = self._src
        have = set(src.columns)
        if self.features_format is None or self.features_format == "auto":
            if have.issuperset(bedpe_cols):
                self.kind = "bedpe"
            elif have.issuperset(bed_cols):
                self.kind = "bed"
            else:
                raise ValueError(
                    "Can't determine kind of input, please specify and/or name columns correctly:"
                    "'chrom1', 'start1', 'end1', 'chrom2', 'start2', 'end2' for bedpe kind"
                    "'chrom', 'start', 'end' for bed kind"
                )
        else:
            self.kind = self.features_format
        if self.kind not in ("bed", "bedpe"):
            raise ValueError('kind can only be "bed" or "bedpe"')
        if self.rescale_flank is None and self.flank % self.resolution != 0:
            raise ValueError(
                f"flank ({self.flank}) must be a multiple of the resolution ({self.resolution}): the window "
                "would not be 2*(flank//resolution)+1 bins wide")

        if self.subset > 0:
            src = self._subset(src)
        if self.kind == "bed":
            assert have.issuperset(bed_cols), "Column names must include chrom, start, and end"
        else:
            assert have.issuperset(bedpe_cols), \
                "Column names must include chrom1, start1, end1, chrom2, start2, and end2"

        # the array path (intervals.build_table): same rows, order and columns as the pandas steps below, without the frame
        from .intervals import build_table
        try:
            self._process_table(src, build_table)
        except BaseException:
            #  the helper of _start_early_draws owns numpy's legacy generator from inside build_table on; an error
            # between there and the end of the constructor would leave it drawing with nobody to adopt or cancel it — the
            # constructor has no object to hand pileup()'s own guard
            early, self._early_ahead = self._early_ahead, None
            if early is not None:
                early[0].cancel(early[2])
            raise

    def _process_table(self, src, build_table):
        tbl = None
        if not os.environ.get("COOLPUPPY_AMD_FRAME_PATH"):
            early = self._start_early_draws if (self._draw_hint and self.kind == "bedpe" and self.nshifts > 0 and not self.trans) else None
            tbl = build_table(src, self.kind, self.resolution, self.flank, self.rescale_flank, self.mindist, self.maxdist,
                              tag_kind=self.nshifts > 0 and self.kind == "bedpe", early=early)
        if tbl is None:
            return self._process_frame(src)
        self._tbl, self._fc = tbl, None
        codes, names = tbl.chrom_codes()
        present = [{names[i] for i in np.flatnonzero(np.bincount(c, minlength=len(names)))} for c in codes]
        if self.kind == "bed":
            base = present[0]
        else:
            if self.local:
                raise ValueError("Can't make local with both sides of loops defined")
            base = (present[0] | present[1]) if self.trans else (present[0] & present[1])
        self._finish_process(base)

    def _start_early_draws(self, S, E, codes, names):
        """Called by intervals.build_table before it sorts: the rows every region of the coming pile-up will hold (the distance
        filter and the region filter on the unsorted columns, pup_host_pair_region_counts) are all the reference's control draws
        depend on (coolpup.py:420-436: rows x nshifts numbers per region, in region order) — the helper thread of _DrawAhead starts
        drawing NOW, 20 ms before the first window is asked for.  pileupsWithControl adopts it when its own sizes agree, else
        cancels it (the generator is put back where it stood)."""
        from .engine import pair_region_counts
        regs = self._draw_hint.get("regions") or []
        if len(regs) < 2 or os.environ.get("COOLPUPPY_AMD_NO_DRAW_AHEAD") or os.environ.get("COOLPUPPY_AMD_NO_EARLY_DRAWS"):
            return
        code_of = {nm: i for i, nm in enumerate(names)}
        rc = [code_of.get(str(c), len(names)) for c, _, _ in regs]
        counts = pair_region_counts(S[0], E[0], S[1], E[1], codes[0], codes[1], self.mindist, self.maxdist, rc,
                                    [int(r[1]) for r in regs], [int(r[2]) for r in regs])
        if counts is None:
            return
        sizes = [int(c) * int(self.nshifts) for c in counts if c > 0]
        if sum(sizes) >= _DRAW_AHEAD_MIN:
            state = np.random.get_state()
            self._early_ahead = (_DrawAhead(self, sizes), sizes, state)

    def _finish_process(self, base):
        self.basechroms = natsorted(list(base))
        if isinstance(self.chroms, str) and self.chroms == "all":
            self.final_chroms = natsorted(list(base))
        else:
            self.final_chroms = natsorted(list(set(self.chroms).intersection(set(self.basechroms))))
        if len(self.final_chroms) == 0:
            raise ValueError(
                """No chromosomes are in common between the coordinate
                   file and the cooler file. Are they in the same
                   format, e.g. starting with "chr"?
                   """
            )
        if logger.isEnabledFor(logging.DEBUG):
            keys = ["stBin", "endBin"] if self.kind == "bed" else ["stBin1", "endBin1", "stBin2", "endBin2"]
            dups = self.intervals.duplicated(subset=keys)
            if dups.any():
                logger.debug(f"{dups.mean() * 100:.2f}% of intervals fall within the same bin as another interval. "
                             "These are all included in the pileup.")

        if self.trans & self.local:
            raise ValueError("Cannot do local with trans=True")

        self.pos_stream = self.get_combinations if self.kind == "bed" else self.get_intervals_stream

    def _process_frame(self, src):
        """The reference's own steps on a pandas frame (:259-385) — the general path, for inputs the array path declines
        (non-integer coordinates or resolution, missing chromosome names, negative starts, nothing left after the distance
        filter).  The caller's frame is not written to: a shallow copy whose columns are only ever replaced whole."""
        iv = src.copy(deep=False)
        if self.kind == "bed":
            iv["chrom"] = iv["chrom"].astype(str)
            iv["center"] = (iv["start"] + iv["end"]) / 2
            iv = expand(iv, self.flank, self.resolution, self.rescale_flank)
        else:
            # same columns / filter as the reference (:296-321), but the distance filter runs BEFORE the new
            # columns are attached: filtering a freshly widened frame makes pandas re-consolidate every block
            c1 = (iv["start1"].values + iv["end1"].values) / 2
            c2 = (iv["start2"].values + iv["end2"].values) / 2
            absd = np.abs(c2 - c1)
            keep = (self.mindist <= absd) & (absd <= self.maxdist)
            if not keep.all():
                iv = iv[keep].reset_index(drop=True)
                c1, c2 = c1[keep], c2[keep]
            elif not (isinstance(iv.index, pd.RangeIndex) and iv.index.start == 0 and iv.index.step == 1):
                iv = iv.reset_index(drop=True)          # (the reference renumbers whether or not the filter dropped a row, :321)
            iv["chrom1"] = iv["chrom1"].astype(str)
            iv["chrom2"] = iv["chrom2"].astype(str)
            iv["center1"] = c1
            iv["center2"] = c2
            iv["distance"] = c2 - c1
            iv = expand2D(iv, self.flank, self.resolution, self.rescale_flank)
        self.intervals = iv

        if iv.shape[0] == 0:
            warnings.warn("No regions in features (maybe all below mindist?), returning empty output", stacklevel=2)
            self.pos_stream = self.empty_stream
            self.final_chroms = []
            return

        if self.nshifts > 0 and self.kind == "bedpe":
            iv = self._control_regions(iv)   # nshifts=0: only tags kind="ROI"

        if self.kind == "bed":
            base = set(iv["chrom"])
        else:
            if self.local:
                raise ValueError("Can't make local with both sides of loops defined")
            u1, u2 = set(iv["chrom1"].unique().tolist()), set(iv["chrom2"].unique().tolist())
            base = (u1 | u2) if self.trans else (u1 & u2)
        self.intervals = iv
        # (final_chroms are settled before the sort, as in the reference; an error below leaves the unsorted frame behind)
        basechroms = natsorted(list(base))
        final = basechroms if (isinstance(self.chroms, str) and self.chroms == "all") else \
            natsorted(list(set(self.chroms).intersection(set(basechroms))))
        if len(final):
            self.intervals = self._binnify(iv)
        self._finish_process(base)

    def _subset(self, df):
        if self.seed is not None:
            np.random.seed(self.seed)
        if 0 < self.subset < len(df):
            return df.sample(self.subset)
        return df

    def bedpe2bed(self, df, ends=True, how="center"):
        """Pairs -> BED intervals (reference :463-487).  ends=True: both anchors of every pair, sorted by
        (chrom, start, end).  Otherwise one interval per pair: how="outer" spans start1..end2, how="inner" the gap
        end1..start2; how="center" raises TypeError as it does upstream (np.mean is called there with two arrays)."""
        if ends:
            anchors = {name: np.concatenate([df[name + "1"].to_numpy(), df[name + "2"].to_numpy()])
                       for name in ("chrom", "start", "end")}
            order = np.lexsort((anchors["end"], anchors["start"], anchors["chrom"].astype(str)))
            return pd.DataFrame({name: col[order] for name, col in anchors.items()})
        spans = {"outer": ("start1", "end2"), "inner": ("end1", "start2")}
        if how not in spans:
            raise TypeError(f"bedpe2bed(how={how!r}) is not usable (upstream's 'center' branch cannot run either)")
        lo, hi = spans[how]
        return pd.DataFrame({"chrom": df["chrom1"].to_numpy(), "start": df[lo].to_numpy(), "end": df[hi].to_numpy()},
                            index=df.index)

    def _binnify(self, intervals):
        """Sort and convert expanded coordinates to bins (reference :489-527). pandas does the sort so the
        row order (hence the control-shift assignment) is the reference's."""
        res = self.resolution
        if self.kind == "bed":
            intervals = intervals.sort_values(["chrom", "start"])
            sides = [""]
        else:
            intervals = intervals.sort_values(["chrom1", "chrom2", "start1", "start2"])
            sides = ["1", "2"]
        for s in sides:
            intervals["stBin" + s] = np.floor(intervals["exp_start" + s] / res).astype(int)
            intervals["endBin" + s] = np.ceil(intervals["exp_end" + s] / res).astype(int)
            intervals["exp_start" + s] = intervals["stBin" + s] * res
            intervals["exp_end" + s] = intervals["endBin" + s] * res
        return intervals

    # -- random-shift controls (reference :387-453) -----------------------------------------------------
    def _control_regions(self, intervals2d, nshifts=0):
        """DataFrame form (API compatibility).  Issues exactly the reference's RNG calls."""
        if nshifts <= 0:
            intervals2d["kind"] = "ROI"
            return intervals2d
        cols = _Cols.from_frame(intervals2d)
        out = self._control_cols(cols, nshifts)
        df = out.frame()
        df["kind"] = np.where(out["kind"] == KIND_ROI, "ROI", "control")
        return df

    def _draw_raw(self, m):
        """The reference's RNG calls for m control windows (:420-436), in its order, un-multiplied: (|shift|, sign) of the
        draw that moves the BINS of both sides (the second pair of a trans pile-up, which only moves bp columns, is drawn
        and dropped).  With a draw-ahead running (see _DrawAhead) the numbers come from its queue: same calls, same order."""
        ahead = getattr(self, "_draw_ahead", None)
        if ahead is not None:
            return ahead.take(m)
        return self._draw_raw_now(m)

    def _draw_dtype(self):
        return np.int32 if max(abs(int(self.minshift)), abs(int(self.maxshift))) < 2 ** 31 else np.int64

    def _draw_raw_now(self, m, out=None):
        narrow = self._draw_dtype()
        shift = _draw_ints(self.minshift, self.maxshift, m, dtype=narrow, out=None if out is None else out[0])
        sign = _draw_signs(m, dtype=narrow, out=None if out is None else out[1])
        if self.trans:
            _draw_ints(self.minshift, self.maxshift, m, discard=True)
            _draw_ints(0, 2, m, discard=True)
        return shift, sign

    def _draw_shifts(self, m):
        """The reference's RNG calls for m control windows (:420-436), in its order: (shift, shift2) in bp."""
        shift = _draw_ints(self.minshift, self.maxshift, m)
        sign = _draw_signs(m)
        shift *= sign
        if self.trans:   # the two sides move independently in bp ...
            shift2 = _draw_ints(self.minshift, self.maxshift, m)
            sign2 = _draw_signs(m)
            shift2 = shift2 * sign2
        else:
            shift2 = shift
        return shift, shift2

    def _control_cols(self, cols, nshifts):
        """Column-table form: returns ROI rows followed by nshifts shifted copies, with an int8 'kind'."""
        n = len(cols)
        if nshifts <= 0:
            out = _Cols(cols)
            out["kind"] = np.full(n, KIND_ROI, np.int8)
            return out
        m = n * nshifts
        if getattr(self, "_draw_only", False):
            # a region another rank piles up: its windows are not needed here, the generator state after them is
            if m:
                self._draw_shifts(m)
            return _Cols()
        ctrl = cols.tiled(nshifts)
        shift, shift2 = self._draw_shifts(m)
        for name in ("exp_start1", "exp_end1", "center1"):
            if name in ctrl:
                ctrl[name] = ctrl[name] + shift
        for name in ("exp_start2", "exp_end2", "center2"):
            if name in ctrl:
                ctrl[name] = ctrl[name] + shift2
        # ... but the BINS of both sides move by `shift` (reference :442-445)
        dbin = np.round(shift / self.resolution).astype(int)
        dbin32 = dbin.astype(np.int32)
        for name in ("stBin1", "endBin1", "stBin2", "endBin2"):
            ctrl[name] = ctrl[name] + (dbin32 if ctrl[name].dtype == np.int32 else dbin)
        roi = _Cols(cols)
        roi["kind"] = np.full(n, KIND_ROI, np.int8)
        ctrl["kind"] = np.full(m, KIND_CONTROL, np.int8)
        return _Cols.concat(roi, ctrl)

    # -- region filters (reference :529-596) --------------------------------------------------------------
    def filter_func_all(self, intervals):
        return intervals

    def _filter_func_chrom(self, intervals, chrom):
        return intervals[intervals["chrom"] == chrom].reset_index(drop=True)

    def _filter_func_pairs_chrom(self, intervals, chrom):
        return intervals[(intervals["chrom1"] == chrom) & (intervals["chrom2"] == chrom)].reset_index(drop=True)

    def filter_func_chrom(self, chrom):
        f = self._filter_func_chrom if self.kind == "bed" else self._filter_func_pairs_chrom
        return partial(f, chrom=chrom)

    def _filter_func_region(self, intervals, region):
        chrom, start, end = region
        keep = (intervals["chrom"] == chrom) & (intervals["start"] >= start) & (intervals["end"] < end)
        return intervals[keep].reset_index(drop=True)

    def _filter_func_pairs_region(self, intervals, region):
        chrom, start, end = region
        keep = ((intervals["chrom1"] == chrom) & (intervals["chrom2"] == chrom)
                & (intervals["start1"] >= start) & (intervals["end1"] < end)
                & (intervals["start2"] >= start) & (intervals["end2"] < end))
        return intervals[keep].reset_index(drop=True)

    def _filter_func_trans_pairs(self, intervals, region1, region2):
        c1, s1, e1 = region1
        c2, s2, e2 = region2
        fwd = ((intervals["chrom1"] == c1) & (intervals["chrom2"] == c2)
               & (intervals["start1"] >= s1) & (intervals["end1"] < e1)
               & (intervals["start2"] >= s2) & (intervals["end2"] < e2))
        rev = ((intervals["chrom2"] == c1) & (intervals["chrom1"] == c2)
               & (intervals["start2"] >= s1) & (intervals["end2"] < e1)
               & (intervals["start1"] >= s2) & (intervals["end1"] < e2))
        # rows in the opposite chromosome order are selected but NOT swapped (reference :578-585)
        return pd.concat([intervals[fwd].reset_index(drop=True), intervals[rev].reset_index(drop=True)])

    def filter_func_trans_pairs(self, region1, region2):
        return partial(self._filter_func_trans_pairs, region1=region1, region2=region2)

    def filter_func_region(self, region):
        f = self._filter_func_region if self.kind == "bed" else self._filter_func_pairs_region
        return partial(f, region=region)

    # -- cached numpy views for fast region selection / grouping -------------------------------------------------
    def _cache(self):
        """Integer chromosome codes and coordinate arrays of the processed table (rebuilt if the table is replaced)."""
        c = self._fc
        tbl = self._tbl
        if c is not None and c["id"] is tbl:
            return c
        codes, names = tbl.chrom_codes()
        c = {"id": tbl, "cols": {}, "gc": {}, "chrom_code": {str(u): i for i, u in enumerate(names)}}
        if self.kind == "bedpe":
            c["c1"], c["c2"] = codes
            for k in ("start1", "end1", "start2", "end2"):
                c[k] = tbl.col(k)
        else:
            c["c"] = codes[0]
            c["start"], c["end"] = tbl.col("start"), tbl.col("end")
        self._fc = c
        return c

    def _col(self, name):
        c = self._cache()
        v = c["cols"].get(name)
        if v is None:
            is_bin = name.startswith(("stBin", "endBin"))
            v = self._tbl.bins32(name) if is_bin else None
            if v is None:
                v = self._tbl.col(name)
                if is_bin and v.dtype.kind in "iu" and len(v) and -2**31 < int(v.min()) and int(v.max()) < 2**31 - 2**24:
                    v = v.astype(np.int32)      # bins: every later pass (tile x nshifts, shift, filter) moves half the bytes
            c["cols"][name] = v
        return v

    def group_codes(self, name):
        """(int codes over the table's rows, uniques) of a grouping column.  Paired bedpe columns X1 / X2 share
        one dictionary (needed when flipped snippets swap them); bed column X serves X1 and X2."""
        c = self._cache()
        if name in c["gc"]:
            return c["gc"][name]
        tbl = self._tbl
        stem = name[:-1]
        if self.kind == "bedpe" and name[-1:] in "12" and tbl.has(stem + "1") and tbl.has(stem + "2"):
            (ca, cb), uniq = tbl.group_codes(stem + "1", stem + "2")
            c["gc"][stem + "1"] = (ca, uniq)
            c["gc"][stem + "2"] = (cb, uniq)
        else:
            (ca,), uniq = tbl.group_codes(name)
            c["gc"][name] = (ca, uniq)
        return c["gc"][name]

    def _pair_bucket(self, k1, k2):
        """Rows of the table with (chrom1, chrom2) codes (k1, k2), ascending: a slice of the sorted table (the array path sorts
        by chromosome pair first, so a pair's rows are one run), else an index array from one stable sort of the pair codes —
        selecting the rows of every region (pair) costs O(rows of that pair), not O(all rows)."""
        c = self._cache()
        if "pair_nc" not in c:
            nc = len(c["chrom_code"]) + 1
            code = c["c1"].astype(np.int64) * nc + c["c2"]
            c["pair_nc"] = nc
            runs = None
            if self._tbl.sorted_pairs:
                cuts = np.flatnonzero(code[1:] != code[:-1]) + 1
                bounds = np.concatenate([[0], cuts, [len(code)]])
                heads = code[bounds[:-1]] if len(code) else code[:0]
                if len(np.unique(heads)) == len(heads):
                    runs = {int(h): slice(int(a), int(b)) for h, a, b in zip(heads, bounds[:-1], bounds[1:])}
            c["pair_runs"] = runs
            if runs is None:
                c["pair_order"] = np.argsort(code, kind="stable")
                c["pair_ptr"] = np.concatenate([[0], np.cumsum(np.bincount(code, minlength=nc * nc))])
        if k1 < 0 or k2 < 0:
            return slice(0, 0)
        code = k1 * c["pair_nc"] + k2
        if c["pair_runs"] is not None:
            return c["pair_runs"].get(code, slice(0, 0))
        return c["pair_order"][c["pair_ptr"][code]:c["pair_ptr"][code + 1]]

    def _rows_pairs_region(self, region):
        chrom, start, end = region
        c = self._cache()
        memo = c.setdefault("rows_pairs", {})
        hit = memo.get(region)
        if hit is not None:
            return hit
        k = c["chrom_code"].get(str(chrom), -1)
        rows = self._pair_bucket(k, k)
        m = ((c["start1"][rows] >= start) & (c["end1"][rows] < end) & (c["start2"][rows] >= start)
             & (c["end2"][rows] < end))
        memo[region] = out = rows if m.all() else _rows_where(rows, m)
        return out

    def _rows_trans_pairs(self, region1, region2):
        c1n, s1, e1 = region1
        c2n, s2, e2 = region2
        c = self._cache()
        k1, k2 = c["chrom_code"].get(str(c1n), -1), c["chrom_code"].get(str(c2n), -1)
        f = self._pair_bucket(k1, k2)
        fwd = _rows_where(f, (c["start1"][f] >= s1) & (c["end1"][f] < e1) & (c["start2"][f] >= s2) & (c["end2"][f] < e2))
        r = self._pair_bucket(k2, k1)
        rev = _rows_where(r, (c["start2"][r] >= s1) & (c["end2"][r] < e1) & (c["start1"][r] >= s2) & (c["end1"][r] < e2))
        if len(rev) == 0 and isinstance(f, slice) and len(fwd) == f.stop - f.start:
            return f
        return np.concatenate([fwd, rev])      # same order as the reference's concat

    def _rows_region(self, region):
        chrom, start, end = region
        c = self._cache()
        k = c["chrom_code"].get(str(chrom), -1)
        return np.flatnonzero((c["c"] == k) & (c["start"] >= start) & (c["end"] < end))

    def feature_ids(self):
        """(uid, rep) of a BED table: uid[row] numbers the distinct (chrom, start, end) features, rep[u] is a row holding feature
        u.  By-window pile-ups group every window under the feature on each of its sides (lib/puputils.py:218-223): the integer
        stands in for the reference's (chrom, start, end) tuple until the output frame is written."""
        c = self._cache()
        if "uid" not in c:
            order = np.lexsort((c["end"], c["start"], c["c"]))
            cs, ss, es = c["c"][order], c["start"][order], c["end"][order]
            new = np.ones(len(order), bool)
            new[1:] = (cs[1:] != cs[:-1]) | (ss[1:] != ss[:-1]) | (es[1:] != es[:-1])
            uid = np.empty(len(order), np.int64)
            uid[order] = np.cumsum(new) - 1
            c["uid"], c["uid_rep"] = uid, order[new]
        return c["uid"], c["uid_rep"]

    def _take(self, rows, names, suffix=""):
        """Column table of the given rows. '_gc_X' names yield the integer group codes of column X, '_uid' the feature ids."""
        out = {}
        for name in names:
            if name == "_uid":
                out[name + suffix] = self.feature_ids()[0][rows]
            elif name.startswith("_gc_"):
                out[name + suffix] = self.group_codes(name[4:])[0][rows]
            else:
                out[name + suffix] = self._col(name)[rows]
        return out

    # -- column-table window generation ---------------------------------------------------------------------
    def _needed(self, want):
        """Columns to carry per snippet: bins + what grouping/flipping asks for (None = everything)."""
        if want is None:
            return None
        base = ["stBin1", "endBin1", "stBin2", "endBin2"]
        return base + [c for c in want if c not in base]

    def skip_region(self, region1, region2=None, control=False):
        """Advance the control-shift generator past one region (pair) exactly as generating its windows would, without
        building them: every rank of a multi-GPU run walks all regions in order and draws-and-discards for the ones it
        does not own, so the shifts of its own regions are those of the reference's single sequence (:387-453)."""
        if not control or self.nshifts <= 0:
            return
        self._draw_only = True
        try:
            self.region_table(region1, region2, control=True, columns=())
        finally:
            self._draw_only = False

    def region_weight(self, region1, region2=None):
        """Cheap, deterministic estimate of the number of ROI windows of a region (pair): what the ranks balance."""
        if not hasattr(self, "kind") or self._tbl.n == 0:
            return 0
        if self.kind == "bedpe":
            rows = self._rows_trans_pairs(tuple(region1), tuple(region2)) if self.trans \
                else self._rows_pairs_region(tuple(region1))
            return _nrows(rows)
        nl = len(self._rows_region(tuple(region1)))
        if self.local:
            return nl
        if region2 is None or tuple(region2) == tuple(region1):
            return nl * (nl - 1) // 2
        return nl * len(self._rows_region(tuple(region2)))

    def region_table(self, region1, region2=None, control=False, columns=()):
        """All windows of one region (pair) as a column table with 'kind' (0 ROI / 1 control).

        region1/region2: (chrom, start, end).  ``columns``: extra columns to carry (e.g. 'distance',
        'strand1'); None carries every column.  Equivalent of running ``pos_stream`` for the region
        (reference get_intervals_stream :716-746 / get_combinations :598-714) up to, not including,
        ``modify_2Dintervals_func`` and ``assign_groups``.
        """
        if not hasattr(self, "kind") or self._tbl.n == 0 or self.pos_stream == self.empty_stream:
            return None
        nshifts = self.nshifts * bool(control)
        have = set(self._tbl.names)
        if self.kind == "bedpe":
            rows = self._rows_trans_pairs(tuple(region1), tuple(region2)) if self.trans \
                else self._rows_pairs_region(tuple(region1))
            if _nrows(rows) == 0:
                return None
            keep = self._needed(columns)
            names = list(self._tbl.names) if keep is None else \
                [c for c in keep if c in have or (c.startswith("_gc_") and c[4:] in have)]
            return self._control_cols(_Cols(self._take(rows, names)), nshifts)
        # ---- bed: combinations ----
        rows_l = self._rows_region(tuple(region1))
        rows_r = rows_l if region2 is None or tuple(region2) == tuple(region1) else self._rows_region(tuple(region2))
        return self._combination_table(rows_l, rows_r, nshifts, columns)

    def _combination_table(self, rows_l, rows_r, nshifts, columns):
        """Windows of all feature pairs (left feature from rows_l, right one from rows_r; row ids into self.intervals)."""
        have = set(self._tbl.names)
        want = None if columns is None else set(columns) | {"center1", "center2"}

        def side(rows, s):
            if want is None:
                names = list(self._tbl.names)
            else:
                names = [c for c in self._tbl.names if c + s in want or c in ("stBin", "endBin", "center")]
                names += ["_gc_" + w[4:-1] for w in want if w.startswith("_gc_") and w.endswith(s) and w[4:-1] in have]
                names += ["_uid"] if "_uid" + s in want else []
            return self._take(rows, names, suffix=s)

        left, right = rows_l, rows_r
        L, R = side(left, "1"), side(right, "2")
        if self.local:
            tbl = _Cols({**L, **{k[:-1] + "2": v for k, v in L.items()}})
            return self._control_cols(tbl, nshifts)
        parts = []
        if self.trans:
            nl, nr = len(left), len(right)
            if nshifts == 0:
                x = np.repeat(np.arange(nl), nr)
                y = np.tile(np.arange(nr), nl)
                tbl = _Cols({**{k: v[x] for k, v in L.items()}, **{k: v[y] for k, v in R.items()}})
                return self._control_cols(tbl, 0) if len(tbl) else None
            for x, y in itertools.product(range(nl), range(nr)):   # one RNG draw set per pair, as the reference
                tbl = _Cols({**{k: v[x:x + 1] for k, v in L.items()}, **{k: v[y:y + 1] for k, v in R.items()}})
                parts.append(self._control_cols(tbl, nshifts))
        else:
            m = len(left)                      # cis: right is left (reference :614-615)
            c1, c2 = L["center1"], R["center2"]
            # the reference loops offsets up to the TOTAL number of features (:682); offsets >= m select
            # nothing and draw nothing
            # features sorted by centre (the usual BED case): the separation at offset i only grows with i, so the first
            # offset at which every pair is beyond maxdist ends the walk (the reference keeps looping; it finds nothing there)
            ordered = m > 1 and bool(np.all(c1[1:] >= c1[:-1]))
            if m > 1 and right is left and not getattr(self, "_draw_only", False) and m < 65_536 \
                    and not os.environ.get("COOLPUPPY_AMD_WALK_COMBINATIONS"):
                if nshifts == 0:
                    return self._combination_table_sorted(L, R, c1, m)
                got = self._combination_table_sorted(L, R, c1, m, nshifts)
                if got is not None:
                    return got                     # (None: no pair at all, or the library declined the draws — nothing was drawn: the walk)
            for i in range(1, min(self._tbl.n, m)):
                k = m - i
                dist = c2[i:i + k] - c1[:k]
                ok = (self.mindist <= np.abs(dist)) & (np.abs(dist) <= self.maxdist)
                a = np.flatnonzero(ok)
                if len(a) == 0:
                    if ordered and dist.min() > self.maxdist:
                        break
                    continue                   # size-0 RNG draws do not advance the generator
                if getattr(self, "_draw_only", False):
                    if nshifts > 0:
                        self._draw_shifts(len(a) * nshifts)
                    continue
                tbl = _Cols({**{kk: v[a] for kk, v in L.items()}, **{kk: v[a + i] for kk, v in R.items()}})
                tbl["distance"] = dist[a]
                parts.append(self._control_cols(tbl, nshifts))
        parts = [p for p in parts if len(p)]
        if not parts:
            return None
        out = _Cols({k: np.concatenate([p[k] for p in parts]) for k in parts[0]})
        return out

    def _combination_table_sorted(self, L, R, c, m, nshifts=0):
        """_combination_table's walk over the offsets (no control draws), without the walk: the pairs (k, j > k) whose centres lie
        mindist ... maxdist apart are found through the centres' sort order — for every feature the partners beyond it in centre
        are a contiguous stretch of that order (two bisections) — and the reference's order (offset i = j - k ascending, k
        ascending inside an offset: coolpup.py:682-700) is two stable 16-bit sorts of the pairs.  (The walk makes one small table
        per offset: thousands of them for a by-window pile-up of a chromosome's CTCF sites, 48 of the 120 ms of such a call.)"""
        srt = np.argsort(c, kind="stable")
        cs = c[srt]
        pos = np.arange(m)
        lo = np.searchsorted(cs, cs + max(float(self.mindist), 0.0), side="left")
        hi = np.searchsorted(cs, cs + self.maxdist, side="right") if np.isfinite(self.maxdist) else np.full(m, m)
        # (one candidate more on either side: the bounds are tested below in the walk's own arithmetic, c[j] - c[k] against them)
        lo = np.maximum(lo - 1, pos + 1)
        hi = np.minimum(hi + 1, m)
        cnt = np.maximum(hi - lo, 0)
        total = int(cnt.sum())
        if total == 0:
            return None
        pa = np.repeat(pos, cnt)
        pb = np.arange(total) - np.repeat(np.cumsum(cnt) - cnt, cnt) + np.repeat(lo, cnt)
        ia, ib = srt[pa], srt[pb]
        k, j = np.minimum(ia, ib), np.maximum(ia, ib)
        dist = R["center2"][j] - L["center1"][k]
        ok = (self.mindist <= np.abs(dist)) & (np.abs(dist) <= self.maxdist) & ((j - k) < min(self._tbl.n, m))
        if not ok.all():
            k, j, dist = k[ok], j[ok], dist[ok]
            if len(k) == 0:
                return None
        o1 = np.argsort(k.astype(np.uint16), kind="stable")                    # (16-bit keys: numpy's radix sort)
        o2 = np.argsort((j - k)[o1].astype(np.uint16), kind="stable")
        order = o1[o2]
        a, b = k[order], j[order]
        if nshifts <= 0:
            tbl = _Cols({**{kk: v[a] for kk, v in L.items()}, **{kk: v[b] for kk, v in R.items()}})
            tbl["distance"] = dist[order]
            return self._control_cols(tbl, 0)
        return self._combination_controls(L, R, a, b, dist[order], (b - a), nshifts)

    def _combination_controls(self, L, R, a, b, dist, off, nshifts):
        """The control copies of _combination_table's walk, all offsets at once.  The walk hands every offset's table to _control_cols:
        its ROI rows, then nshifts shifted copies of them, drawn with ONE randint + ONE choice per offset (coolpup.py:420-436 inside the
        loop of :682-700) — thousands of small draws and small tables for a by-window pile-up with controls (0.18 of its 1.0 s).  Here the
        draws of all offsets are one library job (engine.legacy_randint_plan: the same calls in the same order, so the same numbers and
        generator state) and the output rows are gathered in one go: block i of the output = rows [s_i, s_i + n_i) of the pair table
        once as they are, then nshifts times shifted.  None: the library declined the draws (the caller walks instead)."""
        from .engine import legacy_randint_plan
        P = len(a)
        n_off = np.bincount(off)                                   # pairs per offset; the walk skips the empty ones (no draw)
        n_off = n_off[n_off > 0]
        s_off = np.cumsum(n_off) - n_off                            # first pair of every offset (the pairs are in offset order)
        shift = np.empty(P * nshifts, np.int64)
        sign = np.empty(P * nshifts, np.int64)
        lo, hi = int(self.minshift), int(self.maxshift)
        calls = []
        for s, n in zip((s_off * nshifts).tolist(), (n_off * nshifts).tolist()):
            calls.append((lo, hi, n, 1, 0, shift[s:s + n]))
            calls.append((0, 2, n, 2, -1, sign[s:s + n]))
        if not legacy_randint_plan(calls):
            return None
        shift *= sign
        # output row o lies in block i = the offset whose rows it copies; inside the block: n_i ROI rows, then nshifts x n_i controls
        blk = (1 + nshifts) * s_off
        o = np.arange(P * (1 + nshifts))
        i = np.repeat(np.arange(len(n_off)), n_off * (1 + nshifts))
        within = o - blk[i]
        ni = n_off[i]
        src = s_off[i] + within % ni
        ctrl = within >= ni
        sh = np.where(ctrl, shift[np.where(ctrl, nshifts * s_off[i] + within - ni, 0)], 0)
        rows_a, rows_b = a[src], b[src]
        out = _Cols({**{kk: v[rows_a] for kk, v in L.items()}, **{kk: v[rows_b] for kk, v in R.items()}})
        out["distance"] = dist[src]
        for name in ("exp_start1", "exp_end1", "center1", "exp_start2", "exp_end2", "center2"):      # (cis: both sides move by `shift`)
            if name in out:
                out[name] = out[name] + sh
        dbin = np.round(sh / self.resolution).astype(int)           # the BINS of both sides move by `shift` too (reference :442-445)
        dbin32 = dbin.astype(np.int32)
        for name in ("stBin1", "endBin1", "stBin2", "endBin2"):
            out[name] = out[name] + (dbin32 if out[name].dtype == np.int32 else dbin)
        out["kind"] = np.where(ctrl, np.int8(KIND_CONTROL), np.int8(KIND_ROI))
        return out

    # -- dict-row streams kept for API compatibility (reference :598-749) -----------------------------------
    def get_intervals_stream(self, filter_func1, filter_func2=None, intervals=None, control=False, groupby=[],
                             modify_2Dintervals_func=None):
        """Per-snippet dict rows, as the reference yields them (bedpe). Slow path; the engine uses region_table."""
        if intervals is None:
            intervals = self.intervals
        intervals = filter_func1(intervals)
        intervals = self._control_regions(intervals, self.nshifts * control)
        if modify_2Dintervals_func is not None:
            intervals = modify_2Dintervals_func(intervals)
        intervals = assign_groups(intervals, groupby)
        intervals = intervals.reindex(columns=list(intervals.columns) + ["data", "cov_start", "cov_end",
                                                                        "horizontal_stripe", "vertical_stripe"])
        if not len(intervals) >= 1:
            logger.debug("Empty selection")
            yield None
        for row in intervals.to_dict(orient="records"):
            yield row

    def get_combinations(self, filter_func1, filter_func2=None, intervals=None, control=False, groupby=[],
                         modify_2Dintervals_func=None):
        """Per-snippet dict rows of the pairwise combinations of BED features, as the reference yields them
        (:598-714).  Slow path kept for the low-level API; built on the same column tables the engine path uses."""
        if intervals is None:
            intervals = self.intervals
        if not len(intervals) >= 1:
            logger.debug("Empty selection")
            yield None
            return
        if intervals is not self.intervals:
            raise NotImplementedError("get_combinations works on the CoordCreator's own intervals")
        tagged = intervals.assign(_row=np.arange(len(intervals)))
        rows_l = filter_func1(tagged)["_row"].values
        rows_r = rows_l if filter_func2 is None else filter_func2(tagged)["_row"].values
        tbl = self._combination_table(rows_l, rows_r, self.nshifts * bool(control), None)
        if tbl is None or len(tbl) == 0:
            return
        frame = pd.DataFrame({k: v for k, v in tbl.items() if not k.startswith("_gc_")})
        frame["kind"] = np.where(tbl["kind"] == KIND_ROI, "ROI", "control")
        if modify_2Dintervals_func is not None:
            frame = modify_2Dintervals_func(frame)
        frame = assign_groups(frame, groupby)
        frame = frame.reindex(columns=list(frame.columns) + ["data", "cov_start", "cov_end", "horizontal_stripe",
                                                             "vertical_stripe"])
        for row in frame.to_dict(orient="records"):
            yield row

    def empty_stream(self, *args, **kwargs):
        yield from ()


# ------------------------------------------------------------------------------------------------------
# PileUpper
# ------------------------------------------------------------------------------------------------------
def _view_signature(chromsizes):
    """What a validated view frame was validated against: the chromosome names and sizes."""
    return (tuple(str(c) for c in chromsizes.index), tuple(int(v) for v in chromsizes.values))


# frames _make_cooler_view / _make_viewframe have produced, by identity (weak references): pileup() validates the view and hands the
# SAME object to PileUpper, which would validate it again — a third of a small call's set-up.  (Not DataFrame.attrs: pandas deep-copies
# them into every derived frame.)
_VALID_VIEWS = []


def _remember_view(df, sig):
    import weakref
    _VALID_VIEWS[:] = [(r, s) for r, s in _VALID_VIEWS[-3:] if r() is not None] + [(weakref.ref(df), sig)]
    return df


def _make_cooler_view(clr, remember=False):
    """cooltools.lib.common.make_cooler_view: one whole-chromosome region per chromosome.  remember: pileup()'s own local frame —
    valid by construction, seen by nobody else — need not be validated again by the PileUpper it is handed to."""
    names = [str(c) for c in clr.chromnames]
    df = pd.DataFrame({"chrom": names, "start": 0, "end": [int(clr.chromsizes[c]) for c in clr.chromnames], "name": list(names)})
    return _remember_view(df, _view_signature(clr.chromsizes)) if remember else df


def _make_viewframe(view_df, chromsizes):
    """bioframe.make_viewframe for the DataFrame case: chrom/start/end[/name], bounds-checked.  A frame pileup() has just made or
    validated for the same chromosomes (_remember_view: the very object, a local of pileup() nobody else holds) is handed back as a
    copy; frames that become public attributes are never remembered — their owner may edit them."""
    sig = _view_signature(chromsizes)
    for ref, s in _VALID_VIEWS:
        if ref() is view_df and s == sig and list(view_df.columns) == ["chrom", "start", "end", "name"] \
                and isinstance(view_df.index, pd.RangeIndex):
            return view_df.copy()
    df = pd.DataFrame(view_df).copy()
    if "chrom" not in df.columns:
        df.columns = ["chrom", "start", "end", "name"][: df.shape[1]] + list(df.columns[4:])
    if "name" not in df.columns:
        df["name"] = df["chrom"].astype(str)
    df["chrom"] = df["chrom"].astype(str)
    df = df[["chrom", "start", "end", "name"]].reset_index(drop=True)
    if df["name"].duplicated().any():
        raise ValueError("view_df is not a valid viewframe: region names are not unique")
    sizes = dict(zip(*sig))
    for r in zip(df["chrom"].tolist(), df["start"].tolist(), df["end"].tolist(), df["name"].tolist()):
        if r[0] not in sizes or r[1] < 0 or r[2] > sizes[r[0]] or r[1] >= r[2]:
            raise ValueError(f"view_df is not a valid viewframe or incompatible: region {r} out of bounds")
    return df


_ENGINES = {}


def _engine_for(clr, device_id, rows=None):
    """One resident pixel table per (cooler object, device): repeated pile-ups skip the upload.
    rows: sorted, disjoint (lo, hi) bin ranges — upload only the pixels of these rows (a rank of a multi-GPU run holds
    the rows of the regions it owns; every other row is present but empty).
    COOLPUPPY_AMD_VARIANT (int, see pup_set_tuning) selects a kernel variant — used by the tests to run whole
    pileup() calls through a kernel the engine would not pick for inputs that small."""
    from .engine import PileupEngine
    rows = None if rows is None else tuple((int(a), int(b)) for a, b in rows)
    key = (id(clr), device_id, rows)
    with _ENGINE_LOCK:                         # a prefetch (below) may be uploading this very table right now
        hit = _ENGINES.get(key)
        if hit is not None and hit[0] is clr:
            eng = hit[1]
        else:
            for k in [k for k, v in _ENGINES.items() if k[1] == device_id]:   # one table per device at a time
                _ENGINES.pop(k)[1].close()
            eng = PileupEngine(device_id)
            if rows is None and getattr(clr, "_pixel_source", None) is not None and not clr.pixels_in_memory:
                # a cooler opened with read_cool(..., stream_pixels=True): file -> page-locked slabs -> HBM, chunk by chunk
                from .cool_io import stream_pixels_into
                clr._last_stream_stats = stream_pixels_into(eng, clr)
            else:
                indptr, col, cnt = clr.pixel_table()
                if rows is not None:
                    indptr, col, cnt = _rows_of_table(indptr, col, cnt, rows)
                eng.load_pixels(indptr, col, cnt)
            eng.build_index(clr.chrom_offset)      # optional accelerator; False (too large) just means binary search
            _ENGINES[key] = (clr, eng)
        eng.set_tuning(0, int(os.environ.get("COOLPUPPY_AMD_VARIANT", "0") or 0))
    return eng


_ENGINE_LOCK = threading.RLock()


def _prefetch_engine(clr, device_id, rows=None):
    """Start moving the pixel table to the GPU (and building its index) on a helper thread while the caller works out
    the window coordinates: the copies and kernels release the interpreter, so the first pile-up on a table pays
    max(upload, coordinates) instead of their sum.  Errors surface when run_plan asks for the engine itself."""
    rows_key = None if rows is None else tuple((int(a), int(b)) for a, b in rows)
    hit = _ENGINES.get((id(clr), device_id, rows_key))
    if (hit is not None and hit[0] is clr) or os.environ.get("COOLPUPPY_AMD_NO_PREFETCH", "") == "1":
        return None

    def work():
        try:
            _engine_for(clr, device_id, rows)
        except Exception:           # noqa: BLE001 - reported by the foreground call
            pass
    t = threading.Thread(target=work, name="pup-table-upload", daemon=True)
    _PREFETCH_THREADS[:] = [x for x in _PREFETCH_THREADS if x.is_alive()]
    _PREFETCH_THREADS.append(t)
    t.start()
    return t


_PREFETCH_THREADS = []


def _shutdown_engines():
    """Interpreter exit, step 1: no helper thread may still be inside the HIP runtime (coolpuppy_amd.shutdown)."""
    while _PREFETCH_THREADS:
        t = _PREFETCH_THREADS.pop()
        if t.is_alive():
            t.join(timeout=60)


def _close_engines():
    """Interpreter exit, step 2 (after the RCCL communicators are gone): destroy the cached engines explicitly instead of
    leaving it to __del__ during module teardown, when the HIP runtime may already be unloading."""
    with _ENGINE_LOCK:
        for k in list(_ENGINES):
            try:
                _ENGINES.pop(k)[1].close()
            except Exception:       # noqa: BLE001 - shutting down
                pass


def _rows_of_table(indptr, col, cnt, rows):
    """The CSR table restricted to the given row ranges: same number of rows, the others empty."""
    indptr = np.asarray(indptr, np.int64)
    length = np.zeros(len(indptr) - 1, np.int64)
    for lo, hi in rows:
        length[lo:hi] = indptr[lo + 1:hi + 1] - indptr[lo:hi]
    new_ptr = np.concatenate([[0], np.cumsum(length)]).astype(np.int64)
    if not rows:
        return new_ptr, col[:0], cnt[:0]
    return (new_ptr, np.concatenate([col[indptr[lo]:indptr[hi]] for lo, hi in rows]),
            np.concatenate([cnt[indptr[lo]:indptr[hi]] for lo, hi in rows]))


def _merge_ranges(ranges):
    out = []
    for lo, hi in sorted(ranges):
        if out and lo <= out[-1][1]:
            out[-1][1] = max(out[-1][1], hi)
        else:
            out.append([lo, hi])
    return [(a, b) for a, b in out]


class PileUpper:
    """Creates pile-ups (reference coolpup.py:752-1919) with the per-snippet work on the GPU."""

    def __init__(self, clr, CC, *, view_df=None, clr_weight_name="weight", expected=False,
                 expected_value_col="balanced.avg", ooe=True, control=False, coverage_norm=False, rescale=False,
                 rescale_size=99, flip_negative_strand=False, ignore_diags=2, store_stripes=False, nproc=1):
        self.clr = clr
        self._aclr = as_array_cooler(clr)
        self.resolution = self.clr.binsize
        self.CC = CC
        assert self.resolution == self.CC.resolution
        self.__dict__.update(self.CC.__dict__)
        self.clr_weight_name = clr_weight_name
        self.expected = expected
        self.expected_value_col = expected_value_col
        self.ooe = ooe
        self.control = control
        self.pad_bins = self.CC.flank // self.resolution
        self.coverage_norm = coverage_norm
        self.rescale = rescale
        self.rescale_size = rescale_size
        self.flip_negative_strand = flip_negative_strand
        self.ignore_diags = ignore_diags
        self.store_stripes = store_stripes
        self.nproc = nproc

        if view_df is None:
            self.view_df = _make_cooler_view(clr)
        else:
            self.view_df = _make_viewframe(view_df, clr.chromsizes)
        self._expected_vectors = {}
        if self.expected is not None and self.expected is not False:
            exp = self.expected
            need = {"region1", "region2", self.expected_value_col}
            if not isinstance(exp, pd.DataFrame) or not need.issubset(exp.columns):
                raise ValueError("provided expected is not valid")
            if self.control:
                warnings.warn("Can't do both expected and control shifts; defaulting to expected", stacklevel=2)
                self.control = False
            # rows of the table by view region, without building the filtered frames (a by-diagonal table has one row per
            # region and diagonal: 3e5 rows for a genome; the frames `expected_df` names are assembled on first access)
            names = self.view_df["name"].tolist()
            where = {nm: i for i, nm in enumerate(names)}

            def view_index(col):
                from .engine import factorize_objects
                codes, uniq = factorize_objects(np.ascontiguousarray(np.asarray(exp[col].to_numpy(), dtype=object)))
                lut = np.array([where.get(u, -1) for u in uniq] + [-1], np.int64)
                return lut[codes]

            v1, v2 = view_index("region1"), view_index("region2")
            inview = (v1 >= 0) & (v2 >= 0)
            if self.trans:
                if len(exp) and inview.any() and bool(np.all(v1[inview] == v2[inview])):
                    raise ValueError("provided expected is not valid")
                self._expected_rows = inview
                rows = np.flatnonzero(inview)
                vals = exp[self.expected_value_col].to_numpy()[rows]
                self._trans_expected = {}
                for a, b, v in zip(v1[rows].tolist(), v2[rows].tolist(), vals.tolist()):
                    self._trans_expected.setdefault((names[a], names[b]), []).append(v)
            else:
                if "dist" not in exp.columns:
                    raise ValueError("provided expected is not valid")
                # by-diagonal vector of every view region in table order (ExpectedSnipper.select -> LazyToeplitz)
                self._expected_rows = inview & (v1 == v2)
                rows = np.flatnonzero(self._expected_rows)
                vv = v1[rows]
                order = np.argsort(vv, kind="stable")
                ptr = np.concatenate([[0], np.cumsum(np.bincount(vv, minlength=len(names)))])
                values = exp[self.expected_value_col].to_numpy()[rows].astype(np.float64)
                for i, name in enumerate(names):
                    if ptr[i + 1] == ptr[i]:
                        raise ValueError("provided expected is not valid")
                    self._expected_vectors[name] = values[order[ptr[i]:ptr[i + 1]]]
            self._expected_src = exp
            self.expected = True
        self.view_df = self.view_df.set_index("name")
        self.view_df_extents = {}
        self._global_extents = {}
        self._region_tuples = {}
        for region_name, chrom, start, end in zip(self.view_df.index.tolist(), self.view_df["chrom"].tolist(),
                                                  self.view_df["start"].tolist(), self.view_df["end"].tolist()):
            lo, hi = self._aclr.extent((chrom, start, end))
            chroffset = self._aclr.offset(chrom)
            self.view_df_extents[region_name] = lo - chroffset, hi - chroffset
            self._global_extents[region_name] = (lo, hi, chroffset)
            self._region_tuples[region_name] = (chrom, start, end)

        self.chroms = natsorted(list(set(self.CC.final_chroms) & set(self.clr.chromnames)))
        self.view_df = self.view_df[self.view_df["chrom"].isin(self.chroms)]
        if self.view_df["chrom"].unique().shape[0] == 0:
            raise ValueError(
                """No chromosomes are in common between the coordinate
                   file and the cooler file. Are they in the same
                   format, e.g. starting with "chr"?
                   """
            )
        if self.trans and self.view_df["chrom"].unique().shape[0] < 2:
            raise ValueError("Trying to do trans with fewer than two chromosomes")

        bins_columns = list(self._aclr.bins().columns)     # the adapter: it also holds columns computed here
        if self.coverage_norm is True:
            self.coverage_norm = "cov_tot_raw"
        elif self.coverage_norm == "cis":
            self.coverage_norm = "cov_cis_raw"
        elif self.coverage_norm == "total":
            self.coverage_norm = "cov_tot_raw"
        elif self.coverage_norm and self.coverage_norm not in bins_columns:
            raise ValueError(f"coverage_norm {self.coverage_norm} not found in cooler bins")
        if self.coverage_norm in ["cov_cis_raw", "cov_tot_raw"] and self.coverage_norm not in bins_columns:
            # the reference computes both columns with cooltools' coverage() and STORES them in the cooler
            # (coolpup.py:955-963); here the GPU computes them (K3) and they are kept in the cooler object
            from . import dist as _dist
            if not hasattr(self._aclr, "set_bins_column"):
                raise ValueError(f"cannot store the computed {self.coverage_norm!r} column in this cooler object")
            src = getattr(self, "_coverage_source", None)      # tests: an oracle stand-in with the same result
            if src is not None:
                cis, tot = src(self._aclr, self.ignore_diags)
            else:
                eng = _engine_for(self._aclr, _dist.local_device())
                cis, tot = eng.coverage(self._aclr.chrom_offset, ignore_diags=self.ignore_diags)
            self._aclr.set_bins_column("cov_cis_raw", cis)
            self._aclr.set_bins_column("cov_tot_raw", tot)
        if self.coverage_norm and self.clr_weight_name:
            raise ValueError("Can't do coverage normalization when clr_weight_name is provided")
        if self.rescale:
            if self.rescale_flank is None:
                raise ValueError("Cannot use rescale without setting rescale_flank")
            elif self.rescale_size % 2 == 0:
                raise ValueError("Please provide an odd rescale_size")
            logger.info(f"Rescaling with rescale_flank = {self.rescale_flank} to "
                        f"{self.rescale_size}x{self.rescale_size} pixels")
        elif self.rescale_flank is not None:
            raise ValueError("rescale_flank is set on the CoordCreator but rescale=False: windows would not be "
                             "2*pad_bins+1 bins wide")
        if self.ignore_diags is None or self.ignore_diags < 0:
            raise ValueError("ignore_diags must be >= 0 (the engine reads the upper-triangular pixel table)")

        self.empty_outmap = self.make_outmap()
        self.empty_pup = {
            "data": self.empty_outmap, "horizontal_stripe": [], "vertical_stripe": [], "n": 0,
            "num": self.empty_outmap, "cov_start": np.zeros((self.empty_outmap.shape[0])),
            "cov_end": np.zeros((self.empty_outmap.shape[1])), "coordinates": [],
        }

    # -- small pieces kept from the reference API ------------------------------------------------------------
    @property
    def expected_df(self):
        """The rows of the caller's expected table this pile-up uses (reference attribute): both regions in the view, and for
        cis pile-ups region1 == region2.  Assembled on first access."""
        if "_expected_df" not in self.__dict__:
            self.__dict__["_expected_df"] = self._expected_src[self._expected_rows].reset_index(drop=True)
        return self.__dict__["_expected_df"]

    def get_expected_trans(self, region1, region2):
        got = self._trans_expected.get((region1, region2), [])
        if len(got) != 1:
            raise ValueError("can only convert an array of size 1 to a Python scalar")
        return got[0]

    @property
    def intervals(self):
        """The CoordCreator's processed feature frame (the reference copies the attribute over, coolpup.py:838)."""
        return self.CC.intervals

    def make_outmap(self):
        n = self.rescale_size if self.rescale else 2 * self.pad_bins + 1
        return np.zeros((n, n))

    def _region_tuple(self, name):
        """(chrom, start, end) of a view region; cached — a trans pile-up asks 2 x 253 times."""
        cache = self.__dict__.setdefault("_region_tuples", {})
        if name not in cache:
            cache[name] = tuple(self.view_df.loc[name, ["chrom", "start", "end"]].tolist())
        return cache[name]

    # -- host side of pileup_region: windows of one region (pair) as engine inputs ----------------------------
    def _region_pairs(self):
        if self.trans:
            r1, r2 = [], []
            chrom_of = dict(zip(self.view_df.index.tolist(), self.view_df["chrom"].tolist()))
            for a, b in itertools.combinations(self.view_df.index, 2):
                if chrom_of[a] != chrom_of[b]:
                    r1.append(a)
                    r2.append(b)
            return list(zip(r1, r2))
        return [(r, r) for r in self.view_df.index]

    def _group_source(self, g):
        """Column to carry for grouping by g: integer codes ('_gc_' + g) for string-like feature columns, the
        column itself otherwise; plus the decoder that turns a stored value back into the key element."""
        tbl = self.CC._tbl
        base = g if self.CC.kind == "bedpe" else g[:-1]
        if tbl.has(base) and tbl.dtype(base) == object:
            uniq = self.CC.group_codes(g if self.CC.kind == "bedpe" else base)[1]
            return "_gc_" + g, (lambda i, u=uniq: u[i])
        return g, None

    def region_snippets(self, region1, region2=None, groupby=[], modify_2Dintervals_func=None, columns=(),
                        by_window=False, keep_table=False):
        """Host half of ``pileup_region`` (reference :1285-1358 down to the skip test :1105-1114).

        Returns None when the region has no feature, else a dict with the accepted windows:
        r0, c0 (global top-left bins, int64), kind (int8), flip (bool), group_codes (int64, -1 when
        ungrouped), group_keys (list of tuples in first-appearance order), transpose (bool).
        """
        if region2 is None:
            region2 = region1
        reg1, reg2 = self._region_tuple(region1), self._region_tuple(region2)
        if self._plain_pairs(modify_2Dintervals_func, by_window, keep_table, groupby):
            return self._pair_snippets(region1, region2, reg1, reg2, groupby, modify_2Dintervals_func)
        carry = columns
        # keep_table (callback path): rows must carry the reference's own column values (e.g. band tuples), so
        # even the built-in modify functions run in their DataFrame form
        builtin = (modify_2Dintervals_func is None or _is_builtin_modify(modify_2Dintervals_func)) and not keep_table
        src = {}                       # groupby column -> (carried column, decoder)
        if carry is not None and builtin:
            carry = list(carry)
            if by_window:
                carry += ["_uid1", "_uid2"]
            if self.store_stripes:
                carry += ["chrom1", "start1", "end1", "chrom2", "start2", "end2"]
            for g in groupby:
                src[g] = self._group_source(g)
                if g != "distance_band":
                    carry.append(src[g][0])
            if getattr(self, "ignore_group_order", False):
                for g in groupby:      # flipped snippets swap X1 <-> X2: the partner's codes are needed as well
                    name = src[g][0]
                    if name[-1:] in "12":
                        carry.append(name[:-1] + ("2" if name.endswith("1") else "1"))
        else:
            carry = None               # a user function may read any column
        tbl = self.CC.region_table(reg1, None if region2 == region1 else reg2, control=self.control, columns=carry)
        if tbl is None or len(tbl) == 0:
            return None
        decoders = {}
        if modify_2Dintervals_func is not None:
            if builtin:
                tbl, decoders = _apply_builtin_modify(tbl, modify_2Dintervals_func)
            else:
                tbl = _Cols.from_frame(modify_2Dintervals_func(tbl.frame()))
        lo1, hi1, off1 = self._global_extents[region1]
        lo2, hi2, off2 = self._global_extents[region2]
        W = 2 * self.pad_bins + 1
        # global bin ids fit 32 bits (the engine's own limit); narrow here so every later pass moves half the bytes
        hh = (tbl["endBin1"] - tbl["stBin1"]).astype(np.int32)
        ww = (tbl["endBin2"] - tbl["stBin2"]).astype(np.int32)
        if not getattr(self, "rescale", False) and not (np.all(hh == W) and np.all(ww == W)):
            raise ValueError("window size differs from 2*pad_bins+1")
        r0 = (tbl["stBin1"] + off1).astype(np.int32)
        c0 = (tbl["stBin2"] + off2).astype(np.int32)
        ok = (r0 >= lo1) & (r0 + hh <= hi1) & (c0 >= lo2) & (c0 + ww <= hi2)   # reference :1111-1114
        if not ok.all():
            tbl = tbl.take(ok)
            r0, c0, hh, ww = r0[ok], c0[ok], hh[ok], ww[ok]
        n = len(r0)
        flip = tbl["flip"].astype(bool) if "flip" in tbl else np.zeros(n, bool)
        coords = None
        if self.store_stripes:
            if by_window:
                raise NotImplementedError("store_stripes together with by-window pile-ups is not implemented")
            # str() of each coordinate, as the reference joins and re-splits them (coolpup.py:1170-1182, 1557-1560)
            coords = np.stack([np.asarray(tbl[c]).astype(str) for c in
                               ("chrom1", "start1", "end1", "chrom2", "start2", "end2")], axis=1)
        if groupby:
            keycols, decs = [], []
            for g in groupby:
                name, dec = src.get(g, (g, None))
                if g == "distance_band":
                    name, dec = g, decoders.get(g)
                v = tbl[name]
                if self.ignore_group_order and flip.any():
                    # flipped snippets swap every paired column X1<->X2 before grouping (reference :131-144)
                    partner = name[:-1] + ("2" if name.endswith("1") else "1") if name[-1:] in "12" else None
                    if partner is not None and partner in tbl:
                        v = np.where(flip, tbl[partner], v)
                keycols.append(v)
                decs.append(dec)
            codes, keys = _factorize_rows(keycols, decs)
        elif by_window:
            # every snippet is counted once for the feature on each side (group_by_region, lib/puputils.py:218-223):
            # emit it twice, side 1 then side 2, keyed by (chrom, start, end) of that side's feature
            # The key is the feature's number (CoordCreator.feature_ids) — the (chrom, start, end) tuple it stands for is written
            # out by pileupsByWindowWithControl: a genome's worth of features made 10^5 tuples per pile-up
            two = lambda a, b: np.stack([a, b], axis=1).ravel()      # noqa: E731  interleave side 1 / side 2
            codes, uniq = pd.factorize(two(tbl["_uid1"], tbl["_uid2"]))
            codes, keys = codes.astype(np.int64), uniq.tolist()
            dup = np.repeat(np.arange(n), 2)
            return {"r0": r0[dup], "c0": c0[dup], "kind": tbl["kind"].astype(np.int8)[dup], "flip": flip[dup],
                    "group_codes": codes, "group_keys": keys, "n": 2 * n, "h": hh[dup], "w": ww[dup], "coords": None}
        else:
            codes, keys = np.full(n, -1, np.int64), []
        out = {"r0": r0, "c0": c0, "kind": tbl["kind"].astype(np.int8), "flip": flip, "group_codes": codes,
               "group_keys": keys, "n": n, "coords": coords, "h": hh, "w": ww}
        if keep_table:
            out["table"] = tbl                 # the accepted rows with every carried column (callback path)
        return out

    def _plain_pairs(self, modify, by_window, keep_table, groupby=()):
        """True when the windows of a region are plain feature pairs: bedpe features, no rescaling, stripes, flips,
        by-window grouping or user functions — the case _pair_snippets builds with one fused pass of the library."""
        if self.CC.kind != "bedpe" or by_window or keep_table or self.store_stripes or getattr(self, "rescale", False):
            return False
        if self.flip_negative_strand or getattr(self, "ignore_group_order", False):
            return False
        if self.CC._tbl.n == 0 or self.CC.pos_stream == self.CC.empty_stream:
            return False
        is_banding = modify is bin_distance_intervals or \
            (isinstance(modify, partial) and modify.func is bin_distance_intervals)
        if "distance_band" in (groupby or ()) and not is_banding:
            return False            # a distance_band column of the caller's own: the general path groups by the column itself
        return modify is None or is_banding

    def _pair_snippets(self, region1, region2, reg1, reg2, groupby, modify):
        """region_snippets for plain feature pairs (see _plain_pairs).  Same result as the general path — the ROI windows
        followed by nshifts shifted copies (reference :387-453, same RNG calls in the same order), bounds test (:1105-1114),
        group keys in order of first appearance — but group codes are worked out on the ROI rows only (a control copy
        belongs to the group of its ROI window: distance, strands and every other feature column are copied, :405-419) and
        shifting, bounds test and compaction are one pass of the library (pup_host_windows) into page-locked arrays."""
        from . import engine as _engine
        CC = self.CC
        rows = CC._rows_trans_pairs(tuple(reg1), tuple(reg2)) if CC.trans else CC._rows_pairs_region(tuple(reg1))
        n = _nrows(rows)
        if n == 0:
            return None
        st1, st2 = CC._col("stBin1")[rows], CC._col("stBin2")[rows]
        W = 2 * self.pad_bins + 1
        if not (np.all(CC._col("endBin1")[rows] - st1 == W) and np.all(CC._col("endBin2")[rows] - st2 == W)):
            raise ValueError("window size differs from 2*pad_bins+1")
        codes, keys = None, []
        if groupby:
            keycols, decs = [], []
            for g in groupby:
                if g == "distance_band":
                    kw = modify.keywords if isinstance(modify, partial) else {}
                    edges = kw.get("band_edges", "default")
                    if isinstance(edges, str) and edges == "default":
                        edges = _default_band_edges()
                    keycols.append(_engine.count_le(edges, CC._col("distance")[rows]))
                    decs.append(lambda i, e=edges: tuple(e[i - 1:i + 1]))
                    continue
                name, dec = self._group_source(g)
                keycols.append(CC.group_codes(name[4:])[0][rows] if name.startswith("_gc_") else CC._col(name)[rows])
                decs.append(dec)
            codes, keys = _factorize_rows(keycols, decs)
        nshifts = self.nshifts if self.control else 0
        shift = sign = None
        if nshifts > 0:
            shift, sign = CC._draw_raw(n * nshifts)
        lo1, hi1, off1 = self._global_extents[region1]
        lo2, hi2, off2 = self._global_extents[region2]
        r0, c0, code_out, n_roi = _engine.host_windows(st1, st2, codes, shift, sign, nshifts, self.resolution, off1, off2,
                                                       lo1, hi1, lo2, hi2, W, W, arena=getattr(self, "_arena_ok", False))
        m = len(r0)
        kind = np.empty(m, np.int8)
        kind[:n_roi] = KIND_ROI
        kind[n_roi:] = KIND_CONTROL
        size = np.broadcast_to(np.int32(W), (m,))
        return {"r0": r0, "c0": c0, "kind": kind, "flip": None, "n": m, "n_roi": n_roi, "coords": None, "h": size, "w": size,
                # (ungrouped: a zero-cost read-only view — np.full of 1.1e7 codes per region set cost as much as the window pass)
                "group_codes": code_out if code_out is not None else np.broadcast_to(np.int64(-1), (m,)), "group_keys": keys}

    # -- the pile-up -------------------------------------------------------------------------------------------
    def _flip_column(self, groupby):
        """The paired annotation whose order decides which snippets are flipped: "strand" (flip_negative_strand), the
        stem X of a groupby pair X1 / X2 (ignore_group_order), or None.  Same accepted inputs, errors and warnings as
        the reference's checks (coolpup.py:1431-1475)."""
        igo = self.ignore_group_order
        if not self.flip_negative_strand and not igo:
            return None
        if igo and (self.flip_negative_strand or groupby):
            for bad, what in ((self.local, "local pileups"), (self.kind == "bedpe", "bedpe files")):
                if bad:
                    raise ValueError(f"ignore_group_order doesn't make sense for {what}")
        if self.flip_negative_strand:
            if igo and groupby:
                warnings.warn("flip_negative_strand and ignore_group_order leads to combining strands, not other groups")
            return "strand"
        if not groupby:
            warnings.warn("Need to specify groupby for ignore_group_order")
            return None
        # groupby columns that come as a pair stem+"1" / stem+"2"
        paired_cols = {g for g in groupby if g[:-1] + "1" in groupby and g[:-1] + "2" in groupby}
        if igo is True:
            stems = {g[:-1] for g in paired_cols}
        elif isinstance(igo, str):
            stems = {igo}
        elif len(igo) == 1:
            stems = set(igo)
        else:
            stems = {g[:-1] for g in igo}
        if len(stems) == 1 and next(iter(stems)) + "1" in paired_cols:
            return next(iter(stems))
        raise ValueError("Ambiguous ignore_group_order, please provide str or list of two strings which are in groupby")

    def pileupsWithControl(self, nproc=None, groupby=[], ignore_group_order=False, modify_2Dintervals_func=None,
                           postprocess_func=None, extra_sum_funcs=None, _columns=(), _by_window=False):
        """All regions -> normalised pile-ups DataFrame (reference :1360-1654)."""
        self.ignore_group_order = ignore_group_order
        callbacks = postprocess_func is not None or bool(extra_sum_funcs)
        if callbacks and _by_window:
            raise NotImplementedError("by-window pile-ups already group per feature on the GPU; no callbacks there")
        if nproc is None:
            nproc = self.nproc
        if len(self.chroms) == 0:
            return self.make_outmap(), 0

        columns = list(_columns) if _columns is not None else None
        flipby = self._flip_column(groupby)

        modify = modify_2Dintervals_func
        if self.flip_negative_strand or (self.ignore_group_order and groupby):
            modify = partial(flip_mark_intervals_func, flipby=flipby,
                             flip_negative_strand=self.flip_negative_strand, extra_func=modify_2Dintervals_func)
            if columns is not None:
                columns += ["strand1"] if self.flip_negative_strand else [f"{flipby}1", f"{flipby}2"]
            if callbacks:
                postprocess_func = partial(flip_snip_func, groupby=groupby, ignore_group_order=self.ignore_group_order,
                                           extra_func=postprocess_func)
        if callbacks:
            # per-snippet Python callbacks: the GPU produces the windows, the host runs the callbacks and the
            # reference's per-snippet accumulation on them (coolpup.py:1236-1283)
            pileups = [self._callback_region(r1, r2, groupby, modify, postprocess_func, extra_sum_funcs)
                       for r1, r2 in self._region_pairs()]
            want_control = bool(self.control) or (bool(self.expected) and not self.ooe)
            return finalize_callback_pileups(self, pileups, groupby, want_control, extra_sum_funcs)
        # multi-GPU: region (pairs) are dealt to the ranks (longest first, identical on every rank); a rank builds the
        # windows of its own regions only and steps the control RNG past the others
        from . import dist as _dist
        pairs = self._region_pairs()
        rank, world = _dist.world()
        owned = None
        if world > 1:
            weights = [self.CC.region_weight(self._region_tuple(r1), self._region_tuple(r2)) for r1, r2 in pairs]
            owned = _dist.shard(len(pairs), weights, rank, world)
            # rows this rank's engine needs: those of the earlier region of every pair it owns (upper-triangular table)
            ext = self._global_extents
            self._owned_rows = _merge_ranges([min(ext[r1][:2], ext[r2][:2]) for i, (r1, r2) in enumerate(pairs) if i in owned])
        grouped = bool(groupby) or _by_window
        if getattr(self, "_window_source", None) is None:
            _prefetch_engine(self._aclr, _dist.local_device(), rows=self._owned_rows if world > 1 else None)
        batches = []
        ahead = None
        nsh = self.nshifts if self.control else 0
        early, self.CC._early_ahead = getattr(self.CC, "_early_ahead", None), None      # (draws started while the table was sorted)
        if owned is None and nsh > 0 and len(pairs) > 1 and not os.environ.get("COOLPUPPY_AMD_NO_DRAW_AHEAD") and \
                self._plain_pairs(modify, _by_window, False, groupby):
            # every region takes _pair_snippets: the helper thread draws the regions' control shifts ahead (see _DrawAhead)
            CC = self.CC
            sizes = []
            for region1, region2 in pairs:
                reg1, reg2 = self._region_tuple(region1), self._region_tuple(region2)
                rows = CC._rows_trans_pairs(tuple(reg1), tuple(reg2)) if CC.trans else CC._rows_pairs_region(tuple(reg1))
                if _nrows(rows):
                    sizes.append(_nrows(rows) * nsh)
            if early is not None and early[1] == sizes:
                ahead, early = early[0], None                 # the very sequence this pile-up needs: already under way
                CC._draw_ahead = ahead
            elif sum(sizes) >= _DRAW_AHEAD_MIN:
                if early is not None:
                    early[0].cancel(early[2]); early = None
                ahead = CC._draw_ahead = _DrawAhead(CC, sizes)
        if early is not None:                                 # started for a pile-up that is not this one: stop it, generator back
            early[0].cancel(early[2])
        # (round 6: also without controls — a loop list piled up as it is, nshifts = 0: the per-region window tables and the pass that
        # gathers them were half of such a call's host time at twenty regions)
        fused = (owned is None and (nsh > 0 or not self.control) and not grouped and not self.expected and not self.trans and not self.rescale
                 and self._plain_pairs(modify, _by_window, False, groupby) and not os.environ.get("COOLPUPPY_AMD_NO_FUSED_WINDOWS"))
        plan = None
        try:
            if fused:
                plan = self._fused_plan(pairs)
            else:
                batches = self._region_batches(pairs, owned, groupby, modify, columns, _by_window)
        finally:
            if ahead is not None:
                self.CC._draw_ahead = None
                ahead.close()
        if plan is not None:
            return self.finalize_plan(plan, self.run_plan(plan))
        region_groups = None
        if owned is not None:
            # the global group table needs every region's group keys in region order: the ranks swap them (a few keys each)
            got = _dist.merge_dicts({i: self.region_groups(batches[i][2], grouped) for i in owned})
            region_groups = [got[i] for i in range(len(pairs))]
        return self._pile_and_finalize(batches, groupby, grouped=grouped, region_groups=region_groups, any_order=_by_window)

    def _fused_plan(self, pairs):
        """The plan of an UNGROUPED pile-up of plain feature pairs with random-shift controls (the headline shape: BEDPE features,
        nshifts > 0, no expected), written in one go: every region's ROI windows first — they need no draw —, then region after
        region, as the reference's draws arrive (coolpup.py:420-436, same calls in the same order), its shifted copies, each pass
        of the library (pup_host_windows / pup_host_control_windows: shift, bounds test :1105-1114, compaction) writing straight
        behind the previous one in the page-locked arrays the engine call reads.  Same windows in the same (tile, region, stream)
        order as region_snippets -> make_plan -> group_tiles produce, without the per-region arrays and the pass that gathers them
        (88 MB written twice and page-faulted once per 10^7 windows)."""
        from . import engine as _engine
        from .engine import MODE_COV
        CC = self.CC
        W = 2 * self.pad_bins + 1
        nsh = int(self.nshifts) if self.control else 0               # (0: no controls at all — ROI windows only)
        want_control = nsh > 0
        regs = []
        for bi, (region1, region2) in enumerate(pairs):
            rows = CC._rows_pairs_region(tuple(self._region_tuple(region1)))
            regs.append((bi, region1, rows, _nrows(rows)))
        total = sum(r[3] for r in regs)
        st1c, st2c = CC._col("stBin1"), CC._col("stBin2")
        if total and not (np.all(CC._col("endBin1") - st1c == W) and np.all(CC._col("endBin2") - st2c == W)):
            raise ValueError("window size differs from 2*pad_bins+1")
        r0, c0 = _engine.pinned_empty(total * (1 + nsh)), _engine.pinned_empty(total * (1 + nsh))
        res = self.resolution
        pos = 0
        spans = {}                               # region -> [roi start, roi end, control start, control end] in r0 / c0
        for bi, region1, rows, n in regs:
            if n == 0:
                continue
            lo, hi, off = self._global_extents[region1]
            k = _engine.host_windows_into(r0, c0, pos, st1c[rows], st2c[rows], None, None, 0, res, off, off, lo, hi, lo, hi, W, W)
            spans[bi] = [pos, pos + k, 0, 0]
            pos += k
        roi_total = pos
        for bi, region1, rows, n in regs:
            if n == 0 or not want_control:
                if n and not want_control:
                    logger.info(f"{region1, region1}: {spans[bi][1] - spans[bi][0]}")
                continue
            shift, sign = CC._draw_raw(n * nsh)
            lo, hi, off = self._global_extents[region1]
            k = _engine.host_windows_into(r0, c0, pos, st1c[rows], st2c[rows], shift, sign, nsh, res, off, off, lo, hi, lo, hi, W, W,
                                          controls_only=True)
            spans[bi][2:] = [pos, pos + k]
            pos += k
            logger.info(f"{region1, region1}: {spans[bi][1] - spans[bi][0]}")
        G, T = 1, 2
        igd = int(self.ignore_diags)
        mode = MODE_COV if self.coverage_norm else 0
        live = [bi for bi in spans if spans[bi][1] - spans[bi][0] + spans[bi][3] - spans[bi][2] > 0]
        calls = []
        if live:
            head = pairs[live[0]][0]
            calls.append(_Call({"region1": head, "region2": head, "expected": None, "r0": r0[:pos], "c0": c0[:pos], "flip": None,
                                "flip_from": None, "tile_ptr": np.array([0, roi_total, pos], np.int64), "ignore_diags": igd, "mode": mode}))

        def region_items():
            from .engine import RunTile
            size = lambda m: np.broadcast_to(np.int32(W), (m,))      # noqa: E731
            items = []
            for bi in live:
                a, b, c, d = spans[bi]
                m = (b - a) + (d - c)
                items.append((pairs[bi][0], pairs[bi][0], None, np.concatenate([r0[a:b], r0[c:d]]), np.concatenate([c0[a:b], c0[c:d]]),
                              None, RunTile(b - a, m, 0, G), igd, mode, size(m), size(m), bi))
            return items

        plan = _Plan({"T": T, "G": G, "gid": {"all": 0}, "order": {KIND_ROI: ["all"], KIND_CONTROL: ["all"] if want_control else []},
                      "want_control": want_control,
                      "groupby": [], "grouped": False, "calls": calls, "pad": self.pad_bins, "rescale": False,
                      "n_regions": len(pairs),
                      "region_groups": [({KIND_ROI: [], KIND_CONTROL: []} if bi in live else None) for bi in range(len(pairs))],
                      "store_stripes": False, "expected_table": None, "stripe_jobs": [],
                      "weight_name": self.clr_weight_name if self.clr_weight_name else None,
                      "cov_name": self.coverage_norm if self.coverage_norm else None})
        plan.lazy["region_items"] = region_items
        return plan

    def _region_batches(self, pairs, owned, groupby, modify, columns, _by_window):
        from . import engine as _engine
        _engine._ARENA.reset()             # (the regions' window arrays of THIS pile-up: scratch kept between pile-ups)
        self._arena_ok = True
        try:
            return self._region_batches_impl(pairs, owned, groupby, modify, columns, _by_window)
        finally:
            self._arena_ok = False

    def _region_batches_impl(self, pairs, owned, groupby, modify, columns, _by_window):
        batches = []
        for i, (region1, region2) in enumerate(pairs):
            if owned is not None and i not in owned:
                self.CC.skip_region(self._region_tuple(region1), None if region2 == region1 else self._region_tuple(region2),
                                    control=self.control)
                batches.append((region1, region2, None))
                continue
            b = self.region_snippets(region1, region2, groupby=groupby, modify_2Dintervals_func=modify,
                                     columns=columns, by_window=_by_window)
            batches.append((region1, region2, b))
            if b is not None and b["n"] > 0:
                logger.info(f"{region1, region2}: {int((b['kind'] == KIND_ROI).sum())}")
        return batches

    def region_groups(self, b, grouped):
        """Group keys of one region's windows, per kind, in order of first appearance (without "all"); None for a
        region that yields no window.  This is all the global group table needs to know about a region — under
        multi-GPU sharding it is what the ranks exchange about the regions they own."""
        if b is None or b["n"] == 0:
            return None
        exp_as_control = bool(self.expected) and not self.ooe
        want_control = bool(self.control) or exp_as_control
        out = {KIND_ROI: [], KIND_CONTROL: []}
        if grouped:
            for kind in (KIND_ROI, KIND_CONTROL):
                if kind == KIND_CONTROL and not want_control:
                    continue
                # with expected & !ooe every ROI snippet also emits an expected ("control") snippet
                src = KIND_ROI if (exp_as_control and kind == KIND_CONTROL) else kind
                if "n_roi" in b:       # ROI windows first, then the controls
                    codes = b["group_codes"][:b["n_roi"]] if src == KIND_ROI else b["group_codes"][b["n_roi"]:]
                else:
                    codes = b["group_codes"][b["kind"] == src]
                out[kind] = [b["group_keys"][c] for c in _first_codes(codes, len(b["group_keys"]))]
        return out

    def make_plan(self, batches, groupby, grouped=None, region_groups=None):
        """Turn per-region window tables into a declarative list of engine calls plus the group bookkeeping
        the finaliser needs.  Pure host code (no GPU): tests replay a plan on the CPU oracle.

        region_groups: per batch, what region_groups() returns for it — given when some batches are held by other
        ranks (their entry in `batches` is then (region1, region2, None)); computed here otherwise."""
        from .engine import MODE_COV, MODE_EXPECTED, MODE_OOE, MODE_TRANSPOSE
        MODE_LOCAL = 0x20
        rescale = bool(getattr(self, "rescale", False))
        if grouped is None:
            grouped = bool(groupby)
        # global group table in the reference's first-appearance order: regions in order, each region's
        # groups in snippet order, then "all" (coolpup.py:1263-1283, 1511-1531)
        exp_as_control = bool(self.expected) and not self.ooe
        want_control = bool(self.control) or exp_as_control
        if region_groups is None:
            region_groups = [self.region_groups(b, grouped) for _, _, b in batches]
        # (per kind: the regions' keys in region order, "all" after every region's own — first appearance wins; dict.fromkeys over a
        # chain runs at C speed: a by-window pile-up brings a key per feature)
        order = {}
        for kind, with_all in ((KIND_ROI, True), (KIND_CONTROL, want_control)):
            tail = ("all",) if with_all else ()
            order[kind] = list(dict.fromkeys(itertools.chain.from_iterable(
                itertools.chain(rg[kind] if rg is not None else (), tail) for rg in region_groups)))
        keys_all = list(dict.fromkeys(order[KIND_ROI] + order[KIND_CONTROL]))
        # Tile numbers are the engine's business (results are looked up through `gid`, the output follows `order`): the
        # staged kernel piles up FOUR consecutive groups from one staging of the matrix (pup_staged.hpp, sets of tile
        # pairs), so groups whose windows lie at the same distance from the diagonal get neighbouring numbers
        gb = [g for g in (groupby or [])]
        if "distance_band" in gb and len(keys_all) > 4:
            di = gb.index("distance_band")

            def band_rank(k):
                comp = k[di] if isinstance(k, tuple) and len(k) > di else None
                return (0, tuple(comp)) if isinstance(comp, tuple) and len(comp) else (1, ())

            keys_all = sorted(keys_all, key=band_rank)        # (stable: first-appearance order inside a band)
        gid = {k: i for i, k in enumerate(keys_all)}
        G = len(keys_all)
        # (tile = kind * G + group; a pile-up of many groups without a control — by-window: a tile per feature — does not pay for a
        # second, empty half of accumulators: their reduction, their 300 MB on the way back and through the finaliser)
        T = G if (G > 64 and not want_control) else 2 * G
        # expected of every view region in ONE device table, so that a single engine call can span regions
        exp_table = None
        if self.expected:
            names = sorted(self._global_extents, key=lambda k: self._global_extents[k][0])
            names = [k for k in names if k in self.view_df.index]
            starts = np.array([self._global_extents[k][0] for k in names], np.int64)
            ends = np.array([self._global_extents[k][1] for k in names], np.int64)
            if len(names) and np.all(starts[1:] >= ends[:-1]):          # disjoint regions (overlap: per-region calls)
                if self.trans:
                    pos = {k: i for i, k in enumerate(names)}
                    pair = np.full((len(names), len(names)), np.nan)
                    for r1, r2 in self._region_pairs():
                        v = self.get_expected_trans(r1, r2)
                        pair[pos[r1], pos[r2]] = pair[pos[r2], pos[r1]] = v
                    exp_table = {"names": names, "start": starts, "end": ends, "pair": pair, "vectors": None}
                else:
                    exp_table = {"names": names, "start": starts, "end": ends, "pair": None,
                                 "vectors": [self._expected_vectors[k] for k in names]}
        raw = []
        for bi, (region1, region2, b) in enumerate(batches):
            if b is None or b["n"] == 0:
                continue
            run_tile = (not grouped) and ("n_roi" in b) and not exp_as_control and b.get("flip") is None and not rescale
            tile_done = False
            if grouped:     # (a key none of the region's kept windows uses is not in the table: its code never occurs)
                lut = np.array([gid.get(k, -1) for k in b["group_keys"]], np.int32)
                if "n_roi" in b and not exp_as_control and len(lut):
                    # ROI windows first, then the controls: tile = group, + G from there on — two gathers straight into the tile
                    # array (a gather and an in-place add over the same 10^7 entries were 23 ms of a 1e6-pair call)
                    from .engine import lut_codes
                    g = lut_codes(lut, b["group_codes"], add_from=b["n_roi"], add=G)
                    tile_done = True
                else:
                    g = lut[b["group_codes"]]
            elif run_tile:
                g = None    # (no per-window array at all: tile 0 for the ROI windows, G for the controls — engine.RunTile)
            else:
                g = np.zeros(b["n"], np.int32)
            expected = None
            if self.expected:
                if exp_table is not None:
                    expected = "table"
                else:
                    expected = np.array([self.get_expected_trans(region1, region2)], np.float64) if self.trans \
                        else self._expected_vectors[region1]
            igd = -1 if self.trans else int(self.ignore_diags)
            # engine rows must come from the earlier region of the upper-triangular table
            transpose = self._global_extents[region1][0] > self._global_extents[region2][0]
            r0, c0 = (b["c0"], b["r0"]) if transpose else (b["r0"], b["c0"])
            hh, ww = (b["w"], b["h"]) if transpose else (b["h"], b["w"])
            tr = MODE_TRANSPOSE if transpose else 0
            loc = MODE_LOCAL if (rescale and self.local) else 0
            mode = (MODE_OOE if (self.expected and self.ooe) else 0) | (MODE_COV if self.coverage_norm else 0) | tr | loc
            if run_tile:
                from .engine import RunTile
                tile = RunTile(b["n_roi"], b["n"], 0, G)
            elif tile_done:
                tile = g
            elif "n_roi" in b:         # ROI windows first, then the controls: tile = group, + G from there on
                tile = g if g.flags.writeable and g.base is None else g.copy()
                tile[b["n_roi"]:] += np.int32(G)
            else:
                tile = b["kind"].astype(np.int32) * np.int32(G) + g
            raw.append((region1, region2, expected, r0, c0, b["flip"], tile, igd, mode, hh, ww, bi))
            if exp_as_control:
                roi = b["kind"] == KIND_ROI
                raw.append((region1, region2, expected, r0[roi], c0[roi], None if b["flip"] is None else b["flip"][roi],
                            G + g[roi], igd,
                            MODE_EXPECTED | tr | loc, hh[roi], ww[roi], bi))
        # regions that need no per-region state (no expected vector) and share mode / diagonal mask go to the
        # engine as ONE call: fewer launches, and the engine's interleaved groups span region boundaries
        stripe_jobs = []
        if self.store_stripes:
            for bi, (region1, region2, b) in enumerate(batches):
                if b is None or b["n"] == 0:
                    continue
                roi = np.flatnonzero(b["kind"] == KIND_ROI)
                if len(roi) == 0:
                    continue
                gk = np.array([gid[k] for k in b["group_keys"]], np.int64)[b["group_codes"][roi]] if grouped \
                    else np.full(len(roi), gid["all"], np.int64)
                transpose = self._global_extents[region1][0] > self._global_extents[region2][0]
                r0, c0 = (b["c0"][roi], b["r0"][roi]) if transpose else (b["r0"][roi], b["c0"][roi])
                expected = None
                if self.expected and self.ooe:
                    expected = "table" if exp_table is not None else (
                        np.array([self.get_expected_trans(region1, region2)], np.float64) if self.trans
                        else self._expected_vectors[region1])
                job = {"expected": expected, "ignore_diags": -1 if self.trans else int(self.ignore_diags),
                       "mode": (MODE_OOE if (self.expected and self.ooe) else 0) | (MODE_TRANSPOSE if transpose else 0)
                       | (MODE_LOCAL if (rescale and self.local) else 0),
                       "r0": r0.astype(np.int32), "c0": c0.astype(np.int32), "gid": gk, "coords": b["coords"][roi],
                       "region": bi}
                if rescale:   # stripes of the ZOOMED window (reference :1159-1169): whole windows via pup_extract
                    hh, ww = (b["w"][roi], b["h"][roi]) if transpose else (b["h"][roi], b["w"][roi])
                    job["h"], job["w"] = hh.astype(np.int32), ww.astype(np.int32)
                stripe_jobs.append(job)
        merged = []                     # [head item, [parts of fields 3..6]]
        for item in raw:
            prev = merged[-1][0] if merged else None
            same_exp = (item[2] is None and prev is not None and prev[2] is None) or \
                (isinstance(item[2], str) and prev is not None and isinstance(prev[2], str))
            fields = (item[3], item[4], item[5], item[6], item[9], item[10])
            if prev is not None and same_exp and item[7] == prev[7] and item[8] == prev[8]:
                merged[-1][1].append(fields)
            else:
                merged.append([item, [fields]])
        calls = [_engine_call_parts(head[0], head[1], head[2], parts, T, head[7], head[8], rescale)
                 for head, parts in merged]
        return {"T": T, "G": G, "gid": gid, "order": order, "want_control": want_control,
                "groupby": list(groupby), "grouped": bool(grouped), "calls": calls,
                "pad": (self.rescale_size - 1) // 2 if rescale else self.pad_bins, "rescale": rescale,
                "n_regions": len(batches), "region_groups": region_groups, "region_items": raw,
                "store_stripes": bool(self.store_stripes),
                "expected_table": exp_table, "stripe_jobs": stripe_jobs,
                "weight_name": self.clr_weight_name if self.clr_weight_name else None,
                "cov_name": self.coverage_norm if self.coverage_norm else None}

    def run_plan(self, plan, calls=None, reduce=True):
        """Execute a plan on this process's GPU (its calls hold the regions this rank owns; with several ranks one
        all-reduce of the packed accumulators follows) and fetch the tiles.  calls / reduce=False: run just these
        calls of the plan and return this process's own tiles (per-region tiles for the inf merge rule)."""
        from . import dist as _dist
        rank, world = _dist.world()
        eng = _engine_for(self._aclr, _dist.local_device(), rows=getattr(self, "_owned_rows", None) if world > 1 else None)
        bins = self._aclr.bins()
        eng.load_bins(bins[plan["weight_name"]][:].values if plan["weight_name"] else None,
                      bins[plan["cov_name"]][:].values if plan["cov_name"] else None)
        eng.reset(plan["T"], plan["pad"])
        et = plan.get("expected_table")
        table_set = False
        for c in (plan["calls"] if calls is None else calls):
            if len(c["r0"]) == 0:
                continue
            if isinstance(c["expected"], str):
                if not table_set:
                    eng.set_expected_table(et["start"], et["end"], vectors=et["vectors"], pair=et["pair"])
                    table_set = True
            else:
                eng.set_expected(c["expected"])
                table_set = False
            if plan.get("rescale"):
                eng.accumulate_rescaled(c["r0"], c["c0"], c["h"], c["w"], c["tile_ptr"], flip_from=c["flip_from"],
                                        ignore_diags=c["ignore_diags"], mode=c["mode"])
            else:
                eng.accumulate(c["r0"], c["c0"], c["tile_ptr"], flip_from=c["flip_from"],
                               ignore_diags=c["ignore_diags"], mode=c["mode"])
        if not reduce:
            return eng.fetch()
        if world > 1:
            _dist.check_same_plan(plan)
        if world > 1 and plan["grouped"] and plan["T"] >= _dist.sparse_exchange_min_tiles():
            # many groups (by-window: a tile per feature), each piled up where its region lives: the ranks swap the tiles they
            # hold instead of all-reducing every accumulator ("all" is folded from the groups on the host, finalize_plan)
            plan["exchange_bytes"] = _dist.exchange_tiles(eng, plan_tiles_with_windows(plan))
        else:
            _dist.allreduce_engine(eng)
        acc = eng.fetch()
        if plan.get("stripe_jobs") or (plan.get("store_stripes") and world > 1):
            # O(n*W) per-snippet output, not a reduction: a rank extracts the stripes of its own regions and the ranks
            # swap them, so that every rank ends with the complete result, like the reduced tiles
            acc["stripes"] = []
            for job in plan["stripe_jobs"]:
                if isinstance(job["expected"], str):
                    eng.set_expected_table(et["start"], et["end"], vectors=et["vectors"], pair=et["pair"])
                else:
                    eng.set_expected(job["expected"])
                if "h" in job:
                    win = eng.extract(job["r0"], job["c0"], plan["pad"], height=job["h"], width=job["w"],
                                      ignore_diags=job["ignore_diags"], mode=job["mode"])
                    cntr = plan["pad"]
                    acc["stripes"].append((np.ascontiguousarray(win[:, cntr, :]),
                                           np.ascontiguousarray(win[:, ::-1, cntr])))
                    continue
                acc["stripes"].append(eng.stripes(job["r0"], job["c0"], plan["pad"], ignore_diags=job["ignore_diags"],
                                                  mode=job["mode"]))
            _gather_stripes(plan, acc)
        return acc

    def finalize_plan(self, plan, acc):
        """Fetched tiles -> the reference's output DataFrame."""
        G, gid, order = plan["G"], plan["gid"], plan["order"]
        if plan["grouped"]:   # "all" of a grouped pile-up = sum of its groups (reference :1271-1282)
            for kind in (KIND_ROI, KIND_CONTROL):
                members = np.array([kind * G + gid[k] for k in order[kind] if not (isinstance(k, str) and k == "all")], np.int64)
                if len(members):
                    a = kind * G + gid["all"]
                    # groups numbered in output order (the usual case) and the whole kind present: the "all" tile itself still
                    # holds zeros, so the members' sum in their order IS the sum over the kind's slice — no gather of every tile
                    # (a by-window pile-up has one per feature)
                    # (the control groups of a by-window pile-up with random shifts come in another order than the ROI ones — a permutation
                    # of the same slice: summed in tile order as well; gathering 37 k tiles five times was 0.3 s of such a call)
                    # (... and, by-window, a kind may lack a few of the features altogether — their tiles of that kind hold zeros)
                    whole = not acc["n"][a] and not acc["sum"][a].any() and \
                        (plan.get("any_order", False) or (len(members) == G - 1 and bool(np.all(np.diff(members) > 0))))
                    for name in ("sum", "num", "n", "cov_start", "cov_end"):
                        acc[name][a] = acc[name][kind * G:(kind + 1) * G].sum(axis=0) if whole else acc[name][members].sum(axis=0)
        self._merge_inf_cells(plan, acc)
        stripes = _collect_stripes(plan, acc) if (plan.get("stripe_jobs") or acc.get("stripe_jobs")) else None
        return finalize_pileups(self, acc, order, gid, G, plan["groupby"], plan["want_control"],
                                grouped=plan["grouped"], stripes=stripes, any_order=plan.get("any_order", False))

    def _merge_inf_cells(self, plan, acc):
        """Cells that hold +inf (a pixel over expected == 0) follow the reference's merge arithmetic instead of plain
        addition: every merge of two pile-ups passes both through nan_to_num first (sum_pups, lib/puputils.py:97-98), so
        inf becomes the largest double, two of those add up to inf again, and the value of the cell depends on WHICH
        regions — and, in a grouped pile-up, which groups of a region (their fold into "all" also rewrites the groups'
        own tiles, coolpup.py:1271-1282) — held an inf, in order.  Replayed here on per-region tiles, which the engine
        piles up again region by region; only cells touched by an inf are replaced, everything else keeps the
        one-pass sums.  Rare (needs expected == 0 under a non-zero pixel), so the second pass is off the fast path."""
        S = acc["sum"]
        G, gid, T = plan["G"], plan["gid"], plan["T"]
        # every window is counted under "all" (the fold of the groups has happened): an inf anywhere shows there
        alls = [t for t in (gid["all"], G + gid["all"]) if t < len(S)]
        if np.isfinite(S[alls]).all() or not np.isinf(S).any():
            return
        from . import dist as _dist
        nreg = plan["n_regions"]
        items = {}
        for it in plan["region_items"]:
            items.setdefault(it[11], []).append(it)
        if nreg == 1:
            per = {bi: S.copy() for bi in items}
        else:
            per = {}
            for bi, its in items.items():
                calls = [_engine_call_parts(it[0], it[1], it[2], [(it[3], it[4], it[5], it[6], it[9], it[10])], T, it[7],
                                            it[8], plan["rescale"]) for it in its]
                per[bi] = self.run_plan(plan, calls=calls, reduce=False)["sum"]
            per = _dist.merge_dicts(per)                       # regions piled up by other ranks
        nn = np.nan_to_num
        zero = np.zeros_like(S[0])
        for kind in (KIND_ROI, KIND_CONTROL):
            if kind == KIND_CONTROL and not plan["want_control"]:
                continue
            state = {}
            for bi in range(nreg):
                D, rg = per.get(bi), plan["region_groups"][bi]
                keys = rg[kind] if rg is not None else []
                tiles = {}
                if plan["grouped"]:
                    run = zero
                    for k in keys:
                        tiles[k] = nn(D[kind * G + gid[k]])
                        run = nn(run) + tiles[k]
                    tiles["all"] = run
                else:
                    tiles["all"] = D[kind * G + gid["all"]] if D is not None else zero
                for k, d in tiles.items():
                    state[k] = d if k not in state else nn(state[k]) + nn(d)
            for k, d in state.items():
                t = kind * G + gid[k]
                fix = np.isinf(S[t]) | ~np.isfinite(d)
                S[t][fix] = d[fix]

    def _pile_and_finalize(self, batches, groupby, grouped=None, region_groups=None, any_order=False):
        plan = self.make_plan(batches, groupby, grouped=grouped, region_groups=region_groups)
        plan["any_order"] = bool(any_order)        # (by-window: the caller sorts the rows itself — _by_window_frame)
        return self.finalize_plan(plan, self.run_plan(plan))

    def pileup_region(self, region1, region2=None, groupby=[], modify_2Dintervals_func=None, postprocess_func=None,
                      extra_sum_funcs=None):
        """One region (pair) -> {"ROI": {group: pup}, "control": {group: pup}} with summed, un-normalised
        tiles (reference :1285-1358): pup = data (sum), num, n, cov_start, cov_end."""
        if region2 is None:
            region2 = region1
        if not hasattr(self, "ignore_group_order"):
            self.ignore_group_order = False
        if postprocess_func is not None or extra_sum_funcs:
            return self._callback_region(region1, region2, groupby, modify_2Dintervals_func, postprocess_func,
                                         extra_sum_funcs)
        b = self.region_snippets(region1, region2, groupby=groupby, modify_2Dintervals_func=modify_2Dintervals_func,
                                 columns=None)
        self._owned_rows = None
        plan = self.make_plan([(region1, region2, b)], groupby)
        acc = self.run_plan(plan)
        return _tiles_to_pups(plan, acc)

    # -- per-snippet Python callbacks -------------------------------------------------------------------------------
    _CALLBACK_BATCH = 16384            # windows fetched from the GPU per pup_extract call

    def _windows(self, b, region1, region2, sel, mode_extra=0):
        """Windows of the snippets b[sel] of one region pair, from the GPU (pup_extract): (data, cov_start, cov_end).
        ``self._window_source`` (tests: an oracle stand-in with the same signature) replaces the engine."""
        from .engine import MODE_COV, MODE_OOE, MODE_TRANSPOSE
        transpose = self._global_extents[region1][0] > self._global_extents[region2][0]
        r0, c0 = (b["c0"][sel], b["r0"][sel]) if transpose else (b["r0"][sel], b["c0"][sel])
        hh, ww = (b["w"][sel], b["h"][sel]) if transpose else (b["h"][sel], b["w"][sel])
        rescale = bool(self.rescale)
        mode = mode_extra | (MODE_TRANSPOSE if transpose else 0) | (0x20 if (rescale and self.local) else 0)
        if not (mode_extra & 0x02):
            mode |= (MODE_OOE if (self.expected and self.ooe) else 0) | (MODE_COV if self.coverage_norm else 0)
        expected = None
        if self.expected:
            expected = np.array([self.get_expected_trans(region1, region2)], np.float64) if self.trans \
                else self._expected_vectors[region1]
        igd = -1 if self.trans else int(self.ignore_diags)
        pad = (self.rescale_size - 1) // 2 if rescale else self.pad_bins
        kw = dict(height=hh if rescale else None, width=ww if rescale else None, ignore_diags=igd, mode=mode,
                  coverage=bool(mode & MODE_COV))
        src = getattr(self, "_window_source", None)
        if src is not None:
            return src(self, expected, r0, c0, pad, **kw)
        from . import dist as _dist
        eng = _engine_for(self._aclr, _dist.local_device())
        bins = self._aclr.bins()
        eng.load_bins(bins[self.clr_weight_name][:].values if self.clr_weight_name else None,
                      bins[self.coverage_norm][:].values if self.coverage_norm else None)
        if expected is not None:
            eng.set_expected(expected)
        got = eng.extract(r0, c0, pad, **kw)
        return got if kw["coverage"] else (got, None, None)

    def _callback_region(self, region1, region2, groupby, modify_2Dintervals_func, postprocess_func, extra_sum_funcs):
        """pileup_region with per-snippet Python callbacks (reference :1285-1358): the region's window table as the
        reference's dict rows -> _stream_snips (windows from the GPU) -> accumulate_stream (callbacks, _add_snip)."""
        if region2 is None:
            region2 = region1
        b = self.region_snippets(region1, region2, groupby=groupby, modify_2Dintervals_func=modify_2Dintervals_func,
                                 columns=None, keep_table=True)
        rows = []
        if b is not None and b["n"] > 0:
            frame = pd.DataFrame({k: v for k, v in b["table"].items() if not k.startswith("_gc_")})
            frame["kind"] = np.where(b["kind"] == KIND_ROI, "ROI", "control")
            frame = assign_groups(frame, groupby)
            frame = frame.reindex(columns=list(frame.columns) + ["data", "cov_start", "cov_end", "horizontal_stripe",
                                                                 "vertical_stripe"])
            rows = frame.to_dict(orient="records")
        return self.accumulate_stream(self._stream_snips(iter(rows), region1, region2),
                                      postprocess_func=postprocess_func, extra_funcs=extra_sum_funcs)

    def _stream_snips(self, intervals, region1, region2=None):
        """The reference's snippet generator (:1059-1191) with the per-snippet matrix work done by the GPU: rows of
        ``intervals`` (dicts as CoordCreator.pos_stream yields them: chromosome-relative stBin1/endBin1/stBin2/endBin2,
        kind, group, ...) come back, in order, with ``data`` (window as the reference builds it: balanced, NaN-masked,
        / expected, rescaled), ``cov_start`` / ``cov_end``, stripes and coordinates filled in; a row whose window
        leaves its region is dropped (:1111-1114); with expected and ooe=False each row is followed by its expected
        twin of kind "control".  Windows are fetched from the engine in batches (pup_extract)."""
        from .engine import MODE_EXPECTED
        rows = [r for r in intervals if r is not None]
        if not rows:
            return
        if region2 is None:
            region2 = region1
        exp_as_control = bool(self.expected) and not self.ooe
        if exp_as_control and self.rescale and self.coverage_norm:
            raise NotImplementedError("callbacks with rescale + coverage_norm + non-ooe expected")
        lo1, hi1, off1 = self._global_extents[region1]
        lo2, hi2, off2 = self._global_extents[region2]
        ml1, ml2 = self.view_df_extents[region1][0], self.view_df_extents[region2][0]
        st1 = np.array([r["stBin1"] for r in rows], np.int64); en1 = np.array([r["endBin1"] for r in rows], np.int64)
        st2 = np.array([r["stBin2"] for r in rows], np.int64); en2 = np.array([r["endBin2"] for r in rows], np.int64)
        keep = np.flatnonzero((st1 + off1 >= lo1) & (en1 + off1 <= hi1) & (st2 + off2 >= lo2) & (en2 + off2 <= hi2))
        for i in keep:      # bins become region-relative before the snippet is handed on (reference :1104-1110)
            r = rows[i]
            r["stBin1"], r["endBin1"] = int(st1[i] - ml1), int(en1[i] - ml1)
            r["stBin2"], r["endBin2"] = int(st2[i] - ml2), int(en2[i] - ml2)
        b = {"r0": (st1 + off1).astype(np.int32), "c0": (st2 + off2).astype(np.int32),
             "h": (en1 - st1).astype(np.int32), "w": (en2 - st2).astype(np.int32)}
        if not self.rescale and len(keep) and not (np.all(b["h"][keep] == 2 * self.pad_bins + 1)
                                                   and np.all(b["w"][keep] == 2 * self.pad_bins + 1)):
            raise ValueError("window size differs from 2*pad_bins+1")
        for lo in range(0, len(keep), self._CALLBACK_BATCH):
            sel = keep[lo:lo + self._CALLBACK_BATCH]
            data, cov_s, cov_e = self._windows(b, region1, region2, sel)
            exp_data = self._windows(b, region1, region2, sel, mode_extra=MODE_EXPECTED)[0] if exp_as_control else None
            yield from self._snip_stream(rows, sel, data, cov_s, cov_e, exp_data)

    def accumulate_stream(self, snip_stream, postprocess_func=None, extra_funcs=None):
        """Pile a stream of snippets (dicts with data, cov_start, cov_end, kind, group, ...) up per kind and group
        (reference :1236-1283): postprocess_func maps every snippet (it may return several), _add_snip adds it,
        "all" is the sum of the groups when no snippet was grouped under that name."""
        from functools import reduce
        if postprocess_func is not None:
            snip_stream = map(postprocess_func, snip_stream)
        piles = {"ROI": SnipAccumulator(extra_funcs), "control": SnipAccumulator(extra_funcs)}
        for snip in _collapse(snip_stream):
            key = snip["group"]
            piles[snip["kind"]].add(key if isinstance(key, str) else tuple(key), snip)
        outdict = {kind: pile.entries for kind, pile in piles.items()}
        sum_func = partial(sum_pups, extra_funcs=extra_funcs)
        if "all" not in outdict["ROI"]:
            outdict["ROI"]["all"] = reduce(sum_func, outdict["ROI"].values(), self.empty_pup)
        if self.control or (self.expected and not self.ooe):
            if "all" not in outdict["control"]:
                outdict["control"]["all"] = reduce(sum_func, outdict["control"].values(), self.empty_pup)
        return outdict

    def _snip_stream(self, rows, sel, data, cov_s, cov_e, exp_data):
        """The reference's _stream_snips from the window onwards (:1116-1191): fill the row dict and yield it
        (followed by its expected twin when the expected serves as control)."""
        exp_as_control = exp_data is not None
        for j, i in enumerate(sel):
            snip = rows[i]
            if exp_as_control and snip["kind"] == "ROI":
                exp_snip = snip.copy()
                exp_snip["kind"] = "control"
                exp_snip["data"] = exp_data[j]
                exp_snip["coordinates"] = []
            if cov_s is not None:
                snip["cov_start"], snip["cov_end"] = cov_s[j], cov_e[j]
            snip["data"] = data[j]
            if self.store_stripes:
                cntr = int(np.floor(snip["data"].shape[0] / 2))
                snip["horizontal_stripe"] = np.array(snip["data"][cntr, :], dtype=float)
                snip["vertical_stripe"] = np.array(snip["data"][:, cntr][::-1], dtype=float)
                snip["coordinates"] = ".".join(str(snip[c]) for c in ("chrom1", "start1", "end1", "chrom2", "start2",
                                                                      "end2"))
            else:
                snip["horizontal_stripe"], snip["vertical_stripe"], snip["coordinates"] = [], [], []
            yield snip
            if exp_as_control and snip["kind"] == "ROI":
                yield exp_snip

    # -- by-X wrappers (reference :1656-1919) ---------------------------------------------------------------------
    def pileupsByStrandWithControl(self, nproc=None, groupby=[], ignore_group_order=False):
        if nproc is None:
            nproc = self.nproc
        pups = self.pileupsWithControl(nproc=nproc, groupby=["strand1", "strand2"] + groupby,
                                       ignore_group_order=ignore_group_order)
        pups.insert(0, "orientation", (pups["strand1"] + pups["strand2"]).replace({"allall": "all"}))
        return pups

    def pileupsByWindowWithControl(self, nproc=None):
        """One pile-up per feature: every pair is counted for both of its features (reference :1696-1755)."""
        if nproc is None:
            nproc = self.nproc
        if self.local:
            raise ValueError("Cannot do by-window pileups for local")
        if self.kind != "bed":
            raise ValueError("Can't make by-window pileups without making combinations")
        if self.store_stripes:
            # per-feature stripe lists: the reference's own route (postprocess_func=group_by_region, :1724-1726) —
            # windows from pup_extract, grouping on the host
            from .lib.puputils import group_by_region
            pups = self.pileupsWithControl(nproc=nproc, postprocess_func=group_by_region)
        else:
            pups = self.pileupsWithControl(nproc=nproc, _by_window=True)
            grp = pups["group"].to_numpy()
            kinds = set(map(type, grp.tolist()))             # (C-speed: a row per feature)
            if kinds <= {int, str, np.int64, np.int32} and all(g == "all" for g in grp[np.fromiter(map(str.__instancecheck__, grp), bool, len(grp))]):
                return self._by_window_frame(pups, grp)
        is_all = pups["group"].apply(lambda g: isinstance(g, str) and g == "all")
        coords = pd.DataFrame([("all", -1, -1) if a else tuple(g) for a, g in zip(is_all, pups["group"])],
                              index=pups.index, columns=["chrom", "start", "end"])
        pups = pd.concat([coords, pups], axis=1)
        pups[["start", "end"]] = pups[["start", "end"]].astype(int)
        pups = pups.drop(columns="group")
        # bioframe.sort_bedframe(df, view_df): by the view's chromosome order, then start, end; "all" (not in the
        # view) lands at the end
        view_chroms = list(dict.fromkeys(self.view_df["chrom"]))
        rank = {c: i for i, c in enumerate(view_chroms)}
        key = pups["chrom"].map(lambda c: rank.get(c, len(rank)))
        pups = pups.assign(_k=key).sort_values(["_k", "start", "end"], kind="stable").drop(columns="_k")
        return pups.reset_index(drop=True)

    def _by_window_frame(self, pups, grp):
        """The by-window output frame from rows keyed by feature number: chrom / start / end columns in front, "all" as
        ("all", -1, -1), rows in bioframe.sort_bedframe order (the view's chromosome order, then start, end; "all" — not in the
        view — last).  The reference does this per row (coolpup.py:1729-1755)."""
        is_all = np.fromiter(map(str.__instancecheck__, grp), bool, len(grp))
        u = np.where(is_all, -1, grp).astype(np.int64)
        _, rep = self.CC.feature_ids()
        c = self.CC._cache()
        names = np.asarray(list(c["chrom_code"]), dtype=object)
        row = rep[np.maximum(u, 0)]
        code = c["c"][row]
        chrom = names[code]
        chrom[is_all] = "all"
        start = np.where(is_all, -1, c["start"][row]).astype(np.int64)
        end = np.where(is_all, -1, c["end"][row]).astype(np.int64)
        view_chroms = list(dict.fromkeys(self.view_df["chrom"]))
        rank_of = {ch: i for i, ch in enumerate(view_chroms)}
        rank = np.array([rank_of.get(ch, len(rank_of)) for ch in names.tolist()], np.int64)[code]
        rank[is_all] = len(rank_of)
        order = np.lexsort((end, start, rank))
        cols = {"chrom": chrom[order], "start": start[order], "end": end[order]}
        for name in pups.columns:
            if name != "group":
                cols[name] = pups[name].to_numpy()[order]
        return pd.DataFrame(cols, copy=False)

    def _distance_edges(self, distance_edges):
        if not (isinstance(distance_edges, str) and distance_edges == "default"):
            if not all(isinstance(n, (int, np.integer)) for n in distance_edges):
                raise ValueError("Distance edges must be integers")
            distance_edges = list(np.sort(distance_edges))
            for _ in range(len(distance_edges)):
                if np.min(distance_edges) < self.mindist:
                    distance_edges[np.argmin(distance_edges)] = self.mindist
                else:
                    break
        return distance_edges

    @staticmethod
    def _separation_labels(pups):
        def label(x):
            if isinstance(x, str) and x == "all":
                return x
            if len(x) == 2:
                return f"{x[0] / 1000000}Mb-\n{x[1] / 1000000}Mb"
            return f"{x[0] / 1000000}Mb+"
        return pups["distance_band"].apply(label)

    def pileupsByDistanceWithControl(self, nproc=None, distance_edges="default", groupby=[],
                                     ignore_group_order=False):
        if nproc is None:
            nproc = self.nproc
        if self.trans:
            raise ValueError("Cannot do by-distance pileups for trans")
        elif self.local:
            raise ValueError("Cannot do by-distance pileups for local")
        distance_edges = self._distance_edges(distance_edges)
        bin_func = partial(bin_distance_intervals, band_edges=distance_edges)
        pups = self.pileupsWithControl(nproc=nproc, modify_2Dintervals_func=bin_func,
                                       groupby=["distance_band"] + groupby, ignore_group_order=ignore_group_order,
                                       _columns=("distance",))
        pups = pups.loc[pups["distance_band"] != (), :].reset_index(drop=True)
        pups.insert(0, "separation", self._separation_labels(pups))
        i = np.where(pups["separation"] == "all")[0]
        pups = pd.concat([pups.drop(i).sort_values("distance_band"), pups.iloc[i, :]],
                         ignore_index=True).reset_index(drop=True)
        return pups

    def pileupsByStrandByDistanceWithControl(self, nproc=None, distance_edges="default", groupby=[],
                                             ignore_group_order=False):
        if nproc is None:
            nproc = self.nproc
        if self.trans:
            raise ValueError("Cannot do by-distance pileups for trans")
        distance_edges = self._distance_edges(distance_edges)
        bin_func = partial(bin_distance_intervals, band_edges=distance_edges)
        pups = self.pileupsWithControl(nproc=nproc, modify_2Dintervals_func=bin_func,
                                       groupby=["strand1", "strand2", "distance_band"] + groupby,
                                       ignore_group_order=ignore_group_order, _columns=("distance",))
        pups.insert(0, "orientation", (pups["strand1"] + pups["strand2"]).replace({"allall": "all"}))
        pups = pups.loc[pups["distance_band"] != (), :].reset_index(drop=True)
        pups.insert(0, "separation", self._separation_labels(pups))
        i = np.where(pups["separation"] == "all")[0]
        pups = pd.concat([pups.drop(i).sort_values(["orientation", "distance_band"]), pups.iloc[i, :]],
                         ignore_index=True).reset_index(drop=True)
        return pups


def plan_tiles_with_windows(plan):
    """Sorted numbers of the tiles this process's engine calls pile windows into (what it contributes to a by-window exchange)."""
    held = np.zeros(plan["T"], bool)
    for c in plan["calls"]:
        held |= np.diff(np.asarray(c["tile_ptr"])) > 0
    return np.flatnonzero(held).astype(np.int32)


def _engine_call(region1, region2, expected, r0, c0, flip, tile, T, igd, mode, extra=None):
    """One pup_accumulate call: snippets grouped by (tile, flip) — stable, so genome order is kept inside a
    group; within a tile the anti-transposed snippets come last (flip_from marks where they start)."""
    n = len(tile)
    r0 = np.asarray(r0).astype(np.int32, copy=False)            # narrow BEFORE permuting: half the bytes to move
    c0 = np.asarray(c0).astype(np.int32, copy=False)
    any_flip = flip is not None and bool(np.any(flip))
    key = np.asarray(tile).astype(np.int64, copy=False) * 2
    if any_flip:
        key = key + np.asarray(flip, bool)
    counts = np.bincount(key, minlength=2 * T)                   # per (tile, flip) segment
    if n > 1 and not np.all(key[1:] >= key[:-1]):
        if 2 * T < 65536:
            o = np.argsort(key.astype(np.uint16), kind="stable")   # 16-bit keys: numpy uses a radix sort, O(n)
        else:
            o = np.argsort(key, kind="stable")
        r0, c0 = r0[o], c0[o]
        if extra:
            extra = {k: np.asarray(v)[o] for k, v in extra.items()}
    per_tile = counts.reshape(T, 2)
    tile_ptr = np.concatenate([[0], np.cumsum(per_tile.sum(axis=1))]).astype(np.int64)
    flip_from = (tile_ptr[:-1] + per_tile[:, 0]).astype(np.int64) if any_flip else None
    # after the grouping the tile / flip of every snippet follow from the segment sizes
    tile_sorted = np.repeat(np.arange(T, dtype=np.int32), per_tile.sum(axis=1))
    flip_sorted = np.repeat(np.tile(np.array([0, 1], np.uint8), T), counts) if any_flip else None
    call = {"region1": region1, "region2": region2, "expected": expected,
            "r0": np.ascontiguousarray(r0), "c0": np.ascontiguousarray(c0), "flip": flip_sorted,
            "flip_from": flip_from, "tile": tile_sorted, "tile_ptr": tile_ptr, "ignore_diags": igd, "mode": mode}
    for k, v in (extra or {}).items():
        call[k] = np.ascontiguousarray(v, np.int32)
    return call


class _Plan(dict):
    """A plan whose rarely used entries (the per-region window tables the inf-cell merge replays) are built on first use."""

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.lazy = {}

    def __missing__(self, key):
        if key not in self.lazy:
            raise KeyError(key)
        val = self[key] = self.lazy[key]()
        return val


class _Call(dict):
    """One engine call of a plan.  The per-window `tile` array follows from `tile_ptr` (windows are grouped by tile); only
    the oracle replay of the tests and the rare inf-cell merge ask for it, so it is built on first use — np.repeat over
    1.1e7 windows cost more than the GPU's whole pile-up."""

    def __missing__(self, key):
        if key != "tile":
            raise KeyError(key)
        tp = self["tile_ptr"]
        val = np.repeat(np.arange(len(tp) - 1, dtype=np.int32), np.diff(tp))
        self[key] = val
        return val


def _engine_call_parts(region1, region2, expected, parts, T, igd, mode, rescale):
    """_engine_call over the concatenation of per-region parts (r0, c0, flip, tile, h, w) WITHOUT concatenating
    first: each part is grouped by (tile, flip) on its own (cache-sized stable radix sorts), then the segments
    are copied straight to their final place — region order inside a segment is the stable order of the whole."""
    nk = 2 * T
    if not rescale and all(p[2] is None or not np.any(p[2]) for p in parts):
        # nothing flipped: one stable counting sort of the library over all parts, into page-locked arrays
        from . import engine as _engine
        r0, c0, tile_ptr = _engine.group_tiles([(p[0], p[1], p[3]) for p in parts], T)
        return _Call({"region1": region1, "region2": region2, "expected": expected, "r0": r0, "c0": c0, "flip": None,
                      "flip_from": None, "tile_ptr": tile_ptr, "ignore_diags": igd, "mode": mode})
    parts = [(p[0], p[1], p[2], np.asarray(p[3]), p[4], p[5]) for p in parts]      # (run-coded tiles as arrays from here on)
    if len(parts) == 1 or nk >= 65536 or nk * len(parts) > 200_000:
        f = [np.concatenate([p[k] for p in parts]) if len(parts) > 1 else parts[0][k] for k in range(6)]
        return _engine_call(region1, region2, expected, f[0], f[1], f[2], f[3], T, igd, mode,
                            extra={"h": f[4], "w": f[5]} if rescale else None)
    nf = 6 if rescale else 2
    counts = np.zeros((len(parts), nk), np.int64)
    grouped = []
    any_flip = False
    for i, p in enumerate(parts):
        key = np.asarray(p[3]).astype(np.uint16) * np.uint16(2)
        if p[2] is not None and np.any(p[2]):
            any_flip = True
            key = key + np.asarray(p[2], bool)
        counts[i] = np.bincount(key, minlength=nk)
        cols = [np.asarray(p[0]).astype(np.int32, copy=False), np.asarray(p[1]).astype(np.int32, copy=False)]
        if rescale:
            cols += [np.asarray(p[4]).astype(np.int32, copy=False), np.asarray(p[5]).astype(np.int32, copy=False)]
        if len(key) > 1 and not np.all(key[1:] >= key[:-1]):
            o = np.argsort(key, kind="stable")                     # 16-bit keys: radix sort
            cols = [c[o] for c in cols]
        grouped.append(cols)
    total = counts.sum(axis=0)
    n = int(total.sum())
    key_start = np.concatenate([[0], np.cumsum(total)])[:-1]
    dst = key_start[None, :] + np.cumsum(counts, axis=0) - counts      # where part i's segment of key k goes
    out = [np.empty(n, np.int32) for _ in range(len(grouped[0]))]
    for i, cols in enumerate(grouped):
        src = np.concatenate([[0], np.cumsum(counts[i])])
        for k in np.flatnonzero(counts[i]):
            a, b, d = int(src[k]), int(src[k + 1]), int(dst[i, k])
            for o_, c_ in zip(out, cols):
                o_[d:d + (b - a)] = c_[a:b]
    per_tile = total.reshape(T, 2)
    tile_ptr = np.concatenate([[0], np.cumsum(per_tile.sum(axis=1))]).astype(np.int64)
    flip_from = (tile_ptr[:-1] + per_tile[:, 0]).astype(np.int64) if any_flip else None
    call = {"region1": region1, "region2": region2, "expected": expected, "r0": out[0], "c0": out[1],
            "flip": np.repeat(np.tile(np.array([0, 1], np.uint8), T), total) if any_flip else None,
            "flip_from": flip_from, "tile": np.repeat(np.arange(T, dtype=np.int32), per_tile.sum(axis=1)),
            "tile_ptr": tile_ptr, "ignore_diags": igd, "mode": mode}
    if rescale:
        call["h"], call["w"] = out[2], out[3]
    del nf
    return call


def _gather_stripes(plan, acc):
    """Multi-GPU: every rank ends up with the stripes of all regions, in region order (acc["stripe_jobs"] /
    acc["stripes"]); a single process keeps what it has."""
    from . import dist as _dist
    if _dist.world()[1] == 1:
        return
    mine = {job["region"]: (job["gid"], job["coords"], h, v) for job, (h, v) in zip(plan["stripe_jobs"], acc["stripes"])}
    every = _dist.merge_dicts(mine)
    acc["stripe_jobs"] = [{"gid": every[k][0], "coords": every[k][1]} for k in sorted(every)]
    acc["stripes"] = [(every[k][2], every[k][3]) for k in sorted(every)]


def _collect_stripes(plan, acc):
    """Per group key: (coordinates [n,6] str, horizontal [n,W], vertical [n,W]) in the reference's accumulation
    order — regions in order, snippets in stream order; the "all" row of a grouped pile-up concatenates, region by
    region, the groups in their order of first appearance in that region (reduce(sum_pups), coolpup.py:1271-1282)."""
    gid = plan["gid"]
    inv = {v: k for k, v in gid.items()}
    per = {}
    def push(key, co, h, v):
        e = per.setdefault(key, ([], [], []))
        e[0].append(co); e[1].append(h); e[2].append(v)
    for job, (h, v) in zip(acc.get("stripe_jobs", plan["stripe_jobs"]), acc["stripes"]):
        g = job["gid"]
        if plan["grouped"]:
            for code in pd.unique(g):                       # first-appearance order within the region
                sel = np.flatnonzero(g == code)
                push(inv[int(code)], job["coords"][sel], h[sel], v[sel])
                push("all", job["coords"][sel], h[sel], v[sel])
        else:
            push("all", job["coords"], h, v)
    return {k: (np.concatenate(e[0]), np.concatenate(e[1]), np.concatenate(e[2])) for k, e in per.items()}


def iter_expected_subcalls(plan, call):
    """Split an engine call that uses the plan's expected TABLE into (expected, sub-call) pieces that each use one
    plain expected vector / scalar (what pup_set_expected takes).  Host-only helper for replaying a plan on
    back-ends that have no table support (the test oracle)."""
    if not isinstance(call["expected"], str):
        yield call["expected"], call
        return
    et = plan["expected_table"]
    transpose = bool(call["mode"] & 0x08)
    rr = np.searchsorted(et["end"], call["r0"], side="right")
    cc = np.searchsorted(et["end"], call["c0"], side="right")
    key = rr * (len(et["end"]) + 1) + (cc if et["pair"] is not None else 0)
    for kval in np.unique(key):
        sel = np.flatnonzero(key == kval)
        i = int(rr[sel[0]]); j = int(cc[sel[0]])
        if et["pair"] is not None:
            expected = np.array([et["pair"][i, j] if i < len(et["end"]) and j < len(et["end"]) else np.nan])
        else:
            expected = et["vectors"][i] if i < len(et["end"]) else np.array([np.nan, np.nan])
        sub = dict(call)
        sub["tile"] = call["tile"][sel]                  # (built on first use: _Call)
        for name in ("r0", "c0", "flip", "h", "w"):
            if call.get(name) is not None:
                sub[name] = call[name][sel]
        T = len(call["tile_ptr"]) - 1
        sub["tile_ptr"] = np.concatenate([[0], np.cumsum(np.bincount(sub["tile"], minlength=T))]).astype(np.int64)
        sub["flip_from"] = None if sub.get("flip") is None else \
            sub["tile_ptr"][1:] - np.bincount(sub["tile"][sub["flip"].astype(bool)], minlength=T)
        sub["expected"] = expected
        yield expected, sub
    del transpose


def _tiles_to_pups(plan, acc):
    """Tiles of ONE region as the reference's pileup_region dict.  In a grouped pile-up the fold of the groups into
    "all" (reduce(sum_pups), coolpup.py:1271-1282) passes every tile through nan_to_num — the groups' own included."""
    G, gid, order = plan["G"], plan["gid"], plan["order"]
    out = {"ROI": {}, "control": {}}
    for kind, label in ((KIND_ROI, "ROI"), (KIND_CONTROL, "control")):
        members = [k for k in order[kind] if not (isinstance(k, str) and k == "all")]
        if plan["grouped"] and members:
            a = kind * G + gid["all"]
            run = np.zeros_like(acc["sum"][a])
            for k in members:
                t = kind * G + gid[k]
                acc["sum"][t] = np.nan_to_num(acc["sum"][t])
                run = np.nan_to_num(run) + acc["sum"][t]
            acc["sum"][a] = run
            for name in ("num", "n", "cov_start", "cov_end"):
                acc[name][a] = acc[name][[kind * G + gid[k] for k in members]].sum(axis=0)
        for key in order[kind]:
            t = kind * G + gid[key]
            out[label][key] = {"data": acc["sum"][t].copy(), "num": acc["num"][t].copy(), "n": int(acc["n"][t]),
                               "cov_start": acc["cov_start"][t].copy(), "cov_end": acc["cov_end"][t].copy(),
                               "horizontal_stripe": [], "vertical_stripe": [], "coordinates": []}
    return out


def _factorize_rows(cols, decoders=None):
    """Row-wise factorisation of several columns: (codes int64, keys = list of tuples in first-appearance
    order).  decoders[j], when given, maps the stored value of column j to the value shown in the key."""
    decoders = decoders or [None] * len(cols)
    per = []
    for c in cols:
        c = np.asarray(c)
        if c.dtype.kind in "iu" and len(c) and 0 <= int(c.min()) and int(c.max()) < 4096:
            per.append((c, np.arange(int(c.max()) + 1)))      # already small codes: no hashing needed
        else:
            per.append(pd.factorize(c, sort=False))
    small = 1
    for _, uniq in per:
        small *= len(uniq) + 1
    if small <= (1 << 22) and len(cols[0]):
        # dense mixed-radix code -> first-appearance numbering with two O(n) passes and one small table
        combined = np.zeros(len(cols[0]), np.int32)
        for codes, uniq in per:
            combined = combined * np.int32(len(uniq) + 1) + codes.astype(np.int32, copy=False)
        n = len(combined)
        first = np.full(small, -1, np.int64)
        first[combined[::-1]] = np.arange(n - 1, -1, -1)         # reversed writes: the first occurrence wins
        seen = np.flatnonzero(first >= 0)
        seen = seen[np.argsort(first[seen], kind="stable")]
        lut = np.zeros(small, np.int64)
        lut[seen] = np.arange(len(seen))
        keys = []
        for r in first[seen]:
            vals = []
            for j, (cj, uj) in enumerate(per):
                v = uj[cj[r]]
                vals.append(decoders[j](v) if decoders[j] is not None else v)
            keys.append(tuple(vals))
        return lut[combined], keys
    radix = 1
    for _, uniq in per:
        radix *= len(uniq) + 1
    if radix < 2 ** 62:
        combined = np.zeros(len(cols[0]), np.int64)
        for codes, uniq in per:
            combined = combined * (len(uniq) + 1) + codes
    else:   # too many distinct values for a mixed-radix code: factorize tuples
        combined = pd.factorize(pd.Series(list(zip(*[codes for codes, _ in per]))), sort=False)[0]
    codes, uniq = pd.factorize(combined, sort=False)
    n = len(codes)
    first_rows = np.empty(len(uniq), np.int64)
    first_rows[codes[::-1]] = np.arange(n - 1, -1, -1)       # reversed writes: the first occurrence wins
    keys = []
    for r in first_rows:
        vals = []
        for j, (cj, uj) in enumerate(per):
            v = uj[cj[r]]
            vals.append(decoders[j](v) if decoders[j] is not None else v)
        keys.append(tuple(vals))
    return codes.astype(np.int64), keys


def _is_builtin_modify(func):
    """True for (compositions of) this module's own bin_distance_intervals / flip_mark_intervals_func."""
    if isinstance(func, partial):
        if func.func is flip_mark_intervals_func:
            inner = func.keywords.get("extra_func")
            return inner is None or _is_builtin_modify(inner)
        return func.func is bin_distance_intervals
    return func is bin_distance_intervals


def _apply_builtin_modify(tbl, func):
    """Column-table versions of flip_mark_intervals_func / bin_distance_intervals (no per-row Python objects).
    Returns (table, decoders): 'distance_band' is stored as the searchsorted id and decoded to the band tuple."""
    decoders = {}
    kw = func.keywords if isinstance(func, partial) else {}
    base = func.func if isinstance(func, partial) else func
    if base is flip_mark_intervals_func:
        if kw.get("flip_negative_strand"):
            tbl["flip"] = tbl["strand1"] == "-"
        else:
            fb = kw["flipby"]
            tbl["flip"] = np.asarray(tbl[f"{fb}1"] > tbl[f"{fb}2"], dtype=bool)
        inner = kw.get("extra_func")
        if inner is not None:
            tbl, decoders = _apply_builtin_modify(tbl, inner)
        return tbl, decoders
    edges = kw.get("band_edges", "default")
    if isinstance(edges, str) and edges == "default":
        edges = _default_band_edges()
    d = np.asarray(tbl["distance"])
    # control rows repeat the distances of the ROI rows (ROI block followed by whole copies of it, _control_cols): when
    # the table has that shape the band ids of the first block serve every copy
    n = len(d)
    n_roi = int(np.count_nonzero(np.asarray(tbl["kind"]) == KIND_ROI)) if "kind" in tbl else n
    if 0 < n_roi < n and n % n_roi == 0 and bool(np.all(d.reshape(n // n_roi, n_roi) == d[:n_roi])):
        ids = np.tile(np.searchsorted(edges, d[:n_roi], side="right"), n // n_roi)     # one comparison pass instead
    else:
        ids = np.searchsorted(edges, d, side="right")
    tbl["distance_band"] = ids
    decoders["distance_band"] = lambda i, e=edges: tuple(e[i - 1:i + 1])
    return tbl, decoders


# ------------------------------------------------------------------------------------------------------
# pileup()
# ------------------------------------------------------------------------------------------------------
def pileup(clr, features, features_format="bed", view_df=None, expected_df=None, expected_value_col="balanced.avg",
           clr_weight_name="weight", flank=100000, minshift=10**5, maxshift=10**6, nshifts=0, ooe=True,
           mindist="auto", maxdist=None, min_diag=2, subset=0, by_window=False, by_strand=False, by_distance=False,
           groupby=[], ignore_group_order=False, flip_negative_strand=False, local=False, coverage_norm=False,
           trans=False, rescale=False, rescale_flank=1, rescale_size=99, store_stripes=False, nproc=1, seed=None):
    """Create pile-ups — same signature, defaults and returned columns as the reference's ``pileup``
    (coolpup.py:1922-2279)."""
    if by_distance is not False:
        if local:
            raise ValueError("Can't do local pileups by distance, please specify only one of those arguments")
        if isinstance(by_distance, np.ndarray):
            try:
                distance_edges = [int(i) for i in by_distance]
            except Exception as e:
                raise ValueError(
                    "Distance bin edges have to be an iterable of integers or convertable to integers") from e
            by_distance = True
        elif by_distance is True or (isinstance(by_distance, str) and by_distance == "default"):
            distance_edges = "default"
            by_distance = True
        else:
            raise ValueError("Invalid by_distance value, should be either True, 'default' or a list of integers")

    if not rescale:
        rescale_flank = None
    if nshifts > 0 or subset > 0:
        from . import dist as _dist
        seed = _dist.shared_seed(seed)       # several ranks, no seed given: rank 0 draws one for all
    if seed is not None:
        np.random.seed(seed)
    if nproc == 0:
        nproc = -1
    if view_df is None:
        view_df = _make_cooler_view(clr, remember=True)
    else:
        try:
            view_df = _make_viewframe(view_df, clr.chromsizes).reset_index(drop=True)
            pos = [list(clr.chromnames).index(c) for c in view_df["chrom"]]
            keyed = list(zip(pos, view_df["start"]))
            if keyed != sorted(keyed):
                raise ValueError("view is not sorted by chromosome order and start")
        except Exception as e:
            raise ValueError("view_df is not a valid viewframe or incompatible") from e
        _remember_view(view_df, _view_signature(clr.chromsizes))
    control = nshifts > 0
    if expected_df is None:
        expected_value_col = None
    if mindist is None:
        mindist = "auto"
    if maxdist is None:
        maxdist = np.inf
    if rescale and rescale_size % 2 == 0:
        raise ValueError("Please provide an odd rescale_size")
    chroms = list(view_df["chrom"].unique())
    if by_window:
        if features_format != "bed":
            raise ValueError("Can't make by-window pileups without making combinations")
        if local:
            raise ValueError("Can't make local by-window pileups")

    # a plain pile-up with random-shift controls: its draws can start while the features are still being sorted (CoordCreator.
    # _start_early_draws) — told here which regions, in which order, the pile-up will walk
    hint = None
    if control and expected_df is None and not trans and not rescale and not by_window and not flip_negative_strand \
            and not ignore_group_order and not store_stripes and features_format == "bedpe":
        from . import dist as _dist
        if _dist.world()[1] == 1:
            hint = {"regions": list(zip(view_df["chrom"].astype(str).tolist(), view_df["start"].tolist(), view_df["end"].tolist()))}
    CC = CoordCreator(features=features, resolution=clr.binsize, features_format=features_format, flank=flank,
                      rescale_flank=rescale_flank, chroms=chroms, minshift=minshift, maxshift=maxshift,
                      nshifts=nshifts, mindist=mindist, maxdist=maxdist, local=local, subset=subset, seed=seed,
                      trans=trans, _draw_hint=hint)
    try:
        return _pileup_with(CC, clr, view_df, expected_df, expected_value_col, clr_weight_name, ooe, control, coverage_norm, rescale,
                            rescale_size, flip_negative_strand, min_diag, store_stripes, nproc, by_window, by_strand, by_distance,
                            distance_edges if by_distance else None, groupby, ignore_group_order)
    finally:
        early, CC._early_ahead = getattr(CC, "_early_ahead", None), None
        if early is not None:          # (an error before the pile-up adopted the early draws: stop them, generator back)
            early[0].cancel(early[2])


def _pileup_with(CC, clr, view_df, expected_df, expected_value_col, clr_weight_name, ooe, control, coverage_norm, rescale, rescale_size,
                 flip_negative_strand, min_diag, store_stripes, nproc, by_window, by_strand, by_distance, distance_edges, groupby,
                 ignore_group_order):
    """The second half of pileup() (reference coolpup.py:2210-2279): PileUpper, the by-X dispatch, the closing columns."""
    PU = PileUpper(clr=clr, CC=CC, view_df=view_df, clr_weight_name=clr_weight_name, expected=expected_df,
                   expected_value_col=expected_value_col, ooe=ooe, control=control, coverage_norm=coverage_norm,
                   rescale=rescale, rescale_size=rescale_size, flip_negative_strand=flip_negative_strand,
                   ignore_diags=min_diag, store_stripes=store_stripes, nproc=nproc)

    if by_window:
        pups = PU.pileupsByWindowWithControl()
        flags = (True, False, False)
        if groupby:
            warnings.warn("by-window not compatible with additional groupby")
    elif by_strand and by_distance:
        pups = PU.pileupsByStrandByDistanceWithControl(nproc=nproc, distance_edges=distance_edges, groupby=groupby,
                                                       ignore_group_order=ignore_group_order)
        flags = (False, True, True)
    elif by_strand:
        pups = PU.pileupsByStrandWithControl(groupby=groupby, ignore_group_order=ignore_group_order)
        flags = (False, True, False)
    elif by_distance:
        pups = PU.pileupsByDistanceWithControl(nproc=nproc, distance_edges=distance_edges, groupby=groupby,
                                               ignore_group_order=ignore_group_order)
        flags = (False, False, True)
    else:
        pups = PU.pileupsWithControl(groupby=groupby, ignore_group_order=ignore_group_order)
        flags = (False, False, False)
    pups["by_window"], pups["by_strand"], pups["by_distance"] = flags
    pups["groupby"] = [groupby] * pups.shape[0]
    with warnings.catch_warnings():          # pandas announces a dtype change of object fillna; the value is what matters
        warnings.simplefilter("ignore", FutureWarning)
        pups["expected"] = pups["expected"].fillna(False)
    pups["cooler"] = os.path.splitext(os.path.basename(clr.filename))[0]
    return pups


def snippet_batches(cc, clr, control=False, view_df=None):
    """Engine inputs for an ungrouped pile-up of every view region: (r0, c0, kind) global-bin arrays, regions
    concatenated in view order.  Host-only (no GPU); used by the benchmark and by the coordinate tests."""
    pu = PileUpper.__new__(PileUpper)
    pu.clr = clr
    pu._aclr = as_array_cooler(clr)
    pu.CC = cc
    pu.__dict__.update(cc.__dict__)
    pu.control = control
    pu.pad_bins = cc.flank // cc.resolution
    pu.ignore_group_order = False
    pu.store_stripes = False
    pu.flip_negative_strand = False
    vd = _make_cooler_view(clr) if view_df is None else _make_viewframe(view_df, clr.chromsizes)
    pu.view_df = vd.set_index("name")
    pu._global_extents = {}
    for name, r in pu.view_df.iterrows():
        lo, hi = pu._aclr.extent((r["chrom"], r["start"], r["end"]))
        pu._global_extents[name] = (lo, hi, pu._aclr.offset(r["chrom"]))
    pu.chroms = natsorted(list(set(cc.final_chroms) & set(clr.chromnames)))
    pu.view_df = pu.view_df[pu.view_df["chrom"].isin(pu.chroms)]
    r0s, c0s, ks = [], [], []
    for region1, region2 in pu._region_pairs():
        b = pu.region_snippets(region1, region2)
        if b is not None and b["n"]:
            r0s.append(b["r0"]); c0s.append(b["c0"]); ks.append(b["kind"])
    if not r0s:
        return np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int8)
    return np.concatenate(r0s), np.concatenate(c0s), np.concatenate(ks)
