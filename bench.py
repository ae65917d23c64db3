#!/usr/bin/env python3
"""bench.py — pile-up hot path on N MI355X GPUs (one process per GPU).

Workload (BASELINE.json configs[2], the configuration the north-star target is quoted on):
synthetic human-scale 10 kb pixel table (hg38 chr1-22,X, ~3e8 upper-triangular nnz, balanced with
2 % masked bins) + 1e6 random cis BEDPE pairs, pad=10 (21x21 windows), nshifts=10 random-shift
controls (seed 0) => ~1.1e7 snippets per step.  A "step" = one full pass: zero the accumulators,
pile up every snippet (ROI + controls) of this rank's shard from HBM-resident inputs, and (N>1)
all-reduce the packed sum/num/n/cov accumulators over RCCL.

Input order: the resident snippets are in the order `pileup()` hands them to the engine — grouped by tile
(ROI, control), inside a tile the reference's stream order (view regions in order; per region the ROI rows, then one
block per control shift: coolpuppy/coolpup.py:716-746).  Whatever re-ordering the engine wants (the device-side
block sort of the staged kernel) is therefore INSIDE the timed step.  The same workload pre-sorted into the engine's
block order is timed afterwards and reported as a secondary field (`preblocked`), never as `value`.

Scaling: N=1 is the whole workload on one GPU.  For N>1 the library's own multi-GPU path is what is timed (north_star:
"chromosomes shard across the GPUs with a final RCCL all-reduce"; coolpuppy_amd.PileUpper.run_plan): rank r owns the
chromosomes an LPT assignment by snippet count gives it, holds only THEIR rows of the pixel table, piles up only their
snippets, and the step ends with ONE in-place all-reduce of the packed tiles by the engine itself (pup_allreduce: RCCL on
the engine's stream; `--exchange torch` = torch.distributed.all_reduce on exported buffers instead).  For N>1 the JSON line's
`value` is the WEAK-scaling measurement of that sharded path — N x 1e6 pairs, one bigger pile-up, so that the per-GPU work stays
that of N=1 (`"scaling": "weak"`) — and the STRONG-scaling measurement of the fixed BASELINE workload (the same 1e6 pairs split
over the ranks: a 0.75 ms job, bounded by its fixed per-call cost and the all-reduce latency) rides along in the field `strong`;
`--scaling strong` makes that one the primary.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (pile-up kernels, HIP-event timed
inside this process) and `cpu_baseline` (N=1 only).
"""
import argparse
import hashlib
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import synth  # noqa: E402  (numpy only; torch is imported after generation)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--nshifts", type=int, default=10)
    ap.add_argument("--pad", type=int, default=10)
    ap.add_argument("--lam", type=float, default=4200.0, help="Poisson contacts drawn per row before de-duplication")
    ap.add_argument("--chroms", type=int, default=23, help="use the first K hg38 chromosomes (23 = all)")
    ap.add_argument("--cpu-sample", type=int, default=12_000_000, help="snippets timed on the C oracle (0 = no CPU baseline)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baselines (0 = all cores, capped at 64)")
    ap.add_argument("--ref-algo-sample", type=int, default=300_000,
                    help="snippets timed on the algorithm-faithful scipy restatement, one core (0 = skip)")
    ap.add_argument("--no-cache", action="store_true")
    ap.add_argument("--config", type=int, default=2, choices=[2, 3, 4],
                    help="BASELINE.json configs[k]: 2 = 1e6 cis pairs + 10 control shifts (default, the headline); 3 = the same pairs by "
                         "distance band x strand pair (42 tiles); 4 = 5e5 inter-chromosomal pairs, pad 25 — 3 and 4 run through the "
                         "library's plan path with its region / region-pair sharding (bench_plan.py)")
    ap.add_argument("--trans-nnz", type=int, default=50_000_000, help="--config 4: inter-chromosomal pixels added to the table")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the timing of the public pileup() call (N=1 only)")
    ap.add_argument("--scaling", default="auto", choices=["auto", "weak", "strong"],
                    help="which measurement is the JSON line's `value` for N>1 (the other one is the secondary field): "
                         "strong (auto) = the fixed --pairs workload sharded over the ranks; weak = N x --pairs pairs "
                         "sharded the same way (per-GPU work fixed)")
    ap.add_argument("--no-index", action="store_true", help="binary search only (skip the rank-bitmap index)")
    ap.add_argument("--variant", type=int, default=0, help="pup_set_tuning variant bits (kernel selection; 0 = default)")
    ap.add_argument("--exchange", default="native", choices=["native", "torch"],
                    help="N>1: how the packed tiles are summed every step — the engine's own RCCL call pup_allreduce, in place on "
                         "its stream (the library's default path), or torch.distributed.all_reduce on exported buffers")
    ap.add_argument("--strict-exchange", action="store_true",
                    help="N>1 with --exchange native: exit instead of falling back (labelled in the line) to torch's all-reduce when "
                         "the engine's RCCL communicator does not span the job")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (+ COOLPUPPY_AMD_BENCH_DEVICE=0) lets several ranks share ONE GPU "
                         "to smoke-test the N>1 code path on a single-GPU box")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------
# workload (cached under $TMPDIR so that back-to-back runs at N = 1, 2, 4, 8 do not regenerate it)
# ----------------------------------------------------------------------------------------------------
def exchange_verdict(world, rccl_ranks, strict, rank=0):
    """What an N > 1 line may call its exchange when `--exchange native` was asked for (pure: tests/test_bench_sharding.py).
    `--exchange native` is the path the line claims to measure: a communicator that is missing (rccl_ranks None) or spans fewer
    ranks than the job would time something else under that name.  strict: say so and stop.  Else (round 6: the engine-side path had
    never run on more than one rank before the driver's scaling run — a curve over torch's own RCCL all-reduce, LABELLED as such,
    says more than no curve): fall back, loudly, and say so in the line (`exchange`, `native_exchange_failed`)."""
    if rccl_ranks == world:
        return {"fallback": False, "message": "", "exchange": "pup_allreduce (RCCL on the engine's stream, in place)"}
    msg = (f"[bench] rank {rank}: --exchange native on {world} ranks, but the engine's RCCL communicator spans "
           f"{rccl_ranks} rank(s)")
    if strict:
        raise SystemExit(msg + " — refusing to report a number for a path that did not run "
                         "(--exchange torch times torch.distributed.all_reduce on exported buffers instead)")
    return {"fallback": True,
            "message": msg + " — FALLING BACK to torch.distributed.all_reduce (nccl backend = RCCL) on exported buffers; the "
                             "line says so in `exchange` and `native_exchange_failed`",
            "exchange": (f"FALLBACK: torch.distributed.all_reduce (RCCL) on exported buffers + a host synchronisation per step "
                         f"— the engine's own communicator spanned {rccl_ranks} of {world} ranks")}


def _tmp(name):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), name)


def workload_key(a):
    return hashlib.sha1(f"w6|{a.chroms}|{a.lam}|{a.pairs}|{a.nshifts}|{a.pad}".encode()).hexdigest()[:12]


def traffic_key(a):
    """Key of a measured-traffic entry (profiles/traffic.json): the workload, the BASELINE configuration and the kernel variant."""
    return hashlib.sha1(f"{workload_key(a)}|config{a.config}|variant{a.variant}|trans{a.trans_nnz if a.config == 4 else 0}".encode()).hexdigest()[:12]


def measured_traffic(a):
    """(HBM bytes per pile-up launch, source file) measured with rocprofv3 PMC passes on THESE kernel sources for this workload, or
    (None, None): tools/profile_round.sh -> tools/summarize_profiles.py write profiles/traffic.json."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath) or a.gpus != 1:
        return None, None
    try:
        tj = json.load(open(tpath)).get("entries", {}).get(traffic_key(a))
        if tj and tj.get("source_key") == source_key():
            return tj.get("hbm_bytes_per_launch"), tj.get("source")
    except Exception:
        pass
    return None, None


def source_key():
    """Hash of the kernel sources: measured HBM traffic (profiles/traffic.json) is only quoted for the kernels it was
    measured on."""
    h = hashlib.sha1()
    csrc = os.path.join(ROOT, "coolpuppy_amd", "csrc")
    for f in sorted(os.listdir(csrc)):                  # every kernel / engine source: the staged kernels live in their own headers
        if f.endswith((".hpp", ".hip", ".cpp", ".h")):
            with open(os.path.join(csrc, f), "rb") as fh:
                h.update(f.encode() + b"\0" + fh.read())
    return h.hexdigest()[:16]


def cooler_path(a):
    return _tmp("coolpuppy_amd_bench_cooler_" + hashlib.sha1(f"c4|{a.chroms}|{a.lam}".encode()).hexdigest()[:12] + ".npz")


def snippets_path(a, k):
    return _tmp(f"coolpuppy_amd_bench_snips_{workload_key(a)}_{k}.npz")


def _save(path, **arrays):
    tmp = path + f".tmp{os.getpid()}.npz"
    np.savez(tmp, **arrays)
    os.replace(tmp, path)


def _chromsizes(a):
    return {c: synth.HG38[c] for c in list(synth.HG38)[: a.chroms]}


def build_cooler(a):
    clr = synth.make_cooler(_chromsizes(a), binsize=10_000, lam=a.lam, seed=1000, name="synthetic_hg38_10kb",
                            parallel=True)
    bin1_offset, bin2_id, count = clr.pixel_table()
    return {"bin1_offset": bin1_offset, "bin2_id": bin2_id, "count": count,
            "weight": clr.bins()["weight"][:].values, "chrom_offset": clr.chrom_offset}


def _lower_bound_rows(indptr, col, target):
    """Per matrix row r: first pixel of the row with column >= target[r] (vectorised bisection over all rows)."""
    lo, hi = indptr[:-1].copy(), indptr[1:].copy()
    while True:
        act = lo < hi
        if not act.any():
            return lo
        mid = (lo + hi) >> 1
        less = np.zeros(len(lo), bool)
        less[act] = col[mid[act]] < target[act]
        lo = np.where(act & less, mid + 1, lo)
        hi = np.where(act & ~less, mid, hi)


def touched_tables(cool, r0, c0, W):
    """What the windows necessarily touch of the resident tables (the COMPULSORY traffic of one pass, whatever the
    kernel): per matrix row the column hull [min c0, max c0 + W) over the windows covering that row; pixels inside the
    hulls, 64-byte rank-bitmap index lines (320 columns each, anchored at the chromosome start) under the hulls, rows
    touched."""
    indptr, col, co = cool["bin1_offset"], cool["bin2_id"], cool["chrom_offset"]
    nb = len(indptr) - 1
    BIG = np.int64(1) << 40
    lo_w = np.full(nb + W, BIG, np.int64)
    hi_w = np.full(nb + W, -1, np.int64)
    order = np.argsort(r0, kind="stable")
    rs, cs = r0[order].astype(np.int64), c0[order].astype(np.int64)
    first = np.flatnonzero(np.concatenate([[True], rs[1:] != rs[:-1]]))
    lo_w[rs[first]] = np.minimum.reduceat(cs, first)
    hi_w[rs[first]] = np.maximum.reduceat(cs, first) + W
    lo_r, hi_r = np.full(nb, BIG, np.int64), np.full(nb, -1, np.int64)
    for p in range(W):                      # row r is covered by the windows whose first row is r-p
        lo_r[p:] = np.minimum(lo_r[p:], lo_w[: nb - p])
        hi_r[p:] = np.maximum(hi_r[p:], hi_w[: nb - p])
    rows = hi_r > lo_r
    lo_c = np.where(rows, lo_r, 0)
    hi_c = np.where(rows, hi_r, 0)
    pix = _lower_bound_rows(indptr, col, hi_c) - _lower_bound_rows(indptr, col, lo_c)
    start = co[np.clip(np.searchsorted(co, np.arange(nb), side="right") - 1, 0, len(co) - 2)]
    lines = np.where(rows, (hi_c - 1 - start) // 320 - (lo_c - start) // 320 + 1, 0)
    return {"pixels": int(pix[rows].sum()), "index_lines": int(lines.sum()), "rows": int(rows.sum())}


def build_snippets(a, cool, k, n_pairs=None):
    """Snippet set k: n_pairs (default a.pairs) random cis BEDPE pairs (pair seed 42+k) through the host-side coordinate
    layer (CoordCreator semantics, control-shift RNG seed k) -> (r0, c0) grouped by tile (ROI, control) in the
    reference's stream order, exactly what pileup() passes to pup_accumulate."""
    from coolpuppy_amd.cooler_lite import ArrayCooler
    from coolpuppy_amd.coolpup import CoordCreator, snippet_batches
    clr = ArrayCooler(_chromsizes(a), 10_000, cool["bin1_offset"], cool["bin2_id"], cool["count"],
                      bins={"weight": cool["weight"]}, filename="synthetic_hg38_10kb.cool")
    pairs = synth.random_cis_pairs(clr, n_pairs or a.pairs, min_sep=230_000, max_sep=5_000_000, seed=42 + k)
    np.random.seed(k)
    cc = CoordCreator(pairs, clr.binsize, features_format="bedpe", flank=a.pad * clr.binsize,
                      nshifts=a.nshifts, mindist="auto")
    r0, c0, kind = snippet_batches(cc, clr, control=a.nshifts > 0)
    order = np.argsort(kind, kind="stable")           # tile-grouped; stream order inside a tile
    r0, c0 = r0[order].astype(np.int32), c0[order].astype(np.int32)
    t = touched_tables(cool, r0, c0, 2 * a.pad + 1) if not n_pairs else {"pixels": 0, "index_lines": 0, "rows": 0}
    return {"r0_stream": r0, "c0_stream": c0, "n_roi": np.int64((kind == 0).sum()),
            "touched": np.array([t["pixels"], t["index_lines"], t["rows"]], np.int64)}


def _wait_for(path):
    while not os.path.exists(path):
        time.sleep(0.5)
    time.sleep(0.3)


def load_workload(a, rank, world):
    """Everything happens BEFORE any GPU runtime is initialised in this process (the generator forks workers)."""
    cpath = cooler_path(a)
    if rank == 0 and (a.no_cache or not os.path.exists(cpath)):
        t = time.time()
        _save(cpath, **build_cooler(a))
        print(f"[bench] cooler built in {time.time()-t:.1f}s -> {cpath}", file=sys.stderr, flush=True)
    _wait_for(cpath)
    z = np.load(cpath)
    cool = {k: z[k] for k in z.files}
    spath = snippets_path(a, 0)
    if rank == 0 and (a.no_cache or not os.path.exists(spath)):
        t = time.time()
        _save(spath, **build_snippets(a, cool, 0))
        print(f"[bench] rank {rank}: snippet set 0 built in {time.time()-t:.1f}s", file=sys.stderr, flush=True)
    _wait_for(spath)
    z = np.load(spath)
    cool.update({kk: z[kk] for kk in z.files})
    if world > 1:
        # the weak-scaling job: world x a.pairs pairs (one bigger pile-up, sharded like the strong one)
        wpath = _tmp(f"coolpuppy_amd_bench_snips_{workload_key(a)}_x{world}.npz")
        if rank == 0 and (a.no_cache or not os.path.exists(wpath)):
            t = time.time()
            _save(wpath, **build_snippets(a, cool, 1000 + world, n_pairs=a.pairs * world))
            print(f"[bench] weak-scaling snippet set ({a.pairs * world} pairs) built in {time.time()-t:.1f}s", file=sys.stderr, flush=True)
        _wait_for(wpath)
        z = np.load(wpath)
        cool.update({"weak_" + kk: z[kk] for kk in ("r0_stream", "c0_stream", "n_roi")})
    return cool


def shard_snippets(r0, c0, n_roi, chrom_offset, rank, world):
    """The library's sharding (coolpuppy_amd.dist.shard on chromosome window counts) applied to a tile-grouped snippet set:
    -> (r0, c0, tile_ptr) of this rank, the (lo, hi) bin ranges of the chromosomes it owns, and every chromosome's owner.
    Pure function of (inputs, rank, world): every rank computes the same assignment without communicating."""
    co = np.asarray(chrom_offset, np.int64)
    n = int(r0.shape[0])
    if world == 1:
        return r0, c0, np.array([0, n_roi, n], np.int64), [(int(co[0]), int(co[-1]))], np.zeros(len(co) - 1, np.int64)
    chrom = np.searchsorted(co, r0, side="right") - 1
    owner = lpt_assign(np.bincount(chrom, minlength=len(co) - 1), world)
    mine = owner[chrom] == rank
    kind = np.arange(n) >= n_roi
    tile_ptr = np.array([0, int((~kind[mine]).sum()), int(mine.sum())], np.int64)
    rows = [(int(co[k]), int(co[k + 1])) for k in range(len(co) - 1) if owner[k] == rank]
    return r0[mine], c0[mine], tile_ptr, rows, owner


# ----------------------------------------------------------------------------------------------------
# CPU baselines that fork (before any GPU runtime exists in this process)
# ----------------------------------------------------------------------------------------------------
_REF = {}


def _ref_algo_worker(bounds):
    lo, hi = bounds
    from oracle import pileup_oracle as po
    g = _REF
    t = time.perf_counter()
    acc = po.pileup_scipy(g["big"], g["lo"], g["lo"], g["w"], None, None, g["r0"][lo:hi], g["c0"][lo:hi], None,
                          g["tile"][lo:hi], 2, g["pad"], 2, 0)
    return time.perf_counter() - t, acc


def ref_algo_baseline(a, wl):
    """The reference's ALGORITHM on the host cores, timed the way the reference runs it: per region one symmetric
    scipy CSR (get_data), then per snippet CSR slice -> dense -> NaN rows/cols -> diagonal mask -> nansum / isfinite
    (oracle.pileup_scipy; the reference's quadratic list rebuild is not reproduced).  One region (the chromosome with
    the most snippets), a bounded snippet sample; 1 core, then the same sample split over C forked workers."""
    from oracle import pileup_oracle as po
    import multiprocessing as mp
    co = wl["chrom_offset"]
    r0, c0, n_roi = wl["r0_stream"], wl["c0_stream"], int(wl["n_roi"])
    chrom = np.searchsorted(co, r0, side="right") - 1
    k = int(np.bincount(chrom).argmax())
    lo, hi = int(co[k]), int(co[k + 1])
    sel = np.flatnonzero(chrom == k)
    m = min(a.ref_algo_sample, len(sel))
    sel = sel[np.linspace(0, len(sel) - 1, m).astype(np.int64)]
    t = time.perf_counter()
    big = po.symmetric_csr(wl["bin1_offset"], wl["bin2_id"], wl["count"], wl["weight"], lo, hi, lo, hi)
    t_csr = time.perf_counter() - t
    _REF.update(big=big, lo=lo, w=wl["weight"], r0=r0[sel], c0=c0[sel], tile=(sel >= n_roi).astype(np.int32), pad=a.pad)
    m1 = max(1, m // 8)
    t1, _ = _ref_algo_worker((0, m1))
    C_thr = max(1, min(a.cpu_threads if a.cpu_threads > 0 else (os.cpu_count() or 1), 64))
    cuts = np.linspace(0, m, C_thr + 1).astype(np.int64)
    ctx = mp.get_context("fork")
    t = time.perf_counter()
    with ctx.Pool(C_thr) as pool:
        parts = pool.map(_ref_algo_worker, [(int(cuts[i]), int(cuts[i + 1])) for i in range(C_thr)])
    t_all = time.perf_counter() - t
    acc = {kk: sum(p[1][kk] for p in parts) for kk in ("sum", "num", "n")}
    out = {"one_core_snippets_per_s": round(m1 / t1, 1), "all_cores_snippets_per_s": round(m / t_all, 1), "cores": C_thr,
           "sample": f"{m} snippets of the busiest chromosome (global bins {lo}..{hi}); 1 core on {m1} of them "
                     f"({t1:.1f}s), {C_thr} forked workers on all of them ({t_all:.1f}s incl. pool start); "
                     f"symmetric region CSR built once in {t_csr:.1f}s (not counted)",
           "what": "oracle.pileup_scipy: the reference's per-snippet operation sequence on a scipy CSR "
                   "(coolpuppy/coolpup.py:1104-1157 + lib/puputils.py:12-41)"}
    keep = {"sel": sel, "acc": acc}
    _REF.clear()
    return out, keep


def time_public_call(a, wl, r0, c0, n_set):
    """pileup() of the bench workload as a user calls it — BEDPE frame in, DataFrame out — on a table already resident
    (best of the calls after the first, which is reported separately), plus the H2D copy of the
    window coordinates alone (page-locked source, as the library's own staging arrays are).  NOT the headline value: the
    timed region of the benchmark starts with the coordinates in HBM."""
    import warnings
    import torch
    from coolpuppy_amd import coolpup
    from coolpuppy_amd.cooler_lite import ArrayCooler
    from coolpuppy_amd.engine import pinned_empty
    clr = ArrayCooler(_chromsizes(a), 10_000, wl["bin1_offset"], wl["bin2_id"], wl["count"],
                      bins={"weight": wl["weight"]}, filename="synthetic_hg38_10kb.cool")
    pairs = synth.random_cis_pairs(clr, a.pairs, min_sep=230_000, max_sep=5_000_000, seed=42)
    kw = dict(features_format="bedpe", flank=a.pad * clr.binsize, nshifts=a.nshifts, seed=0, mindist="auto")
    walls = []
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        for _ in range(4):
            t = time.perf_counter()
            df = coolpup.pileup(clr, pairs.copy(), **kw)
            walls.append(time.perf_counter() - t)
    # coordinates alone over PCIe
    p0, p1 = pinned_empty(len(r0)), pinned_empty(len(c0))
    p0[:], p1[:] = r0, c0
    d0 = torch.empty(len(r0), dtype=torch.int32, device="cuda"); d1 = torch.empty_like(d0)
    h2d = []
    for _ in range(5):
        torch.cuda.synchronize()
        t = time.perf_counter()
        d0.copy_(torch.from_numpy(p0), non_blocking=True); d1.copy_(torch.from_numpy(p1), non_blocking=True)
        torch.cuda.synchronize()
        h2d.append(time.perf_counter() - t)
    best = min(walls[1:])
    cool_file = None
    try:
        cool_file = time_cool_file(a, wl, (pairs, kw))
    except Exception as e:           # noqa: BLE001 - a secondary measurement must not cost the line
        cool_file = {"error": f"{type(e).__name__}: {e}"}
    return {"pileup_wall_s": round(best, 4), "cool_file": cool_file, "first_call_wall_s": round(walls[0], 3),
            "snippets_per_s": round(n_set / best, 1), "roi_windows_kept": int(df["n"].iloc[-1]),
            "h2d_coordinates_ms": round(min(h2d) * 1e3, 3), "coordinate_bytes": int(8 * len(r0)),
            "h2d_GBps": round(8 * len(r0) / min(h2d) / 1e9, 1),
            "note": "pileup(clr, bedpe_frame, flank, nshifts=10, seed=0) end to end on a resident table: host coordinate layer "
                    "(pandas sort of the pairs, the reference's 2 x 10^7 legacy-RNG draws, window generation), H2D of the "
                    "coordinates, device sort + pile-up, finaliser.  first_call_wall_s: the first such call of the process (uploads the pixel table and "
                    "builds the index unless an engine for the same table is already cached).  h2d_coordinates_ms: the two int32 coordinate arrays from page-locked memory"}


def time_cool_file(a, wl, clr_pairs_kw):
    """First pile-up of a process from a real .cool FILE (written here, once, under $TMPDIR): read_cool(stream_pixels=True) +
    pileup(): the pixel table goes file -> page-locked slabs -> HBM in chunks (pup_load_pixels_stream) while the host layer
    builds the windows.  Reports the wall of that first call and the copy rate of the streamed upload."""
    import warnings
    from coolpuppy_amd import cool_io, coolpup
    from coolpuppy_amd.cooler_lite import ArrayCooler
    path = _tmp("coolpuppy_amd_bench_" + hashlib.sha1(f"f1|{a.chroms}|{a.lam}".encode()).hexdigest()[:12] + ".cool")
    t_write = None
    if not os.path.exists(path):
        clr = ArrayCooler(_chromsizes(a), 10_000, wl["bin1_offset"], wl["bin2_id"], wl["count"], bins={"weight": wl["weight"]},
                          filename="synthetic_hg38_10kb.cool")
        t = time.perf_counter()
        cool_io.write_cool(path + ".tmp", clr)
        os.replace(path + ".tmp", path)
        t_write = time.perf_counter() - t
    pairs, kw = clr_pairs_kw
    for k in list(coolpup._ENGINES):                       # the bench's resident engine must not be mistaken for this table's
        coolpup._ENGINES.pop(k)[1].close()
    t = time.perf_counter()
    lazy = cool_io.read_cool(path, stream_pixels=True)
    t_open = time.perf_counter() - t
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        t = time.perf_counter()
        df = coolpup.pileup(lazy, pairs.copy(), **kw)
        first = time.perf_counter() - t
        t = time.perf_counter()
        coolpup.pileup(lazy, pairs.copy(), **kw)
        second = time.perf_counter() - t
    st = getattr(lazy, "_last_stream_stats", None) or {}
    return {"file": os.path.basename(path), "file_bytes": os.path.getsize(path), "write_s": None if t_write is None else round(t_write, 2),
            "open_s": round(t_open, 3), "first_pileup_wall_s": round(first, 3), "second_pileup_wall_s": round(second, 3),
            "pixels_in_host_memory": bool(lazy.pixels_in_memory), "roi_windows_kept": int(df["n"].iloc[-1]),
            "h2d_pixel_table_ms": None if not st else round(st["h2d_ms"], 2), "h2d_pixel_table_bytes": st.get("h2d_bytes"),
            "h2d_pixel_table_GBps": None if not st or not st.get("h2d_GBps") else round(st["h2d_GBps"], 1),
            "note": "read_cool(path, stream_pixels=True) then pileup(): pixels/bin2_id and pixels/count leave the file by hyperslab reads "
                    "into two page-locked slabs of the library and travel by hipMemcpyAsync on a copy stream, slab k + 1 being read while "
                    "slab k is in flight; h2d_* = device time and bytes of those copies alone (HIP events)"}


def lpt_assign(costs, world):
    """Longest-processing-time assignment of items to ranks (deterministic)."""
    load = np.zeros(world)
    owner = np.zeros(len(costs), np.int64)
    for i in np.argsort(-np.asarray(costs), kind="stable"):
        r = int(np.argmin(load))
        owner[i] = r
        load[r] += costs[i]
    return owner


# ----------------------------------------------------------------------------------------------------
def main():
    a = parse()
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus:
        if world == 1 and a.gpus > 1:
            sys.exit("bench.py --gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        a.gpus = world
    if world == 1:
        a.scaling = "strong"                  # N=1: the whole workload either way
    elif a.scaling == "auto":
        a.scaling = "weak"                    # N>1: per-GPU work as at N=1 (the strong reading of the 1e6-pair job rides along)
    if a.config != 2:
        import bench_plan
        return bench_plan.main_plan(a, rank, world, local_rank, sys.modules[__name__])

    wl = load_workload(a, rank, world)
    ref_algo = ref_keep = None
    if rank == 0 and a.gpus == 1 and a.cpu_sample > 0 and a.ref_algo_sample > 0:
        ref_algo, ref_keep = ref_algo_baseline(a, wl)

    import torch
    import torch.distributed as dist
    from coolpuppy_amd.build import build_hip
    from coolpuppy_amd.engine import PileupEngine, MODE_DEVPTR  # noqa: F401

    if rank == 0:
        build_hip()          # no-op when the in-tree .so is current
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (torch.cuda.is_available() is False)")
    if os.environ.get("COOLPUPPY_AMD_BENCH_DEVICE", "") != "":
        local_rank = int(os.environ["COOLPUPPY_AMD_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)

    def allreduce(t, op=None):
        """all-reduce a CUDA tensor in place (through host memory when the backend is gloo)."""
        kw = {} if op is None else {"op": op}
        if a.backend == "nccl":
            dist.all_reduce(t, **kw)
        else:
            h = t.cpu()
            dist.all_reduce(h, **kw)
            t.copy_(h)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    W = 2 * a.pad + 1
    r0_all, c0_all = wl["r0_stream"], wl["c0_stream"]
    n_set = int(r0_all.shape[0])
    n_roi = int(wl["n_roi"])
    co = wl["chrom_offset"]
    # strong: the fixed workload, chromosomes -> ranks by LPT on their snippet counts; a rank piles up only its
    # chromosomes' snippets and holds only their rows of the pixel table (what PileUpper.run_plan uploads per rank)
    r0, c0, tile_ptr, rows, owner = shard_snippets(r0_all, c0_all, n_roi, co, rank, world)
    sharding = ("single GPU" if world == 1 else
                "chromosomes sharded over the ranks (LPT on snippet counts); every rank holds the pixel rows of its own chromosomes")
    n_local = int(tile_ptr[-1])

    eng = PileupEngine(local_rank)
    t_h2d = time.time()
    if world == 1:
        eng.load_pixels(wl["bin1_offset"], wl["bin2_id"], wl["count"])
    else:
        from coolpuppy_amd.coolpup import _rows_of_table
        eng.load_pixels(*_rows_of_table(wl["bin1_offset"], wl["bin2_id"], wl["count"], rows))
    eng.load_bins(wl["weight"], None)
    eng.sync()
    t_h2d = time.time() - t_h2d
    t_idx = time.time()
    have_index = eng.build_index(co) if not a.no_index else False
    t_idx = time.time() - t_idx
    eng.set_tuning(0, a.variant)
    eng.reset(2, a.pad)
    nf, ni = eng.packed_sizes()
    buf_f = torch.zeros(nf, dtype=torch.float64, device="cuda")
    buf_i = torch.zeros(ni, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()

    native_comm, rccl_ranks, exchange_used, native_failed = None, None, "none", False
    if world > 1:
        exchange_used = "torch.distributed.all_reduce on exported buffers"
        if a.exchange == "native" and a.backend == "nccl":
            from coolpuppy_amd import dist as pdist
            try:
                native_comm = pdist.native_comm(eng)       # every rank gets a communicator, or every rank gets None
            except (RuntimeError, OSError) as e:
                print(f"[bench] rank {rank}: engine-side RCCL unavailable ({e}); torch.distributed.all_reduce instead",
                      file=sys.stderr, flush=True)
            if native_comm is not None:
                rccl_ranks = pdist.comm_ranks(native_comm)
                exchange_used = "pup_allreduce (RCCL on the engine's stream, in place)"
            verdict = exchange_verdict(world, rccl_ranks, a.strict_exchange, rank)
            if verdict["message"]:
                print(verdict["message"], file=sys.stderr, flush=True)
            if verdict["fallback"]:
                native_comm = None
                native_failed = True
                exchange_used = verdict["exchange"]

    def make_step(p_r0, p_c0, n, tptr):
        def step():
            eng.reset(2, a.pad)
            if n > 0:
                eng.accumulate_device(p_r0, p_c0, n, tptr, ignore_diags=2, mode=0)
            if world > 1 and native_comm is not None:
                eng.allreduce(native_comm)               # asynchronous, ordered on the engine's stream
            elif world > 1:
                eng.export_to(buf_f.data_ptr(), buf_i.data_ptr())
                allreduce(buf_f)
                allreduce(buf_i)
                torch.cuda.synchronize()
                eng.import_from(buf_f.data_ptr(), buf_i.data_ptr())
            # (no synchronisation here: a step is ordered on the engine's stream, the timed region is bracketed by
            # barrier + torch.cuda.synchronize() — the library never lets the host run more than one call ahead)
        return step

    def timed(stepf, steps, warmup):
        for _ in range(warmup):
            stepf()
        barrier()
        t0 = time.perf_counter()
        for _ in range(steps):
            stepf()
        barrier()
        return time.perf_counter() - t0

    d_r0 = torch.from_numpy(np.ascontiguousarray(r0)).cuda()
    d_c0 = torch.from_numpy(np.ascontiguousarray(c0)).cuda()
    step = make_step(d_r0.data_ptr(), d_c0.data_ptr(), n_local, tile_ptr)
    # pixel statistics (nnz per window, for the algorithmic byte count) are gathered once, outside the timed region
    eng.set_profiling(1)
    eng.clear_stats()
    # the first step with a collective in it runs under a watchdog: an exchange that never completes (the engine-side RCCL
    # path has only ever run on one-GPU boxes before the driver's scaling run) must end the job with a message, not hang it
    watchdog = None
    if world > 1:
        import threading

        def _give_up():
            print(f"[bench] rank {rank}: the first step did not complete within 180 s (exchange: {exchange_used}); "
                  "re-run with --exchange torch", file=sys.stderr, flush=True)
            os._exit(3)
        watchdog = threading.Timer(180.0, _give_up)
        watchdog.daemon = True
        watchdog.start()
    step()
    eng.sync()
    if watchdog is not None:
        barrier()
        watchdog.cancel()
    st0 = eng.stats()
    pix_per_step_local = float(st0["pixels_in_windows"])
    eng.set_profiling(0)
    for _ in range(a.warmup):
        step()
    eng.set_profiling(3)          # HIP events around the kernels, no pixel counting inside the timed kernels
    eng.clear_stats()
    dt = timed(step, a.steps, 0)
    st = eng.stats()
    staged_regions = int(st.get("staged_regions", 0))
    eng.set_profiling(0)
    out = eng.fetch()

    # ---- weak scaling of the same sharded path: world x a.pairs pairs (N>1 only) ---------------------------------------
    weak = None
    if world > 1:
        wr0, wc0, wtp, wrows, wowner = shard_snippets(wl["weak_r0_stream"], wl["weak_c0_stream"], int(wl["weak_n_roi"]), co,
                                                      rank, world)
        if not np.array_equal(wowner, owner):            # another pair set may shard differently: this rank's rows change
            from coolpuppy_amd.coolpup import _rows_of_table
            eng.load_pixels(*_rows_of_table(wl["bin1_offset"], wl["bin2_id"], wl["count"], wrows))
            eng.load_bins(wl["weight"], None)
            eng.build_index(co)
            eng.set_tuning(0, a.variant)
        w_r0 = torch.from_numpy(np.ascontiguousarray(wr0)).cuda()
        w_c0 = torch.from_numpy(np.ascontiguousarray(wc0)).cuda()
        wstep = make_step(w_r0.data_ptr(), w_c0.data_ptr(), int(wtp[-1]), wtp)
        wsteps = a.steps if a.scaling == "weak" else max(10, min(a.steps, 50))
        wdt = timed(wstep, wsteps, max(2, a.warmup))
        wn = torch.tensor([float(wtp[-1])], dtype=torch.float64, device="cuda")
        allreduce(wn)
        wt = torch.tensor([wdt], dtype=torch.float64, device="cuda")
        allreduce(wt, dist.ReduceOp.MAX)
        wout = eng.fetch()
        weak = {"scaling": "weak", "pairs": a.pairs * world, "snippets_per_step": int(wn[0].item()), "steps": wsteps,
                "ms_per_step": round(float(wt[0].item()) / wsteps * 1e3, 4),
                "value": round(wn[0].item() * wsteps / float(wt[0].item()), 1),
                "check_n": [int(x) for x in wout["n"]],
                "note": "the same sharded path on world x pairs pairs: per-GPU work stays that of the N=1 run"}
        del w_r0, w_c0

    n_all = n_local
    if world > 1:
        nn = torch.tensor([float(n_local), pix_per_step_local], dtype=torch.float64, device="cuda")
        allreduce(nn)
        n_all, pix_per_step = int(nn[0].item()), float(nn[1].item())
        tt = torch.tensor([dt, st["k1_ms"]], dtype=torch.float64, device="cuda")
        allreduce(tt, dist.ReduceOp.MAX)
        dt, k1_ms_max = float(tt[0].item()), float(tt[1].item())
    else:
        k1_ms_max, pix_per_step = st["k1_ms"], pix_per_step_local

    # ---- secondary: the same snippets pre-sorted into the engine's block order (the device sort finds them sorted) ---
    preblocked = None
    if world == 1:
        order = PileupEngine.block_order(r0, c0, co, tile=(np.arange(n_local) >= n_roi), pad=a.pad)
        p_r0 = torch.from_numpy(np.ascontiguousarray(r0[order])).cuda()
        p_c0 = torch.from_numpy(np.ascontiguousarray(c0[order])).cuda()
        pstep = make_step(p_r0.data_ptr(), p_c0.data_ptr(), n_local, tile_ptr)
        psteps = max(10, min(a.steps, 50))
        pdt = timed(pstep, psteps, 3)
        preblocked = {"ms_per_step": round(pdt / psteps * 1e3, 4), "snippets_per_s": round(n_local * psteps / pdt, 1),
                      "note": "same snippets handed over already in the staged kernel's block order: the device sort still runs (every "
                              "call does it) but its gather reads sequentially"}
        del p_r0, p_c0

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = n_all * a.steps / dt
        # ---- roofline of the pile-up kernels, per launch ------------------------------------------------
        launches = max(int(st["k1_launches"]), 1)
        k1_ms = k1_ms_max / launches                                            # slowest rank
        # (1) SURVEY 8(d) algorithmic bytes: every window charged its own bytes
        alg_bytes = n_all * (8 * (W + 1) + 16 * W + 12) + 8.0 * pix_per_step
        achieved = alg_bytes / (k1_ms * 1e-3) / 1e9
        # (2) compulsory bytes: what one pass must read whatever the kernel — the pixels' values (8 B, `bal`) and the
        #     rank-bitmap index lines (64 B) under the windows' row hulls, the snippet coordinates (8 B each)
        tp = wl["touched"]
        compulsory = float(tp[0]) * 8 + float(tp[1]) * 64 + 8.0 * n_set
        #     ... and the looser "one pass over the resident tables" (every pixel value + the whole index + coordinates)
        nnz_all, nb_all = float(wl["bin2_id"].shape[0]), float(wl["bin1_offset"].shape[0] - 1)
        idx_lines_all = float(sum((int(co[k + 1] - co[k]) * -(-int(co[k + 1] - co[k]) // 320)) for k in range(len(co) - 1)))
        table_pass = nnz_all * 8 + idx_lines_all * 64 + 8.0 * n_set
        # (3) measured HBM bytes (rocprofv3 PMC passes over this command; quoted only for these kernel sources)
        traffic, traffic_src = measured_traffic(a)
        peak = HBM_PEAK_GBPS * a.gpus
        frac_comp = compulsory / (k1_ms * 1e-3) / 1e9 / peak if a.scaling != "weak" or a.gpus == 1 else None
        frac_traffic = None if traffic is None else traffic / (k1_ms * 1e-3) / 1e9 / peak
        frac_table = table_pass / (k1_ms * 1e-3) / 1e9 / peak if (a.scaling != "weak" or a.gpus == 1) else None
        peak_measured = None
        try:
            peak_measured = json.load(open(os.path.join(ROOT, "profiles", "hbm_peak.json")))
        except Exception:
            pass
        staged = staged_regions > 0
        family = eng.last_kernel()                       # which kernel family the timed calls ran (pup_last_kernel)
        reg_rows, reg_cols = PileupEngine.staged_region(a.pad)
        if a.variant & 128:
            reg_rows = 72
        if family.startswith("wide"):
            reg_rows = reg_cols = 128
            wg = PileupEngine.wide_geometry(W)
            kernel_name = (f"pup::pileup_wide_kernel<{wg['CH']}, {wg['NCH']}, false, {'true' if family == 'wide_fact' else 'false'}> (K1w: persistent 16-wave "
                           f"workgroups, 128 x 128 regions staged in LDS from the dense band; {wg['NGr']} x {wg['NGc']} sub-windows of "
                           f"{wg['SH']} x {wg['SW']} bins on the 256 lanes of four waves: {wg['NCH']} column chunks per row, {wg['CH']} cells per lane)")
        elif family == "staged":
            kernel_name = (f"pup::pileup_staged_kernel<{W}, false, {reg_rows}, {reg_cols}, {reg_rows // 8}, {1 if a.variant & 64 else 2}, "
                           f"{'false' if a.variant & 4 else 'true'}, false, {'false' if a.variant & (1 << 27) else 'true'}> (persistent workgroups, "
                           f"{reg_rows} x {reg_cols} regions staged in LDS, ROI and control tile in one pass)")
        else:
            kernel_name = (f"pup::pileup_regtile_kernel<{W}, false>" if family == "regtile" else
                           (f"pup::pileup_band_kernel<{4 if W <= 64 else (8 if W <= 128 else 16)}, false>" if family == "band" else family))
        # the kernel's own bound when it piles up from LDS-staged regions: every window reads its W x W cells (f64) from LDS and
        # adds them with f64 VALU instructions; every region is written to LDS once
        lds_bytes = (n_all // a.gpus) * W * W * 8 + staged_regions * reg_rows * reg_cols * 8
        lds_peak = 256 * 256 * 2.4                         # CUs x B/clk/CU (ds_read_b64, MI355X_MICROARCH.md) x GHz = GB/s
        lds_frac = lds_bytes / (k1_ms * 1e-3) / 1e9 / lds_peak
        valu_f64_frac = ((n_all // a.gpus) * W * W / 64.0 * 4.0) / (1024 * 2.4e9 * k1_ms * 1e-3)    # v_add_f64 wave instructions x 4 clk / SIMD clocks
        roofline = {
            # what binds the dominant kernel: the staged kernels read 0.9 x the compulsory HBM bytes once and are LDS / f64-issue
            # bound; the per-window kernels are bound by what they pull from L2 / HBM
            "bound": "lds" if staged else "hbm",
            "kernel": kernel_name,
            "kernel_family": family,
            "achieved": round(achieved, 1), "peak": peak, "unit": "GB/s",
            # the HBM fraction: MEASURED bytes (rocprofv3 PMC, profiles/traffic.json) / kernel time / peak when a measurement of
            # these kernel sources and this workload is on file, else COMPULSORY bytes (what any kernel must read once).  (The
            # contract's A/P with the SURVEY 8(d) per-window bytes is algorithmic_over_peak: > 1 by construction for a kernel
            # that serves hundreds of overlapping windows from one staged region.)
            "frac": (None if frac_comp is None else round(frac_comp, 4)) if frac_traffic is None else round(frac_traffic, 4),
            "frac_is": "frac_compulsory" if frac_traffic is None else "frac_traffic",
            "lds_frac": round(lds_frac, 4) if staged else None,
            "lds_achieved_GBps": round(lds_bytes / (k1_ms * 1e-3) / 1e9, 1) if staged else None,
            "lds_peak_GBps": round(lds_peak, 1),
            "valu_f64_frac": round(valu_f64_frac, 4) if staged else None,
            "frac_compulsory": None if frac_comp is None else round(frac_comp, 4),
            "frac_table_pass": None if frac_table is None else round(frac_table, 4),
            "frac_traffic": None if frac_traffic is None else round(frac_traffic, 4),
            "algorithmic_over_peak": round(achieved / peak, 4),
            "traffic": traffic, "traffic_source": traffic_src,
            "kernel_ms_per_launch": round(k1_ms, 4),
            "algorithmic_bytes_per_launch": round(alg_bytes / a.gpus),
            "compulsory_bytes_per_launch": round(compulsory),
            "table_pass_bytes_per_launch": round(table_pass),
            "compulsory": {"pixels_under_row_hulls": int(tp[0]), "index_lines": int(tp[1]), "rows": int(tp[2]),
                           "coordinate_bytes": 8 * n_set},
            "nnz_win_mean": round(pix_per_step / max(n_all, 1), 1),
            "note": ("achieved = SURVEY 8(d) algorithmic bytes / kernel time: every window is charged its own bytes, so a "
                     "kernel that serves many windows from one LDS-staged region can exceed the HBM peak on it "
                     "(algorithmic_over_peak is NOT a roofline fraction).  frac = measured HBM bytes (rocprofv3 PMC, "
                     "profiles/) / kernel time / peak when a measurement of these kernel sources is on file, else "
                     "compulsory bytes / kernel time / peak.  compulsory = bytes of the resident tables under the windows' "
                     "row hulls (what any kernel must read once); table_pass = every pixel value + the whole index once"),
            "peak_measured_GBps": (None if peak_measured is None else
                                   {k: peak_measured[k] for k in ("read_GBps", "copy_GBps", "triad_GBps")}),
            "staged_regions_per_launch": staged_regions,
            "prepass_ms_per_launch": round(st.get("prepare_ms", 0.0) / launches, 4),
        }
        if staged:
            roofline["lds"] = {"bytes_per_launch": int(lds_bytes), "achieved": round(lds_bytes / (k1_ms * 1e-3) / 1e9, 1),
                           "peak": round(lds_peak, 1), "unit": "GB/s",
                           "frac": round(lds_bytes / (k1_ms * 1e-3) / 1e9 / lds_peak, 4),
                           "note": "the kernel's binding resource together with f64 VALU issue; peak = 256 CUs x 256 B/clk x 2.4 GHz"}
        # ---- CPU baselines + same-run parity on bounded samples (N=1 only) ------------------------------
        cpu = None
        if a.gpus == 1 and a.cpu_sample > 0:
            from oracle import pileup_oracle as po
            po.build()
            m = min(a.cpu_sample, n_set)
            idx = np.linspace(0, n_set - 1, m).astype(np.int64)
            sr0, sc0 = r0_all[idx], c0_all[idx]
            stile = (idx >= n_roi).astype(np.int32)
            arrs = [np.ascontiguousarray(x) for x in (wl["bin1_offset"], wl["bin2_id"], wl["count"], wl["weight"])]
            C_thr = max(1, min(a.cpu_threads if a.cpu_threads > 0 else (os.cpu_count() or 1), 64))
            # (1) the scalar per-cell port (the test oracle as is), one core, on 1/16 of the sample
            m1 = max(1, m // 16)
            t = time.perf_counter()
            po.pileup_c(arrs[0], arrs[1], arrs[2], arrs[3], None, None, sr0[:m1], sc0[:m1], None, stile[:m1], 2, a.pad, 2, 0)
            percell_one = m1 / (time.perf_counter() - t)
            # (2) "best CPU": row-sliced windows, one core and C OpenMP threads
            m2 = max(1, m // 8)
            t = time.perf_counter()
            po.pileup_c_mt(arrs[0], arrs[1], arrs[2], arrs[3], None, None, sr0[:m2], sc0[:m2], None, stile[:m2], 2,
                           a.pad, 2, 0, 1)
            rows_one = m2 / (time.perf_counter() - t)
            t = time.perf_counter()
            want = po.pileup_c_mt(arrs[0], arrs[1], arrs[2], arrs[3], None, None, sr0, sc0, None, stile, 2, a.pad, 2, 0,
                                  C_thr)
            cpu_s = time.perf_counter() - t
            sptr = np.array([0, int((stile == 0).sum()), m], np.int64)
            eng.reset(2, a.pad)
            eng.accumulate(sr0, sc0, sptr, ignore_diags=2, mode=0)
            got = eng.fetch()
            ok = (np.array_equal(got["n"], want["n"]) and np.array_equal(got["num"], want["num"])
                  and np.allclose(got["sum"], want["sum"], rtol=1e-6, atol=0, equal_nan=True))
            ok_ref = None
            if ref_keep is not None:
                sel = ref_keep["sel"]
                rt = (sel >= n_roi).astype(np.int64)
                o2 = np.argsort(rt, kind="stable")
                eng.reset(2, a.pad)
                eng.accumulate(r0_all[sel][o2], c0_all[sel][o2], np.array([0, int((rt == 0).sum()), len(sel)], np.int64),
                               ignore_diags=2, mode=0)
                g2, w2 = eng.fetch(), ref_keep["acc"]
                ok_ref = bool(np.array_equal(g2["n"], w2["n"]) and np.array_equal(g2["num"], w2["num"])
                              and np.allclose(g2["sum"], w2["sum"], rtol=1e-6, atol=0, equal_nan=True))
            cpu = {"value": round(m / cpu_s, 1), "unit": "snippets/s", "cores": C_thr, "kind": "port",
                   "sample": f"{m} snippets strided over the {n_set} of this workload; best-CPU form of the C oracle "
                             f"(oracle/pileup_oracle.c: row-sliced windows, {C_thr} OpenMP threads, private accumulators), "
                             f"{cpu_s:.1f}s; one thread: {rows_one:.0f} snippets/s; per-cell-bisection form (the test "
                             f"checker), one thread: {percell_one:.0f} snippets/s",
                   "single_core_value": round(rows_one, 1), "percell_single_core_value": round(percell_one, 1),
                   "host_cpu_count": os.cpu_count(),
                   "cpu_ref_algo": ref_algo,
                   "gpu_matches_oracle_on_sample": bool(ok), "gpu_matches_ref_algo_on_sample": ok_ref}
            if not ok or ok_ref is False:
                print("[bench] PARITY FAILURE against the oracle on the sample", file=sys.stderr)
        # ---- the same workload through the PUBLIC call: features -> coordinates -> H2D -> pile-up -> finaliser (N=1) ----
        end_to_end = None
        if a.gpus == 1 and not a.no_end_to_end:
            end_to_end = time_public_call(a, wl, r0, c0, n_set)
        strong = {"scaling": "strong", "pairs": a.pairs, "snippets_per_step": n_all, "steps": a.steps,
                  "ms_per_step": round(ms_per_step, 4), "value": round(value, 1)}
        primary = weak if (a.scaling == "weak" and weak is not None) else strong
        line = {
            "metric": f"snippets/sec ({W}x{W} windows @10kb, ROI + control snippets accumulated)",
            "value": primary["value"], "unit": "snippets/s", "n_gpus": a.gpus, "steps": primary["steps"], "warmup": a.warmup,
            "ms_per_step": primary["ms_per_step"], "higher_is_better": True, "scaling": primary["scaling"],
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": ("BASELINE configs[2]: " if (a.pad == 10 and a.pairs == 1_000_000 and a.nshifts == 10) else
                             "BASELINE configs[2] table with other windows: ") +
                            f"synthetic hg38 10kb CSR + {a.pairs:.0e} random cis BEDPE pairs, pad={a.pad}, "
                            f"nshifts={a.nshifts}, balanced, ignore_diags=2",
                "order": "reference stream",
                "nnz": int(wl["bin2_id"].shape[0]), "nbins": int(wl["bin1_offset"].shape[0] - 1),
                "pairs": primary["pairs"], "nshifts": a.nshifts, "pad": a.pad, "snippets_per_step": primary["snippets_per_step"],
                "parallelism": f"{a.gpus} rank(s); {sharding}; one all-reduce of the packed tiles every step (N>1): {exchange_used}",
                "variant": a.variant,
            },
            "exchange": exchange_used, "rccl_ranks": rccl_ranks, "native_exchange_failed": native_failed,
            "strong": strong if a.gpus > 1 else None, "weak": weak,
            "roofline": roofline, "cpu_baseline": cpu, "preblocked": preblocked, "end_to_end": end_to_end,
            "h2d_pixel_table_s": round(t_h2d, 3), "rank_bitmap_index": bool(have_index),
            "index_build_s": round(t_idx, 3),
            "check": {"n": [int(x) for x in out["n"]],
                      "center_roi_over_ctrl": float((out["sum"][0, a.pad, a.pad] / out["num"][0, a.pad, a.pad]) /
                                                    (out["sum"][1, a.pad, a.pad] / out["num"][1, a.pad, a.pad]))},
        }
        print(json.dumps(line), flush=True)
    eng.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
