#!/usr/bin/env python3
"""bench.py — pile-up hot path on N MI355X GPUs (one process per GPU).

Workload (BASELINE.json configs[2], the configuration the north-star target is quoted on):
synthetic human-scale 10 kb pixel table (hg38 chr1-22,X, ~3e8 upper-triangular nnz, balanced with
2 % masked bins) + 1e6 random cis BEDPE pairs, pad=10 (21x21 windows), nshifts=10 random-shift
controls (seed 0) => ~1.1e7 snippets per step.  A "step" = one full pass: zero the accumulators,
pile up every snippet (ROI + controls) of this rank's shard from HBM-resident inputs, and (N>1)
all-reduce the packed sum/num/n/cov accumulators over RCCL.

Scaling (default WEAK, per-GPU work fixed): the pixel table is replicated on every GPU (2.7 GB of 288 GB) and
every rank piles up its OWN set of 1e6 pairs (pair seed 42+rank, control seed rank); the step ends with the real
exchange of the path — one all-reduce of the packed tiles — so the job's result is the pile-up over all N sets.
`--scaling strong` splits ONE 1e6-pair set over the ranks instead (contiguous slices of the sorted snippets).

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (K1 kernel,
HIP-event timed inside this process) and `cpu_baseline` (the C oracle on a bounded sample, N=1 only).
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

from coolpuppy_amd import synth  # noqa: E402  (numpy only; torch is imported after generation)

HBM_PEAK_GBPS = 8000.0   # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (~6.3 TB/s achievable)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--pairs", type=int, default=1_000_000)
    ap.add_argument("--nshifts", type=int, default=10)
    ap.add_argument("--pad", type=int, default=10)
    ap.add_argument("--lam", type=float, default=4200.0, help="Poisson contacts drawn per row before de-duplication")
    ap.add_argument("--chroms", type=int, default=23, help="use the first K hg38 chromosomes (23 = all)")
    ap.add_argument("--cpu-sample", type=int, default=4_000_000, help="snippets timed on the CPU oracle (0 = skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the CPU baseline (0 = all cores, capped at 64)")
    ap.add_argument("--no-cache", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="weak: every rank piles up its own set of --pairs pairs on the shared table; "
                         "strong: the ranks split ONE set")
    ap.add_argument("--no-index", action="store_true", help="binary search only (skip the rank-bitmap index)")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="collective backend; gloo (+ COOLPUPPY_AMD_BENCH_DEVICE=0) lets several ranks share ONE GPU "
                         "to smoke-test the N>1 code path on a single-GPU box")
    return ap.parse_args()


# ----------------------------------------------------------------------------------------------------
# workload (cached under $TMPDIR so that back-to-back runs at N = 1, 2, 4, 8 do not regenerate it)
# ----------------------------------------------------------------------------------------------------
def _tmp(name):
    return os.path.join(os.environ.get("TMPDIR", "/tmp"), name)


def workload_key(a):
    return hashlib.sha1(f"w5|{a.chroms}|{a.lam}|{a.pairs}|{a.nshifts}|{a.pad}".encode()).hexdigest()[:12]


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


def build_snippets(a, cool, k):
    """Snippet set k: a.pairs random cis BEDPE pairs (pair seed 42+k) through the host-side coordinate layer
    (CoordCreator semantics, control-shift RNG seed k) -> block-ordered (r0, c0) with ROI first."""
    from coolpuppy_amd.cooler_lite import ArrayCooler
    from coolpuppy_amd.coolpup import CoordCreator, snippet_batches
    clr = ArrayCooler(_chromsizes(a), 10_000, cool["bin1_offset"], cool["bin2_id"], cool["count"],
                      bins={"weight": cool["weight"]}, filename="synthetic_hg38_10kb.cool")
    pairs = synth.random_cis_pairs(clr, a.pairs, min_sep=230_000, max_sep=5_000_000, seed=42 + k)
    np.random.seed(k)
    cc = CoordCreator(pairs, clr.binsize, features_format="bedpe", flank=a.pad * clr.binsize,
                      nshifts=a.nshifts, mindist="auto")
    r0, c0, kind = snippet_batches(cc, clr, control=a.nshifts > 0)
    # resident order = the engine's preferred input layout: ROI tile first, then inside each tile by 16 x 16 block of
    # top-left corners (block row, block column), then position — see pup_set_tuning in include/pup_hip.h
    from coolpuppy_amd.engine import PileupEngine
    order = PileupEngine.block_order(r0, c0, clr.chrom_offset, tile=kind)
    return {"r0": r0[order].astype(np.int32), "c0": c0[order].astype(np.int32), "n_roi": np.int64((kind == 0).sum())}


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
    k = rank if a.scaling == "weak" else 0
    spath = snippets_path(a, k)
    if (rank == k or a.scaling == "weak") and (a.no_cache and rank == k or not os.path.exists(spath)):
        t = time.time()
        _save(spath, **build_snippets(a, cool, k))
        print(f"[bench] rank {rank}: snippet set {k} built in {time.time()-t:.1f}s", file=sys.stderr, flush=True)
    _wait_for(spath)
    z = np.load(spath)
    cool.update({kk: z[kk] for kk in z.files})
    return cool


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

    wl = load_workload(a, rank, world)

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
    n_set = int(wl["r0"].shape[0])
    n_roi = int(wl["n_roi"])
    if a.scaling == "weak":
        # every rank owns a full workload (its own pairs / control shifts) on the replicated table
        r0, c0 = wl["r0"], wl["c0"]
        tile_ptr = np.array([0, n_roi, n_set], np.int64)
    else:
        # strong: contiguous slice of each tile's (sorted) snippet range of the one shared set
        def part(lo, hi):
            m = hi - lo
            return lo + (m * rank) // world, lo + (m * (rank + 1)) // world
        a0, a1 = part(0, n_roi)
        b0, b1 = part(n_roi, n_set)
        r0 = np.concatenate([wl["r0"][a0:a1], wl["r0"][b0:b1]])
        c0 = np.concatenate([wl["c0"][a0:a1], wl["c0"][b0:b1]])
        tile_ptr = np.array([0, a1 - a0, (a1 - a0) + (b1 - b0)], np.int64)
    n_local = int(tile_ptr[-1])

    eng = PileupEngine(local_rank)
    t_h2d = time.time()
    eng.load_pixels(wl["bin1_offset"], wl["bin2_id"], wl["count"])
    eng.load_bins(wl["weight"], None)
    eng.sync()
    t_h2d = time.time() - t_h2d
    t_idx = time.time()
    have_index = eng.build_index(wl["chrom_offset"]) if not a.no_index else False
    t_idx = time.time() - t_idx
    eng.reset(2, a.pad)
    d_r0 = torch.from_numpy(r0).cuda()
    d_c0 = torch.from_numpy(c0).cuda()
    nf, ni = eng.packed_sizes()
    buf_f = torch.zeros(nf, dtype=torch.float64, device="cuda")
    buf_i = torch.zeros(ni, dtype=torch.int64, device="cuda")
    torch.cuda.synchronize()

    def step():
        eng.reset(2, a.pad)
        eng.accumulate_device(d_r0.data_ptr(), d_c0.data_ptr(), n_local, tile_ptr, ignore_diags=2, mode=0)
        if world > 1:
            eng.export_to(buf_f.data_ptr(), buf_i.data_ptr())
            allreduce(buf_f)
            allreduce(buf_i)
            torch.cuda.synchronize()
            eng.import_from(buf_f.data_ptr(), buf_i.data_ptr())
        else:
            eng.sync()

    for _ in range(a.warmup):
        step()
    eng.set_profiling(True)
    eng.clear_stats()
    barrier()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    barrier()
    dt = time.perf_counter() - t0
    st = eng.stats()
    eng.set_profiling(False)

    n_all = n_local
    if world > 1:
        nn = torch.tensor([float(n_local)], dtype=torch.float64, device="cuda")
        allreduce(nn)
        n_all = int(nn.item())
        tt = torch.tensor([dt], dtype=torch.float64, device="cuda")
        allreduce(tt, dist.ReduceOp.MAX)
        dt = float(tt.item())
        agg = torch.tensor([st["k1_ms"], float(st["pixels_in_windows"]), float(st["snippets"])],
                           dtype=torch.float64, device="cuda")
        mx = agg.clone()
        allreduce(agg)                            # sums over ranks
        allreduce(mx, dist.ReduceOp.MAX)
        k1_ms_max = float(mx[0].item())
        pix_total, snip_total = float(agg[1].item()), float(agg[2].item())
    else:
        k1_ms_max = st["k1_ms"]
        pix_total, snip_total = float(st["pixels_in_windows"]), float(st["snippets"])

    out = eng.fetch()

    if rank == 0:
        ms_per_step = dt / a.steps * 1e3
        value = n_all * a.steps / dt
        # ---- roofline of the dominant kernel (K1), per launch -----------------------------------------
        # algorithmic bytes per snippet: 8(W+1) indptr + 8*nnz_win pixels + 16W weights + 12
        launches = max(int(st["k1_launches"]), 1)
        alg_bytes_total = snip_total * (8 * (W + 1) + 16 * W + 12) + 8.0 * pix_total     # all ranks, all steps
        k1_ms_per_launch = k1_ms_max / launches                                           # slowest rank
        achieved = alg_bytes_total / a.steps / (k1_ms_per_launch * 1e-3) / 1e9             # GB/s, whole job
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload_key") == workload_key(a) and a.gpus == 1:
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        peak_measured = None
        try:
            peak_measured = json.load(open(os.path.join(ROOT, "profiles", "hbm_peak.json")))
        except Exception:
            pass
        roofline = {
            "bound": "hbm",
            "kernel": ((f"pup::pileup_tiled_kernel<{W}, false, 16, 16> (dense tile) + pup::pileup_regtile_kernel<{W}, false> "
                        "(sparse tile), side by side on two streams, one launch each per step" if st.get("staged_regions", 0) > 0
                        else f"pup::pileup_regtile_kernel<{W}, false>") if W <= 31 else f"pup::pileup_band_kernel"),
            "achieved": round(achieved, 1),
            "peak": HBM_PEAK_GBPS * a.gpus, "unit": "GB/s", "frac": round(achieved / (HBM_PEAK_GBPS * a.gpus), 4),
            "traffic": traffic, "kernel_ms_per_launch": round(k1_ms_per_launch, 4),
            "algorithmic_bytes_per_launch": round(alg_bytes_total / a.steps / a.gpus),
            "nnz_win_mean": round(pix_total / max(snip_total, 1), 1),
            "note": ("frac > 1 is possible by construction: 'achieved' charges every window its own algorithmic bytes "
                     "(SURVEY 8d), while the block-staged kernel serves all windows of a 16x16 block from one region "
                     "staged in LDS; 'traffic' is the real HBM byte count per launch (rocprofv3 PMC)"
                     if st.get("staged_regions", 0) > 0 else "achieved = algorithmic bytes per launch / kernel time"),
            "peak_measured_GBps": (None if peak_measured is None else
                                   {k: peak_measured[k] for k in ("read_GBps", "copy_GBps", "triad_GBps")}),
            "staged_regions_per_launch": int(st.get("staged_regions", 0)),
            "prepass_ms_per_launch": round(st.get("prepare_ms", 0.0) / launches, 4),
        }
        # ---- CPU baseline + same-run parity on a bounded sample (N=1 only) ------------------------------
        cpu = None
        if a.gpus == 1 and a.cpu_sample > 0:
            from oracle import pileup_oracle as po
            po.build()
            m = min(a.cpu_sample, n_set)
            idx = np.linspace(0, n_set - 1, m).astype(np.int64)
            sr0, sc0 = wl["r0"][idx], wl["c0"][idx]
            stile = (idx >= n_roi).astype(np.int32)
            # (1) one core: the scalar port as is, on a quarter of the sample
            m1 = max(1, m // 4)
            t = time.perf_counter()
            po.pileup_c(wl["bin1_offset"], wl["bin2_id"], wl["count"], wl["weight"], None, None,
                        sr0[:m1], sc0[:m1], None, stile[:m1], 2, a.pad, 2, 0)
            one_core = m1 / (time.perf_counter() - t)
            # (2) all host cores (capped): the same C function on C threads (ctypes releases the GIL), each with its
            #     own accumulators, summed afterwards — the reference's own grain is one region per process
            from concurrent.futures import ThreadPoolExecutor
            C_thr = max(1, min(a.cpu_threads if a.cpu_threads > 0 else (os.cpu_count() or 1), 64))
            cuts = np.linspace(0, m, C_thr + 1).astype(np.int64)
            arrs = [np.ascontiguousarray(x) for x in (wl["bin1_offset"], wl["bin2_id"], wl["count"], wl["weight"])]

            def work(k):
                lo_, hi_ = int(cuts[k]), int(cuts[k + 1])
                return po.pileup_c(arrs[0], arrs[1], arrs[2], arrs[3], None, None, sr0[lo_:hi_], sc0[lo_:hi_], None,
                                   stile[lo_:hi_], 2, a.pad, 2, 0)
            po._load()
            t = time.perf_counter()
            with ThreadPoolExecutor(C_thr) as ex:
                parts = list(ex.map(work, range(C_thr)))
            cpu_s = time.perf_counter() - t
            want = {k: sum(p[k] for p in parts) for k in ("sum", "num", "n")}
            sptr = np.array([0, int((stile == 0).sum()), m], np.int64)
            eng.reset(2, a.pad)
            eng.accumulate(sr0, sc0, sptr, ignore_diags=2, mode=0)
            got = eng.fetch()
            ok = (np.array_equal(got["n"], want["n"]) and np.array_equal(got["num"], want["num"])
                  and np.allclose(got["sum"], want["sum"], rtol=1e-6, atol=0, equal_nan=True))
            cpu = {"value": round(m / cpu_s, 1), "unit": "snippets/s", "cores": C_thr, "kind": "port",
                   "sample": f"{m} snippets strided over the {n_set} of this workload, C oracle (oracle/pileup_oracle.c) "
                             f"on {C_thr} threads, {cpu_s:.1f}s; one thread: {one_core:.0f} snippets/s",
                   "single_core_value": round(one_core, 1), "host_cpu_count": os.cpu_count(),
                   "gpu_matches_oracle_on_sample": bool(ok)}
            if not ok:
                print("[bench] PARITY FAILURE against the oracle on the sample", file=sys.stderr)
        line = {
            "metric": f"snippets/sec ({W}x{W} windows @10kb, ROI + control snippets accumulated)",
            "value": round(value, 1), "unit": "snippets/s", "n_gpus": a.gpus, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": a.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {
                "workload": "BASELINE configs[2]: synthetic hg38 10kb CSR + 1e6 random cis BEDPE pairs, pad=10, "
                            "nshifts=10, balanced, ignore_diags=2",
                "nnz": int(wl["bin2_id"].shape[0]), "nbins": int(wl["bin1_offset"].shape[0] - 1),
                "pairs": a.pairs, "nshifts": a.nshifts, "pad": a.pad, "snippets_per_step": n_all,
                "parallelism": (f"{a.gpus} rank(s), pixel table replicated; " +
                                ("each rank piles up its own 1e6-pair set" if a.scaling == "weak"
                                 else "one snippet set split evenly over the ranks") +
                                "; RCCL all-reduce of the packed tiles every step"),
            },
            "roofline": roofline, "cpu_baseline": cpu,
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
