"""Multi-GPU plumbing: one process per GPU, regions sharded, one RCCL all-reduce of the packed tiles.

The reference parallelises with ``multiprocessing.Pool`` over regions and merges per-region dicts with
``reduce(sum_pups)`` on the host (coolpuppy/coolpup.py:1495-1531).  Here the regions (chromosomes, trans pairs) are
dealt to the ranks (`shard`), each rank piles up its own on its GPU from the rows of the pixel table it needs, and the
(kind, group) accumulators — additive by construction — are summed across ranks by one all-reduce (`allreduce_engine`:
RCCL over xGMI on the engine's stream; "gloo" on CPU for tests).  Control-plane facts (group keys of the regions, the
seed, stripes) travel as small pickled objects (`merge_dicts`, `shared_seed`).  Message: n_tiles*(W*W+2W) float64 + n_tiles*(W*W+1) int64, ~1 MB at most outside by-window mode,
so the collective is latency-bound and a single flat all-reduce per dtype is the right shape.

torch is imported lazily and only when a process group exists: single-GPU use has no torch dependency.
"""
import os

import numpy as np


def _dist():
    """torch.distributed when THIS process has initialised a process group, else None — without importing torch: a group
    can only exist if the caller imported torch.distributed already (importing it here cost every single-GPU pileup()
    its first 3 s)."""
    import sys
    dist = sys.modules.get("torch.distributed")
    if dist is None:
        return None
    try:
        if dist.is_available() and dist.is_initialized():
            return dist
    except Exception:
        return None
    return None


def world():
    d = _dist()
    return (d.get_rank(), d.get_world_size()) if d is not None else (0, 1)


def local_device():
    """GPU index of this process: COOLPUPPY_AMD_DEVICE, else LOCAL_RANK, else 0."""
    for var in ("COOLPUPPY_AMD_DEVICE", "LOCAL_RANK"):
        if os.environ.get(var, "") != "":
            return int(os.environ[var])
    return 0


def _bind_device(d):
    """Object collectives of the nccl backend stage through the CURRENT CUDA device: make that this rank's GPU."""
    if d.get_backend() == "nccl":
        import torch
        torch.cuda.set_device(local_device())


def shard(n_units, weights=None, rank=None, world_size=None):
    """Indices of the units (regions / region pairs) this rank piles up.

    Longest-processing-time assignment on ``weights`` (snippets per unit): deterministic, identical on
    every rank, no communication.  Returns a set.
    """
    if rank is None or world_size is None:
        rank, world_size = world()
    if world_size == 1:
        return set(range(n_units))
    w = np.ones(n_units) if weights is None else np.asarray(weights, dtype=np.float64)
    load = np.zeros(world_size)
    mine = set()
    for i in np.argsort(-w, kind="stable"):
        r = int(np.argmin(load))
        load[r] += w[i]
        if r == rank:
            mine.add(int(i))
    return mine


def merge_dicts(mine):
    """Union over all ranks of per-rank dicts with disjoint keys (small control-plane objects: group keys of the
    regions a rank owns, per-region tiles of the rare inf merge) — identical on every rank afterwards."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return mine
    _bind_device(d)
    parts = [None] * d.get_world_size()
    d.all_gather_object(parts, mine)
    out = {}
    for p in parts:
        out.update(p)
    return out


def check_same_plan(plan):
    """Before tiles are summed across ranks: every rank must hold the same tile layout (number of tiles, window size,
    group keys in the same order) — it does by construction (the group table is built from the swapped region keys,
    the control RNG from a broadcast seed); a mismatch would add unrelated tiles silently, so it is an error here."""
    import hashlib
    text = repr((plan["T"], plan["pad"], plan["n_regions"], sorted((repr(k), v) for k, v in plan["gid"].items())))
    digest = hashlib.sha1(text.encode()).hexdigest()
    rank, _ = world()
    seen = merge_dicts({rank: digest})
    if len(set(seen.values())) != 1:
        raise RuntimeError(f"multi-GPU pile-up: the ranks built different plans ({seen}); same inputs and seed on every rank?")


def shared_seed(seed):
    """The seed every rank uses: the caller's, or — when it is None and there are several ranks — one drawn by rank 0
    and broadcast, so that the random control shifts are the same sequence everywhere."""
    d = _dist()
    if seed is not None or d is None or d.get_world_size() == 1:
        return seed
    _bind_device(d)
    box = [int(np.random.randint(0, 2**31 - 1)) if d.get_rank() == 0 else None]
    d.broadcast_object_list(box, src=0)
    return box[0]


def allreduce_arrays(f64, i64):
    """Sum a float64 and an int64 numpy array over all ranks (in place where possible); returns both."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return f64, i64
    import torch
    backend = d.get_backend()
    dev = torch.device("cuda", local_device()) if backend == "nccl" else torch.device("cpu")
    tf = torch.from_numpy(np.ascontiguousarray(f64)).to(dev)
    ti = torch.from_numpy(np.ascontiguousarray(i64)).to(dev)
    d.all_reduce(tf)
    d.all_reduce(ti)
    return tf.cpu().numpy(), ti.cpu().numpy()


_NATIVE_COMMS = {}     # (device, world) -> (ncclComm_t address | None, librccl handle | None); None: the ranks agreed to do without


def _all_ok(d, ok, device_id):
    """True when `ok` holds on EVERY rank (one small all-reduce on the process group): the ranks must take the same
    exchange path — a rank that fell back on its own would sit in torch.distributed.all_reduce while the others wait in
    ncclAllReduce on the private communicator."""
    import torch
    t = torch.tensor([1 if ok else 0], dtype=torch.int32)
    if d.get_backend() == "nccl":
        t = t.cuda(device_id)
    d.all_reduce(t, op=d.ReduceOp.MIN)
    return bool(int(t.cpu()[0]))


def native_comm(eng):
    """An RCCL communicator (ncclComm_t address) for this engine's device over all ranks, created once, or None when the
    ranks agree that it cannot be had.  The librccl is the one beside the HIP runtime this process runs on (`_ffi.rccl`,
    pup_rccl_path): one ROCm stack per process.  Rank 0 draws the ncclUniqueId; torch.distributed (any backend) only
    carries its 128 bytes.  Every step that can fail on one rank alone (dlopen, ncclGetUniqueId, ncclCommInitRank) is
    followed by an agreement over the process group, so either every rank returns a communicator or every rank returns
    None — never a mixture (which would dead-lock the exchange)."""
    import ctypes as C
    import torch
    from . import _ffi
    d = _dist()
    rank, world = d.get_rank(), d.get_world_size()
    key = (eng.device_id, world)
    if key in _NATIVE_COMMS:
        return _NATIVE_COMMS[key][0]

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    rccl, why = None, ""
    # every rank-local step that can fail is caught HERE and turned into ok = False: a rank that raised before joining the
    # agreement below would leave its peers waiting in it (PupError / AttributeError escaped past `except OSError`)
    try:
        rccl = _ffi.rccl()
        if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
            rccl, why = None, "ncclGetUniqueId failed"
    except Exception as e:       # noqa: BLE001 - any failure must reach the agreement
        rccl, why = None, f"{type(e).__name__}: {e}"
    if not _all_ok(d, rccl is not None, eng.device_id):      # rank 0's failure reaches everybody BEFORE the broadcast
        _NATIVE_COMMS[key] = (None, None)
        raise RuntimeError(why or "librccl unavailable on another rank")
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
    if d.get_backend() == "nccl":
        t = t.cuda(eng.device_id)
    d.broadcast(t, src=0)
    C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
    comm = C.c_void_p()
    rc, why = -1, ""
    try:
        rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
        if torch.cuda.is_available():
            torch.cuda.set_device(eng.device_id)
        eng.sync()                                           # binds the engine's device in the HIP runtime
        rc = rccl.ncclCommInitRank(C.byref(comm), world, uid, rank)
    except Exception as e:       # noqa: BLE001
        rc, why = -1, f"{type(e).__name__}: {e}"
    if not _all_ok(d, rc == 0 and bool(comm.value), eng.device_id):
        if rc == 0 and comm.value:
            rccl.ncclCommDestroy.argtypes = [C.c_void_p]
            rccl.ncclCommDestroy(comm)
        _NATIVE_COMMS[key] = (None, None)
        raise RuntimeError(why or ("ncclCommInitRank failed" + ("" if rc else " on another rank")))
    _NATIVE_COMMS[key] = (comm.value, rccl)
    return comm.value


def comm_ranks(comm):
    """ncclCommCount of a communicator made by `native_comm` (diagnostics: bench.py prints it)."""
    import ctypes as C
    from . import _ffi
    n = C.c_int(0)
    rccl = _ffi.rccl()
    rccl.ncclCommCount.argtypes = [C.c_void_p, C.POINTER(C.c_int)]
    if rccl.ncclCommCount(C.c_void_p(comm), C.byref(n)) != 0:
        return -1
    return n.value


def destroy_native_comms():
    """ncclCommDestroy every communicator `native_comm` made (interpreter exit; tests)."""
    import ctypes as C
    for key in list(_NATIVE_COMMS):
        comm, rccl = _NATIVE_COMMS.pop(key)
        if comm and rccl is not None:
            try:
                rccl.ncclCommDestroy.argtypes = [C.c_void_p]
                rccl.ncclCommDestroy(C.c_void_p(comm))
            except Exception:       # noqa: BLE001 - shutting down
                pass


def allreduce_engine(eng):
    """All-reduce the engine's packed accumulators across ranks.  With the nccl (= RCCL) backend the engine does it
    itself: pup_allreduce, in place on its own stream over a communicator of its own — no staging buffers, no host
    synchronisation (COOLPUPPY_AMD_NATIVE_RCCL=0 selects torch.distributed's all_reduce on exported buffers instead, which
    is also what is used if the engine's communicator cannot be set up).  Any other backend (gloo: CPU tests, two ranks
    sharing one GPU) goes through host memory."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return
    import torch
    if d.get_backend() == "nccl" and os.environ.get("COOLPUPPY_AMD_NATIVE_RCCL", "1") != "0":
        try:
            comm = native_comm(eng)         # all ranks get one, or all ranks get None / the exception (agreed inside)
        except (RuntimeError, OSError) as e:
            import warnings
            warnings.warn(f"engine-side RCCL communicator unavailable ({e}); using torch.distributed.all_reduce")
            comm = None
        if comm is not None:
            eng.allreduce(comm)
            return
    nf, ni = eng.packed_sizes()
    dev = torch.device("cuda", eng.device_id)
    bf = torch.empty(nf, dtype=torch.float64, device=dev)
    bi = torch.empty(ni, dtype=torch.int64, device=dev)
    torch.cuda.synchronize(dev)
    eng.export_to(bf.data_ptr(), bi.data_ptr())
    if d.get_backend() == "nccl":          # RCCL over xGMI, device to device
        d.all_reduce(bf)
        d.all_reduce(bi)
    else:                                  # e.g. gloo (tests: two ranks sharing one GPU): through host memory
        hf, hi = bf.cpu(), bi.cpu()
        d.all_reduce(hf)
        d.all_reduce(hi)
        bf.copy_(hf)
        bi.copy_(hi)
    torch.cuda.synchronize(dev)
    eng.import_from(bf.data_ptr(), bi.data_ptr())


def sparse_exchange_min_tiles():
    """Plans of at least this many tiles swap the tiles each rank piled windows into (`exchange_tiles`) instead of all-reducing
    every accumulator: below it the flat all-reduce is one latency-bound call and cannot be beaten."""
    return int(os.environ.get("COOLPUPPY_AMD_SPARSE_EXCHANGE_MIN_TILES", "1024"))


def gather_tile_lists(mine):
    """(tile numbers of all ranks concatenated in rank order, rank_ptr) from this rank's sorted list — control plane, a few
    thousand integers per rank."""
    d = _dist()
    _bind_device(d)
    parts = [None] * d.get_world_size()
    d.all_gather_object(parts, np.asarray(mine, np.int32))
    ptr = np.concatenate([[0], np.cumsum([len(p) for p in parts])]).astype(np.int64)
    ids = np.concatenate(parts).astype(np.int32) if ptr[-1] else np.zeros(0, np.int32)
    return ids, ptr


def exchange_tiles(eng, mine):
    """By-window exchange (reference: the per-feature pile-ups of the worker processes, coolpup.py:1696-1755, merged key by key):
    every rank contributes the tiles it piled windows into (`mine`, sorted tile numbers); afterwards every such tile holds, on
    every rank, the sum over the ranks that list it, added in rank order.  With the nccl backend the engine does it itself
    (pup_allgather_tiles: a broadcast of each rank's block on its stream); otherwise the blocks travel through
    torch.distributed (gloo: through host memory) and pup_unpack_tiles adds them.  Returns the bytes this rank sent."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return 0
    import torch
    rank, world_size = d.get_rank(), d.get_world_size()
    ids, ptr = gather_tile_lists(mine)
    nf_mine, ni_mine = eng.tile_block_sizes(int(ptr[rank + 1] - ptr[rank]))
    sent = 8 * (nf_mine + ni_mine)
    if ptr[-1] == 0:
        return 0
    if d.get_backend() == "nccl" and os.environ.get("COOLPUPPY_AMD_NATIVE_RCCL", "1") != "0":
        try:
            comm = native_comm(eng)
        except (RuntimeError, OSError) as e:
            import warnings
            warnings.warn(f"engine-side RCCL communicator unavailable ({e}); using torch.distributed broadcasts")
            comm = None
        if comm is not None:
            eng.allgather_tiles(comm, ids, ptr, rank)
            return sent
    dev = torch.device("cuda", eng.device_id)
    on_host = d.get_backend() != "nccl"
    blocks = []
    for r in range(world_size):
        nf, ni = eng.tile_block_sizes(int(ptr[r + 1] - ptr[r]))
        bf = torch.empty(max(nf, 1), dtype=torch.float64, device=dev)
        bi = torch.empty(max(ni, 1), dtype=torch.int64, device=dev)
        if r == rank and nf:
            eng.pack_tiles(ids[ptr[r]:ptr[r + 1]], bf.data_ptr(), bi.data_ptr())
        blocks.append((bf, bi, nf))
    eng.unpack_tiles(ids[ptr[rank]:ptr[rank + 1]], mode=2)          # own tiles cleared: every block is ADDED below, in rank order
    for r, (bf, bi, nf) in enumerate(blocks):
        if not nf:
            continue
        if on_host:
            hf, hi = bf.cpu(), bi.cpu()
            d.broadcast(hf, src=r)
            d.broadcast(hi, src=r)
            bf.copy_(hf)
            bi.copy_(hi)
        else:
            d.broadcast(bf, src=r)
            d.broadcast(bi, src=r)
        torch.cuda.synchronize(dev)
        eng.unpack_tiles(ids[ptr[r]:ptr[r + 1]], bf.data_ptr(), bi.data_ptr(), mode=1)
    return sent


def exchange_tile_arrays(acc, mine):
    """exchange_tiles on host arrays (the CPU stand-in of the tests): acc = dict of [T, ...] arrays (sum, num, n, cov_start,
    cov_end); same rule — a listed tile becomes the sum of the listing ranks' tiles, added in rank order."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return
    mine = np.asarray(mine, np.int64)
    parts = [None] * d.get_world_size()
    d.all_gather_object(parts, (mine, {k: np.ascontiguousarray(v[mine]) for k, v in acc.items()}))
    for v in acc.values():
        v[mine] = 0
    for ids, block in parts:
        for k, v in acc.items():
            np.add.at(v, ids, block[k])
