"""Multi-GPU plumbing: one process per GPU, regions sharded, one RCCL all-reduce of the packed tiles.

The reference parallelises with ``multiprocessing.Pool`` over regions and merges per-region dicts with
``reduce(sum_pups)`` on the host (coolpuppy/coolpup.py:1495-1531).  Here each rank piles up its share of
the regions on its own GPU and the (kind, group) accumulators — additive by construction — are summed
across ranks with ``torch.distributed.all_reduce`` (backend "nccl" = RCCL over xGMI; "gloo" on CPU for
tests).  Message: n_tiles*(W*W+2W) float64 + n_tiles*(W*W+1) int64, ~1 MB at most outside by-window mode,
so the collective is latency-bound and a single flat all-reduce per dtype is the right shape.

torch is imported lazily and only when a process group exists: single-GPU use has no torch dependency.
"""
import os

import numpy as np


def _dist():
    try:
        import torch.distributed as dist
    except Exception:
        return None
    if dist.is_available() and dist.is_initialized():
        return dist
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


def slice_call(call, rank, world_size):
    """This rank's share of one engine call: an even, contiguous slice of every (tile, flip) segment of the
    call's tile-grouped snippets (deterministic, no communication).  Returns a call dict of the same shape."""
    if world_size == 1:
        return call
    tp = np.asarray(call["tile_ptr"], np.int64)
    T = len(tp) - 1
    ff = tp[1:] if call.get("flip_from") is None else np.asarray(call["flip_from"], np.int64)
    # segment boundaries: [tp[t], ff[t]) as is, [ff[t], tp[t+1]) flipped
    a = np.stack([tp[:-1], ff], axis=1).ravel()
    b = np.stack([ff, tp[1:]], axis=1).ravel()
    n = b - a
    lo = a + (n * rank) // world_size
    hi = a + (n * (rank + 1)) // world_size
    cnt = hi - lo
    total = int(cnt.sum())
    if total:
        starts = np.repeat(lo - np.concatenate([[0], np.cumsum(cnt)[:-1]]), cnt)
        idx = starts + np.arange(total)
    else:
        idx = np.zeros(0, np.int64)
    per_tile = cnt.reshape(T, 2)
    new_tp = np.concatenate([[0], np.cumsum(per_tile.sum(axis=1))]).astype(np.int64)
    out = dict(call)
    for k in ("r0", "c0", "tile", "flip", "h", "w"):
        if call.get(k) is not None:
            out[k] = np.ascontiguousarray(call[k][idx])
    out["tile_ptr"] = new_tp
    out["flip_from"] = None if call.get("flip_from") is None else (new_tp[:-1] + per_tile[:, 0]).astype(np.int64)
    return out


def merge_dicts(mine):
    """Union over all ranks of per-rank dicts with disjoint keys (small control-plane objects: group keys of the
    regions a rank owns, per-region tiles of the rare inf merge) — identical on every rank afterwards."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return mine
    parts = [None] * d.get_world_size()
    d.all_gather_object(parts, mine)
    out = {}
    for p in parts:
        out.update(p)
    return out


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


_NATIVE_COMMS = {}


def native_comm(eng):
    """An RCCL communicator (ncclComm_t address) for this engine's device over all ranks, created once: rank 0 draws
    the ncclUniqueId, torch.distributed (any backend) only carries its 128 bytes to the other ranks.  Opt-in path
    (COOLPUPPY_AMD_NATIVE_RCCL=1): exercised on hardware with a one-rank communicator only in round 1."""
    import ctypes as C
    d = _dist()
    rank, world = d.get_rank(), d.get_world_size()
    key = (eng.device_id, world)
    if key in _NATIVE_COMMS:
        return _NATIVE_COMMS[key][0]
    try:
        rccl = C.CDLL("librccl.so.1")
    except OSError:
        rccl = C.CDLL("librccl.so")

    class UniqueId(C.Structure):
        _fields_ = [("internal", C.c_char * 128)]
    uid = UniqueId()
    if rank == 0 and rccl.ncclGetUniqueId(C.byref(uid)) != 0:
        raise RuntimeError("ncclGetUniqueId failed")
    import torch
    t = torch.frombuffer(bytearray(bytes(uid)), dtype=torch.uint8).clone()
    if d.get_backend() == "nccl":
        t = t.cuda(eng.device_id)
    d.broadcast(t, src=0)
    C.memmove(C.byref(uid), bytes(t.cpu().numpy().tobytes()), 128)
    comm = C.c_void_p()
    rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
    torch.cuda.set_device(eng.device_id)
    if rccl.ncclCommInitRank(C.byref(comm), world, uid, rank) != 0:
        raise RuntimeError("ncclCommInitRank failed")
    _NATIVE_COMMS[key] = (comm.value, rccl)
    return comm.value


def allreduce_engine(eng):
    """All-reduce the engine's packed accumulators across ranks: device to device with the nccl (= RCCL) backend,
    through host memory with any other backend."""
    d = _dist()
    if d is None or d.get_world_size() == 1:
        return
    import torch
    if os.environ.get("COOLPUPPY_AMD_NATIVE_RCCL", "") == "1":
        eng.allreduce(native_comm(eng))       # in place on the engine's stream: no staging copies, no host sync
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
