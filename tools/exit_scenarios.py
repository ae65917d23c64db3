"""Does the interpreter exit cleanly?  Fresh interpreters, one per scenario of library load order (GPU box).
GPUTEST_r02's rc 134 was a double free in a static destructor of librocm_smi64 at exit(): which orders trigger it?"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENGINE = r"""
import sys; sys.path.insert(0, %r)
import numpy as np
from coolpuppy_amd.engine import PileupEngine
import synth
clr = synth.make_cooler({"chrA": 8_000_000}, lam=60, seed=3)
eng = PileupEngine(0); eng.load_pixels(*clr.pixel_table()); eng.load_bins(clr.bins()["weight"][:].values, None)
eng.reset(1, 10); eng.accumulate(np.arange(100, dtype=np.int32), np.arange(100, dtype=np.int32) + 30, np.array([0, 100])); eng.fetch()
""" % ROOT
RCCL = r"""
import ctypes as C
from coolpuppy_amd import _ffi
rccl = _ffi.rccl()
class UniqueId(C.Structure):
    _fields_ = [("internal", C.c_char * 128)]
uid = UniqueId(); assert rccl.ncclGetUniqueId(C.byref(uid)) == 0
comm = C.c_void_p(); rccl.ncclCommInitRank.argtypes = [C.POINTER(C.c_void_p), C.c_int, UniqueId, C.c_int]
assert rccl.ncclCommInitRank(C.byref(comm), 1, uid, 0) == 0
eng.allreduce(comm); eng.sync()
rccl.ncclCommDestroy.argtypes = [C.c_void_p]; rccl.ncclCommDestroy(comm)
"""
TORCH = "import torch; print('torch sees', torch.cuda.device_count(), 'GPU(s)')\n"
TORCH_NOCOUNT = "import torch\n"
SCEN = {
    "engine": ENGINE,
    "engine+rccl": ENGINE + RCCL,
    "engine, then torch": ENGINE + TORCH,
    "engine, then torch (no device_count)": ENGINE + TORCH_NOCOUNT,
    "engine+rccl, then torch": ENGINE + RCCL + TORCH,
    "engine+rccl, then torch (no device_count)": ENGINE + RCCL + TORCH_NOCOUNT,
    "torch, then engine+rccl": TORCH + ENGINE + RCCL,
    "torch, then engine": TORCH + ENGINE,
    "engine, then torch, then rccl": ENGINE + TORCH + RCCL,
}
for env_name, env in (("default", {}), ("SYSTEM_HIP", {"COOLPUPPY_AMD_SYSTEM_HIP": "1"})):
    for name, code in SCEN.items():
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=dict(os.environ, **env), timeout=600)
        tail = (r.stdout + r.stderr).strip().splitlines()[-2:]
        print(f"[{env_name}] {name}: rc={r.returncode}  {' | '.join(tail)[-200:]}", flush=True)
