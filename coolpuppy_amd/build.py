"""Build the gfx950 shared library in-tree (explicit hipcc; no JIT cache, no CUDA-compat layer).

The library is compiled as separate translation units — the engine, the host array passes, and eight units holding the
instantiations of the workgroup-staged kernel (csrc/pup_staged_tu.hip, one per group of window widths) — side by side
(one hipcc process per unit), then linked.  Objects live in csrc/_obj and are reused while their sources are older.
"""
import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor

_HERE = os.path.dirname(os.path.abspath(__file__))
_CSRC = os.path.join(_HERE, "csrc")
_OBJ = os.path.join(_CSRC, "_obj")
SRC = os.path.join(_CSRC, "pup_engine.hip")
SRC_HOST = os.path.join(_CSRC, "pup_host.cpp")          # pinned memory + host array passes (no kernels)
SRC_TU = os.path.join(_CSRC, "pup_staged_tu.hip")
SRC_WTU = os.path.join(_CSRC, "pup_wide_tu.hip")
WIDE_PARTS = range(0, 9)                                 # lane shapes of the wide-window staged kernel (csrc/pup_wide.hpp: wide_shape_ch / _nch)
N_STAGED_PARTS = 8                                       # = pup::kStagedParts (csrc/pup_staged_launch.hpp)
HEADER = os.path.join(os.path.dirname(_HERE), "include", "pup_hip.h")
# (every header under csrc/: a new one must make the library stale without anybody remembering to list it — tests/test_abi.py)
_KERNEL_HEADERS = sorted(os.path.join(_CSRC, h) for h in os.listdir(_CSRC) if h.endswith((".hpp", ".h")))
DEPS = sorted(os.path.join(_CSRC, f) for f in os.listdir(_CSRC) if f.endswith((".hip", ".cpp", ".hpp", ".h"))) + [HEADER]
OUT = os.path.join(_HERE, "libpup_hip.so")
_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-unused-result", "-pthread"] \
    + os.environ.get("COOLPUPPY_AMD_EXTRA_CXXFLAGS", "").split()       # (experiments: e.g. -DPUP_WIDE_NW=8)


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")


def _rocm_include():
    root = os.path.dirname(os.path.dirname(os.path.realpath(hipcc_path())))
    for cand in (os.path.join(root, "include"), "/opt/rocm/include"):
        if os.path.exists(os.path.join(cand, "hip", "hip_runtime.h")):
            return cand
    return "/opt/rocm/include"


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def _units():
    """(object file, compile arguments after the flags, sources it depends on)"""
    units = [(os.path.join(_OBJ, "pup_engine.o"), [SRC], [SRC, HEADER] + _KERNEL_HEADERS),
             (os.path.join(_OBJ, "pup_host.o"), ["-x", "c++", "-D__HIP_PLATFORM_AMD__", "-I" + _rocm_include(), SRC_HOST], [SRC_HOST, HEADER])]        # host only: plain C++ with the HIP runtime API
    for k in range(N_STAGED_PARTS):
        units.append((os.path.join(_OBJ, f"pup_staged_tu{k}.o"), [f"-DPUP_TU_PART={k}", SRC_TU], [SRC_TU, HEADER] + _KERNEL_HEADERS))
    for k in WIDE_PARTS:
        units.append((os.path.join(_OBJ, f"pup_wide_tu{k}.o"), [f"-DPUP_TU_PART={k}", SRC_WTU], [SRC_WTU, HEADER] + _KERNEL_HEADERS))
    return units


def build_hip(force=False, verbose=False, jobs=None):
    """Compile coolpuppy_amd/libpup_hip.so for gfx950. Returns the path."""
    if not force and not is_stale():
        return OUT
    os.makedirs(_OBJ, exist_ok=True)
    hipcc = hipcc_path()
    todo = []
    for obj, args, deps in _units():
        if force or not os.path.exists(obj) or any(os.path.getmtime(d) > os.path.getmtime(obj) for d in deps):
            todo.append([hipcc] + _FLAGS + ["-c"] + args + ["-o", obj])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        res = subprocess.run(cmd, capture_output=True, text=True)
        if res.returncode != 0:
            raise RuntimeError(f"hipcc failed ({res.returncode}): {' '.join(cmd)}\n{res.stdout}\n{res.stderr}")

    jobs = jobs or int(os.environ.get("COOLPUPPY_AMD_BUILD_JOBS", "0")) or max(1, min(len(todo), os.cpu_count() or 1))
    with ThreadPoolExecutor(max_workers=max(1, jobs)) as pool:
        list(pool.map(run, todo))
    run([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-pthread"] + [u[0] for u in _units()] + ["-o", OUT])
    return OUT


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
