"""Build the gfx950 shared library in-tree (explicit hipcc; no JIT cache, no CUDA-compat layer)."""
import os
import shutil
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_HERE, "csrc", "pup_engine.hip")
SRC_HOST = os.path.join(_HERE, "csrc", "pup_host.cpp")          # pinned memory + host array passes (no kernels)
DEPS = [SRC, SRC_HOST, os.path.join(_HERE, "csrc", "pup_kernels.hpp"),
        os.path.join(os.path.dirname(_HERE), "include", "pup_hip.h")]
OUT = os.path.join(_HERE, "libpup_hip.so")


def hipcc_path():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found (looked on PATH and in /opt/rocm/bin)")


def is_stale():
    if not os.path.exists(OUT):
        return True
    t = os.path.getmtime(OUT)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build_hip(force=False, verbose=False):
    """Compile coolpuppy_amd/libpup_hip.so for gfx950. Returns the path."""
    if not force and not is_stale():
        return OUT
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden",
           "-Wall", "-Wno-unused-result", "-pthread", SRC, "-x", "hip", SRC_HOST, "-o", OUT]
    if os.environ.get("COOLPUPPY_AMD_DEV_W21", "") == "1":      # development only: see launch_staged in pup_engine.hip
        cmd.insert(1, "-DPUP_DEV_W21")
    if verbose:
        print(" ".join(cmd))
    res = subprocess.run(cmd, capture_output=True, text=True)
    if res.returncode != 0:
        raise RuntimeError(f"hipcc failed ({res.returncode}):\n{res.stdout}\n{res.stderr}")
    return OUT


if __name__ == "__main__":
    print(build_hip(force=True, verbose=True))
