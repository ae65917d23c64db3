"""ctypes binding of ``libpup_hip.so`` (C ABI declared in ``include/pup_hip.h``).

This is the ONLY way the Python host layer reaches the HIP kernels.  There is no CPU fallback:
if the shared library is missing the import of :func:`lib` fails loudly, and if no GPU is visible
``pup_create`` fails with the HIP error text.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpup_hip.so")

# mode bits (mirror include/pup_hip.h)
MODE_OOE = 0x01
MODE_EXPECTED = 0x02
MODE_COV = 0x04
MODE_TRANSPOSE = 0x08
MODE_DEVPTR = 0x10
MODE_LOCAL = 0x20

PUP_OK = 0
ERROR_NAMES = {
    -1: "PUP_EINVAL", -2: "PUP_ENOMEM", -3: "PUP_EHIP", -4: "PUP_ESTATE", -5: "PUP_ERANGE", -6: "PUP_ENOTSUP",
}


class PupStats(C.Structure):
    _fields_ = [
        ("k1_ms", C.c_double),
        ("reduce_ms", C.c_double),
        ("k1_launches", C.c_int64),
        ("snippets", C.c_int64),
        ("pixels_in_windows", C.c_int64),
        ("probe_loads", C.c_int64),
        ("coverage_ms", C.c_double),
        ("staged_regions", C.c_int64),
        ("prepare_ms", C.c_double),
    ]


class PupError(RuntimeError):
    """A libpup_hip call returned a negative code."""

    def __init__(self, code, msg):
        self.code = code
        super().__init__(f"{ERROR_NAMES.get(code, code)}: {msg}")


# every symbol include/pup_hip.h declares, with its signature
_SIGNATURES = {
    "pup_create": (C.c_int, [C.c_int, C.POINTER(C.c_void_p)]),
    "pup_destroy": (None, [C.c_void_p]),
    "pup_last_error": (C.c_char_p, [C.c_void_p]),
    "pup_version": (C.c_int, []),
    "pup_device_count": (C.c_int, []),
    "pup_load_pixels": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int64, C.c_int64]),
    "pup_load_pixel_values": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pup_build_index": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int64]),
    "pup_coverage": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "pup_load_bins": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pup_load_pixels_stream": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int, C.c_int64, C.c_void_p, C.c_void_p,
                                       C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "pup_set_expected": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pup_set_expected_table": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                         C.c_void_p, C.c_int64, C.c_void_p]),
    "pup_reset": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "pup_accumulate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                 C.c_int32, C.c_uint32]),
    "pup_accumulate_rescaled": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                          C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32]),
    "pup_stripes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_uint32,
                              C.c_void_p, C.c_void_p]),
    "pup_extract": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32,
                              C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pup_sync": (C.c_int, [C.c_void_p]),
    "pup_fetch": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pup_packed_sizes": (C.c_int, [C.c_void_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "pup_export": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pup_import": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "pup_allreduce": (C.c_int, [C.c_void_p, C.c_void_p]),
    "pup_pack_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "pup_unpack_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32]),
    "pup_allgather_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32]),
    "pup_rccl_path": (C.c_int, [C.c_char_p, C.c_size_t]),
    "pup_set_profiling": (C.c_int, [C.c_void_p, C.c_int]),
    "pup_get_stats": (C.c_int, [C.c_void_p, C.POINTER(PupStats)]),
    "pup_clear_stats": (C.c_int, [C.c_void_p]),
    "pup_debug_timing": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64]),
    "pup_last_kernel": (C.c_char_p, [C.c_void_p]),
    "pup_last_prepass": (C.c_char_p, [C.c_void_p]),
    "pup_event_record": (C.c_int, [C.c_void_p, C.c_int]),
    "pup_event_elapsed_ms": (C.c_int, [C.c_void_p, C.c_int, C.c_int, C.POINTER(C.c_float)]),
    "pup_set_tuning": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32]),
    "pup_host_alloc": (C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    "pup_host_free": (C.c_int, [C.c_void_p]),
    "pup_host_windows": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                     C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                     C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int64)]),
    "pup_host_control_windows": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32,
                                             C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                             C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pup_host_mt_randint": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64,
                                      C.c_void_p, C.c_int32]),
    "pup_host_lut_i32": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "pup_host_count_le": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_void_p]),
    "pup_host_normalise_tiles": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64]),
    "pup_host_mt_randint_plan": (C.c_int, [C.c_void_p, C.POINTER(C.c_int32), C.c_int32] + [C.c_void_p] * 7),
    "pup_host_factorize_ptr": (C.c_int64, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int64]),
    "pup_host_argsort": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "pup_host_sort_pairs": (C.c_int64, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64,
                                        C.c_void_p, C.c_int32, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_int32)]),
    "pup_host_pair_region_counts": (C.c_int, [C.c_void_p] * 6 + [C.c_int64, C.c_double, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p,
                                                C.c_int32, C.c_void_p]),
    "pup_host_take_rows": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64]),
    "pup_host_group_tiles_runs": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]),
    "pup_host_group_tiles": (C.c_int, [C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p,
                                       C.c_void_p, C.c_void_p]),
}

FILL_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p)     # pup_fill_fn

_lib = None


def _prefer_torch_hip_runtime():
    """If PyTorch-ROCm is installed but not imported yet, load ITS bundled libamdhip64 first.

    libpup_hip.so needs libamdhip64.so.7; a process can hold only one copy of that soname.  When this library loads
    the system ROCm copy first and torch is imported afterwards, torch ends up on a runtime it was not built with
    and reports "No HIP GPUs are available".  Loading torch's copy first (what happens anyway when torch is
    imported first) keeps both on the runtime torch ships.  Without torch the system ROCm runtime is used."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("COOLPUPPY_AMD_SYSTEM_HIP", "") == "1":
        return
    try:
        spec = importlib.util.find_spec("torch")
    except Exception:
        spec = None
    if spec is None or not spec.origin:
        return
    cand = os.path.join(os.path.dirname(spec.origin), "lib", "libamdhip64.so")
    if os.path.exists(cand):
        try:
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
        except OSError:
            pass


def lib():
    """Load (once) and return the ctypes handle; raises if the HIP extension was not built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is not built. Run "
                "`python -c 'import __graft_entry__ as g; g.build()'` (or coolpuppy_amd.build.build_hip()). "
                "There is no CPU fallback."
            )
        _prefer_torch_hip_runtime()
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in _SIGNATURES.items():
            fn = getattr(handle, name)   # AttributeError if the .so lacks a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


_rccl = None


def _torch_before_rccl():
    """Load order matters on ROCm: a process that maps librccl (and with it librocm_smi64) FIRST and imports PyTorch-ROCm
    LATER dies at exit with a double free in a static destructor of librocm_smi64 — with either ROCm stack, measured on
    the GPU box (tools/exit_scenarios.py; this was GPUTEST_r02's rc 134: a test imported torch after the single-rank RCCL
    test).  torch first, RCCL second is clean.  In the library's own multi-GPU path torch.distributed is imported long
    before a communicator is made; for direct users of pup_allreduce the import is done here when torch is installed."""
    import importlib.util
    import sys
    if "torch" in sys.modules or os.environ.get("COOLPUPPY_AMD_NO_TORCH_PRELOAD", "") == "1":
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:       # noqa: BLE001 - a broken torch install must not keep RCCL from loading
        pass


def rccl():
    """ctypes handle of THE librccl of this process: the copy beside the HIP runtime the engine runs on (pup_rccl_path),
    so that communicators handed to pup_allreduce, the engine's own dlopen and torch (when imported) share one ROCm
    stack.  Loading the system librccl into a process that holds torch's libamdhip64 aborts at interpreter exit."""
    global _rccl
    if _rccl is None:
        _torch_before_rccl()
        buf = C.create_string_buffer(4096)
        n = lib().pup_rccl_path(buf, len(buf))
        if n < 0:
            raise PupError(n, "pup_rccl_path failed")
        _rccl = C.CDLL(buf.value.decode(), mode=C.RTLD_GLOBAL)
    return _rccl


def rccl_path():
    buf = C.create_string_buffer(4096)
    n = lib().pup_rccl_path(buf, len(buf))
    if n < 0:
        raise PupError(n, "pup_rccl_path failed")
    return buf.value.decode()


def declared_symbols():
    return sorted(_SIGNATURES)
