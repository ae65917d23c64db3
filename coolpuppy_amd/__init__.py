"""coolpuppy_amd — MI355X-native pile-up engine behind coolpuppy's pileup()/PileUpper API.

Only the hot path of open2c/coolpuppy lives here: HIP kernels + C ABI in ``csrc/``,
the ctypes binding, and the host-side mirror of the reference's coordinate/orchestration layer.
"""
__version__ = "0.1.0"
