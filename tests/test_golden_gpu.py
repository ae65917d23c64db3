"""GPU tests: pileup() through the real HIP engine reproduces the reference's outputs (tests/golden)."""
import pytest

import golden_util as gu
from coolpuppy_amd import coolpup

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", gu.SCENARIOS)
def test_reference_golden_on_gpu(name, hip_lib):
    z, df = gu.run(name, coolpup.pileup)
    gu.compare(z, df, rtol=1e-6)   # north-star tolerance; integers are compared exactly


# every scenario again with the workgroup-staged kernel (K1q) forced for each eligible engine call (cis, pad <= 15): covers its
# expected-table path, sub-chromosomal views, flips and grouped tiles on the reference's own outputs
@pytest.mark.parametrize("name", gu.SCENARIOS)
def test_reference_golden_on_gpu_staged_kernel(name, hip_lib, monkeypatch):
    monkeypatch.setenv("COOLPUPPY_AMD_VARIANT", "8")
    z, df = gu.run(name, coolpup.pileup)
    gu.compare(z, df, rtol=1e-6)


def test_native_library_is_what_ran(hip_lib):
    """The HIP shared object is loaded in this process (no silent fallback exists)."""
    loaded = open("/proc/self/maps").read()
    assert "libpup_hip.so" in loaded
