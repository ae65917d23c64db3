"""Per-snippet Python callbacks (postprocess_func / extra_sum_funcs) against goldens produced by the reference's own
PileUpper.pileupsWithControl (oracle/make_golden.py::callback_goldens, scenarios in oracle/callbacks.py).

CPU: the host orchestration with the numpy/scipy oracle standing in for pup_extract.  GPU: the product path —
windows from the HIP extraction kernels through the C ABI."""
import os

import numpy as np
import pytest

from coolpuppy_amd import coolpup
from coolpuppy_amd.lib import puputils
from oracle import callbacks as cbs
from oracle import make_golden as mg
from oracle import pileup_oracle as po

import golden_util as gu

_SC = None


def _scenarios():
    global _SC
    if _SC is None:
        import synth
        small = gu.cooler("small")
        _SC = {s["name"]: s for s in cbs.scenarios(mg.bedpe_features(small), mg.bed_features(small), mg.tad_features(),
                                                   synth.cis_expected(small),
                                                   inf_patch=mg.inf_weight_patch(int(small.nbins)))}
    return _SC


def _cooler(sc):
    import synth
    small = gu.cooler("small")
    return synth.patched_cooler(small, sc["patch"]) if sc.get("patch") else small


NAMES = ["G13a_bedpe_controls_collect_centre", "G13b_rescale_local_expected_domain_score", "G13c_bedpe_double_data",
         "G13d_bed_strand_flip_double", "G13e_bedpe_group_by_region", "G13f_expected_not_ooe_stripes_centre",
         "G13g_band_group_postprocess", "G13h_ignore_group_order_strands", "G13i_raw_covnorm_double",
         "G14e_inf_weights_callback_centre"]


def _check(name, df, rtol):
    z = np.load(os.path.join(gu.GOLD, name + ".npz"))
    gu.compare(z, df, rtol)
    for key in ("centre", "control_centre"):
        if f"extra__{key}__ptr" not in z.files:
            assert key not in df.columns
            continue
        ptr, vals, is_list = z[f"extra__{key}__ptr"], z[f"extra__{key}__vals"], z[f"extra__{key}__is_list"]
        for i, cell in enumerate(df[key]):
            if is_list[i]:
                np.testing.assert_allclose(np.asarray(cell, float), vals[ptr[i]:ptr[i + 1]], rtol=rtol, atol=0,
                                           equal_nan=True)
            else:
                assert not isinstance(cell, (list, tuple, np.ndarray))


class _OraclePileUpper(coolpup.PileUpper):
    _window_source = staticmethod(gu.oracle_windows)


class _HostMod:
    """coolpup with the oracle window source plugged in (CPU tests only)."""
    CoordCreator = coolpup.CoordCreator
    PileUpper = _OraclePileUpper


@pytest.mark.parametrize("name", NAMES)
def test_callbacks_host_logic_vs_reference(name):
    df = cbs.run(_HostMod, puputils, _cooler(_scenarios()[name]), _scenarios()[name])
    _check(name, df, rtol=1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_callbacks_gpu_vs_reference(name, hip_lib):
    df = cbs.run(coolpup, puputils, _cooler(_scenarios()[name]), _scenarios()[name])
    _check(name, df, rtol=1e-9)


@pytest.mark.gpu
def test_extract_matches_oracle_windows(hip_lib):
    """pup_extract alone: plain, OOE, EXPECTED, TRANSPOSE and coverage windows, bit-for-bit against the oracle."""
    from coolpuppy_amd.engine import MODE_COV, MODE_EXPECTED, MODE_OOE, MODE_TRANSPOSE, PileupEngine
    import synth
    clr = gu.cooler("small")
    indptr, col, cnt = clr.pixel_table()
    weight = clr.bins()["weight"][:].values
    cov = clr.bins()["cov_tot_raw"][:].values
    nb = indptr.shape[0] - 1
    big = po.symmetric_csr(indptr, col, cnt, weight, 0, nb, 0, nb)
    raw = po.symmetric_csr(indptr, col, cnt, None, 0, nb, 0, nb)
    exp = synth.cis_expected(clr)
    ev = exp[exp.region1 == "chrA"]["balanced.avg"].values.astype(float)
    rng = np.random.default_rng(7)
    hiA = int(clr.chrom_offset[1])
    for pad in (3, 10, 17):
        W = 2 * pad + 1
        r0 = rng.integers(0, hiA - 3 * W, 300).astype(np.int32)
        c0 = (r0 + rng.integers(0, 2 * W, 300)).astype(np.int32)
        for use_idx in (False, True):
            eng = PileupEngine(0)
            eng.load_pixels(indptr, col, cnt)
            if use_idx:
                eng.build_index(clr.chrom_offset)
            eng.load_bins(weight, cov)
            eng.set_expected(ev)
            for mode in (0, MODE_OOE, MODE_EXPECTED, MODE_COV):
                got = eng.extract(r0, c0, pad, ignore_diags=2, mode=mode, coverage=bool(mode & MODE_COV))
                want = po.windows_scipy(big, 0, 0, weight, cov, ev, r0, c0, pad, 2, mode)
                if mode & MODE_COV:
                    np.testing.assert_array_equal(got[1], want[1]); np.testing.assert_array_equal(got[2], want[2])
                    got = got[0]
                np.testing.assert_array_equal(got, want[0])
            # trans-style call: no diagonal mask, rows/cols handed over transposed
            rt = rng.integers(0, hiA - W, 100).astype(np.int32)
            ct = rng.integers(hiA, nb - W, 100).astype(np.int32)
            got = eng.extract(rt, ct, pad, ignore_diags=-1, mode=MODE_TRANSPOSE)
            want = po.windows_scipy(big, 0, 0, weight, None, None, rt, ct, pad, -1, MODE_TRANSPOSE)[0]
            np.testing.assert_array_equal(got, want)
            eng.load_bins(None, None)
            got = eng.extract(r0, c0, pad, ignore_diags=0, mode=0)
            np.testing.assert_array_equal(got, po.windows_scipy(raw, 0, 0, None, None, None, r0, c0, pad, 0, 0)[0])
            eng.close()
