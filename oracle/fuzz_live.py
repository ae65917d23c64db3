"""TEST INFRASTRUCTURE — live differential fuzzing against the REFERENCE (build container only: needs /root/reference).

Random combinations of pileup() options (bedpe / bed / local / trans / rescaled x controls / expected / coverage x
by-strand / by-distance / by-window / flips / stripes) are run through the unchanged reference (oracle/refshim.py
stand-ins) and through coolpuppy_amd's host layer with the CPU oracle as back-end, and compared like the goldens
(rows, columns, scalars, integers exact, floats rtol 1e-11).  Nothing is written:

    python -m oracle.fuzz_live <seed> <n_scenarios>

Round 1: seeds 101 / 202 / 303, 320 scenarios, 0 mismatches.
"""
import io
import json
import os
import sys
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, pandas as pd
from oracle import refshim, make_golden as mg
ref = refshim.import_reference()
import golden_util as gu
from coolpuppy_amd import coolpup
import synth
coolpup.PileUpper.run_plan = gu.oracle_run_plan
coolpup.PileUpper._window_source = staticmethod(gu.oracle_windows)
clr = mg.small_cooler()
bedpe = mg.bedpe_features(clr); bed = mg.bed_features(clr); tads = mg.tad_features()
exp_chrom = synth.cis_expected(clr)
view_chrom = pd.DataFrame({"chrom": clr.chromnames, "start": 0, "end": [int(clr.chromsizes[c]) for c in clr.chromnames], "name": clr.chromnames})
tr_exp = mg.trans_expected(clr, view_chrom)
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 1
N = int(sys.argv[2]) if len(sys.argv) > 2 else 60
rng = np.random.default_rng(seed0)
bad = 0
for k in range(N):
    kind = rng.choice(["bedpe", "bed", "bed_local", "trans", "rescale"])
    kw = dict(flank=int(rng.choice([50_000, 100_000, 150_000])))
    expected = None; view = None
    if kind == "bedpe":
        feats = bedpe.iloc[rng.choice(len(bedpe), int(rng.integers(40, 200)), replace=False)].sort_index(); kw["features_format"] = "bedpe"
        if rng.random() < 0.5: kw["mindist"] = int(rng.choice([0, 300_000]))
    elif kind in ("bed", "bed_local"):
        feats = bed.iloc[rng.choice(len(bed), int(rng.integers(20, 60)), replace=False)].sort_index(); kw["features_format"] = "bed"
        if kind == "bed_local": kw["local"] = True
        else: kw["mindist"] = int(rng.choice([200_000, 400_000])); kw["maxdist"] = int(rng.choice([2_000_000, 4_000_000]))
    elif kind == "trans":
        feats = mg.trans_bedpe(clr, int(rng.integers(60, 160)), int(rng.integers(0, 1000))); kw.update(features_format="bedpe", trans=True)
    else:
        feats = tads.iloc[rng.choice(len(tads), 9, replace=False)].sort_index()
        kw = dict(features_format="bed", local=True, rescale=True, rescale_flank=float(rng.choice([0.5, 1, 2])), rescale_size=int(rng.choice([15, 21, 33])))
    mode = rng.choice(["plain", "controls", "expected_ooe", "expected_not_ooe", "raw_cov"])
    if mode == "controls" and kind != "rescale":
        kw["nshifts"] = int(rng.integers(1, 5)); kw["seed"] = int(rng.integers(0, 100))
    elif mode == "raw_cov":
        kw["clr_weight_name"] = None; kw["coverage_norm"] = str(rng.choice(["total", "cis"])); kw["min_diag"] = int(rng.choice([0, 2]))
    elif mode.startswith("expected"):
        expected = tr_exp if kind == "trans" else exp_chrom
        if kind == "trans": view = view_chrom
        if mode == "expected_not_ooe": kw["ooe"] = False
    if kind in ("bedpe", "bed"):
        g = rng.random()
        if g < 0.25: kw["by_strand"] = True
        elif g < 0.45 and kind == "bedpe": kw["by_distance"] = True
        elif g < 0.6: kw["by_strand"] = True; kw["by_distance"] = True
        elif g < 0.7 and kind == "bed": kw["by_window"] = True
        if kw.get("by_strand") and rng.random() < 0.5: kw["flip_negative_strand"] = True
        if kw.get("by_strand") and kind == "bed" and rng.random() < 0.4: kw["ignore_group_order"] = True
    if rng.random() < 0.25: kw["store_stripes"] = True
    def run(mod, c):
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            return mod.pileup(c, feats.copy(), view_df=None if view is None else view.copy(), expected_df=None if expected is None else expected.copy(), **kw)
    try:
        want = run(ref, refshim.ShimCooler(clr))
    except Exception as e:
        try:
            run(coolpup, clr); print(k, kind, mode, "REF RAISED", type(e).__name__, "but mine ran", kw)
        except Exception as e2:
            pass
        continue
    W = kw["rescale_size"] if kw.get("rescale") else 2 * (kw["flank"] // clr.binsize) + 1
    try:
        rec = mg.record(want, W)
        buf = io.BytesIO(); np.savez(buf, **rec); buf.seek(0); z = np.load(buf)
        got = run(coolpup, clr)
        gu.compare(z, got, rtol=1e-11)
    except Exception as e:
        bad += 1
        print(k, kind, mode, "MISMATCH", type(e).__name__, str(e)[:300], json.dumps(kw, default=str))
print("done", N, "scenarios, mismatches:", bad)
