"""Dev probe (GPU box): wall time of the FIRST pileup() on a table (upload + index build + coordinates + pile-up), with the
table upload prefetched on a helper thread (default) and without (COOLPUPPY_AMD_NO_PREFETCH=1)."""
import os, sys, time, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coolpuppy_amd import coolpup
import synth
from coolpuppy_amd.engine import PileupEngine
warnings.simplefilter("ignore")
hg = synth.make_cooler({c: synth.HG38[c] for c in synth.HG38}, binsize=10_000, lam=4200, seed=1000, name="synthetic_hg38_10kb", parallel=True)
feats = synth.random_cis_pairs(hg, 1_000_000, seed=42, strands=True)
PileupEngine(0).close()                                  # HIP runtime / context creation is not what is being compared
t = time.perf_counter()
df = coolpup.pileup(hg, feats, features_format="bedpe", flank=100_000, nshifts=10, seed=0)
first = time.perf_counter() - t
t = time.perf_counter()
coolpup.pileup(hg, feats, features_format="bedpe", flank=100_000, nshifts=10, seed=0)
print(f"prefetch={'off' if os.environ.get('COOLPUPPY_AMD_NO_PREFETCH') == '1' else 'on'}: first call {first:.3f}s, second {time.perf_counter() - t:.3f}s, n={int(df['n'].iloc[0])}")
