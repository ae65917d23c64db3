"""Dev tool: the library's legacy-generator draws against numpy's own (numbers, state, time).  Run on the GPU box."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from coolpuppy_amd.engine import legacy_randint  # noqa: E402

legacy_randint(0, 2, 5000)
for rep in range(3):
    for seed, (lo, hi), m in [(0, (100000, 1000000), 10_000_000), (1, (0, 2), 10_000_000)]:
        np.random.seed(seed)
        st0 = np.random.get_state()
        t = time.time(); a = np.random.randint(lo, hi, m); t1 = time.time() - t
        sa = np.random.get_state()
        np.random.set_state(st0)
        t = time.time(); b = legacy_randint(lo, hi, m); t2 = time.time() - t
        sb = np.random.get_state()
        print(seed, (lo, hi), m, np.array_equal(a, b) and np.array_equal(sa[1], sb[1]) and sa[2] == sb[2],
              f"numpy {t1 * 1e3:.1f} ms  library {t2 * 1e3:.1f} ms", flush=True)
