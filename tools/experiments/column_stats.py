"""Share of the non-zeros of the config-3 synthetic corpus by document frequency of their word: how much
of the column pass' P(z|d) gathering is inherently without reuse (rare words)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np
from enstop_amd.engine import Engine
with Engine(0) as eng:
    eng.generate_synthetic(1_000_000, 100_000, 100_000_000, seed=0)
    X = eng.download_active_csr()
n = X.shape[0]
cnt = np.bincount(X.indices, minlength=X.shape[1]).astype(np.int64)
tot = cnt.sum()
for f in (1e-5, 1e-4, 1e-3, 1e-2, 1e-1, 0.5, 1.0):
    sel = cnt <= f * n
    print("doc frequency <= %-7g : %6d words, %5.1f %% of the non-zeros" % (f, sel.sum(), 100.0 * cnt[sel].sum() / tot))
