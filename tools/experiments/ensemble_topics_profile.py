#!/usr/bin/env python3
"""Wall time of EnsembleTopics.fit_transform on the 20NG-shaped corpus (config 4: 32 members x 50 iterations,
k = 20) for the two combiners that run without umap, with a cProfile of the slower one."""
import cProfile
import io
import os
import pstats
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np            # noqa: E402
import enstop_amd             # noqa: E402
from enstop_amd.engine import Engine   # noqa: E402

with Engine(0) as eng:
    eng.generate_synthetic(18_846, 173_762, 2_950_000, seed=0)
    X = eng.download_active_csr()
X = X.astype(np.int64)
for comb in ("hellinger", "kl_divergence"):
    m = enstop_amd.EnsembleTopics(n_components=20, n_starts=32, topic_combination=comb, n_iter=50, random_state=3)
    pr = cProfile.Profile()
    try:
        m.fit(X[:4000])                                       # warm-up: contexts, buffers
        t0 = time.perf_counter()
        pr.enable()
        emb = m.fit_transform(X)
        pr.disable()
    except ValueError as e:                                   # a corpus without topic structure may yield no cluster
        print("EnsembleTopics(%s): %s" % (comb, e))
        continue
    dt = time.perf_counter() - t0
    print("EnsembleTopics(%s).fit_transform: %.3f s, %d stable topics, embedding %s" % (comb, dt, m.n_components_, emb.shape))
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(14)
print(s.getvalue()[:3000])
