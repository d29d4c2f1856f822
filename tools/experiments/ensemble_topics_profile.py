import os, sys, time, cProfile, pstats, io
sys.path.insert(0, "/root/repo")
import numpy as np
import enstop_amd
from enstop_amd.engine import Engine
with Engine(0) as eng:
    eng.generate_synthetic(18_846, 173_762, 2_950_000, seed=0)
    X = eng.download_active_csr()
X.data = np.round(X.data).astype(np.int64).astype(np.float64); X = X.astype(np.int64)
m = enstop_amd.EnsembleTopics(n_components=20, n_starts=32, topic_combination="hellinger", n_iter=50, random_state=3)
m.fit(X[:2000])
pr = cProfile.Profile(); t0 = time.perf_counter(); pr.enable()
emb = m.fit_transform(X)
pr.disable(); dt = time.perf_counter() - t0
print("EnsembleTopics.fit_transform: %.2f s, %d stable topics, embedding %s" % (dt, m.n_components_, emb.shape))
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(18); print(s.getvalue()[:3500])
