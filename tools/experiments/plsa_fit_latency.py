"""Wall time of the reference-shaped call PLSA(n_components=20, n_iter=50).fit(X) on the 20NG-shaped corpus (BASELINE
configs[0]), host work included, with a cProfile of one call."""
import cProfile, io, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import enstop_amd
from enstop_amd.engine import Engine
with Engine(0) as eng:
    eng.generate_synthetic(18_846, 173_762, 2_950_000, seed=0)
    X = eng.download_active_csr()
Xi = X.astype(np.int64)           # what CountVectorizer hands over
for name, A in (("int64 counts", Xi), ("float32 counts", X)):
    m = enstop_amd.PLSA(n_components=20, n_iter=50, tolerance=0.0, random_state=1)
    m.fit(A); m.fit(A)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); m.fit(A); ts.append(time.perf_counter() - t0)
    print("PLSA(k=20, 50 iterations).fit on %s: best %.2f ms, median %.2f ms" % (name, min(ts) * 1e3, sorted(ts)[2] * 1e3))
pr = cProfile.Profile(); pr.enable(); m.fit(Xi); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25); print(s.getvalue()[:5000])
