import time, numpy as np, sys
sys.path.insert(0, '/root/repo')
import enstop_amd as amd
from scipy.optimize import linear_sum_assignment
K0 = 20
with amd.Engine() as eng:
    eng.generate_synthetic(18846, 173762, 2_950_000, seed=5, topics=K0, alpha=0.05, background=0.1)
    X = eng.download_active_csr(); labels = eng.synthetic_dominant_topics()
X = X.astype(np.int64)
print(X.shape, X.nnz, np.bincount(labels))
for comb in ("hellinger", "kl_divergence"):
    t0 = time.time()
    et = amd.EnsembleTopics(n_components=20, n_starts=32, topic_combination=comb, n_iter=50, n_jobs=4, random_state=1)
    emb = et.fit_transform(X)
    dt = time.time() - t0
    mf = et.n_components_
    C = np.zeros((mf, K0), np.int64); np.add.at(C, (emb.argmax(axis=1), labels), 1)
    r, c = linear_sum_assignment(-C)
    print(comb, "found", mf, "acc", C[r, c].sum() / len(labels), "distinct planted matched", len(set(c.tolist())), "time %.2f s" % dt, flush=True)
