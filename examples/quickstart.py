#!/usr/bin/env python3
"""Quick start: the enstop estimators on an MI355X (needs the built library and a gfx950 device).

A corpus with 12 planted topics is generated on the device (documents as sparse Dirichlet mixtures of latent topics, each
topic with its own Zipf ranking of the vocabulary), fitted with PLSA and with EnsembleTopics, and the documents' dominant
found topic is compared with the generator's ground truth -- the check the reference's notebook makes by eye on
20-Newsgroups (notebooks/EnsTop with 20-Newsgroups.ipynb)."""
import os
import sys
import time

import numpy as np
from scipy.optimize import linear_sum_assignment

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enstop_amd                                              # noqa: E402

K = 12
with enstop_amd.Engine() as eng:
    eng.generate_synthetic(20_000, 30_000, 1_600_000, seed=1, topics=K, alpha=0.05, background=0.1)
    X = eng.download_active_csr().astype(np.int64)            # counts: integer input is used as it is (utils.py:276-280)
    truth = eng.synthetic_dominant_topics()


def on_their_topic(assign, n_found):
    C = np.zeros((n_found, K), np.int64)
    np.add.at(C, (assign, truth), 1)
    r, c = linear_sum_assignment(-C)
    return C[r, c].sum() / float(len(truth))


t0 = time.perf_counter()
model = enstop_amd.PLSA(n_components=K, random_state=0).fit(X)
print("PLSA: %d iterations in %.3f s, %.1f %% of the documents on their planted topic, coherence %.3f"
      % (model.n_iter_, time.perf_counter() - t0, 100 * on_their_topic(model.embedding_.argmax(axis=1), K), model.coherence()))
held = model.transform(X[:2000])
print("transform of 2000 documents: rows sum to %.4f ... %.4f" % (held.sum(axis=1).min(), held.sum(axis=1).max()))
t0 = time.perf_counter()
ens = enstop_amd.EnsembleTopics(n_components=K, n_starts=16, topic_combination="hellinger", random_state=0)
emb = ens.fit_transform(X)
print("EnsembleTopics: 16 bootstrapped fits -> %d stable topics in %.3f s, %.1f %% of the documents on their planted topic"
      % (ens.n_components_, time.perf_counter() - t0, 100 * on_their_topic(emb.argmax(axis=1), ens.n_components_)))
