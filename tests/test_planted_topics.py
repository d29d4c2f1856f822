"""End-to-end FUNCTION of the drop-in on a corpus with a known answer (needs a real MI355X: -m gpu).

Every other GPU test is a parity test (HIP == oracle == reference on the same numbers).  These ask the question a user
of the reference asks: does the thing find the topics?  The corpus comes from plsa_generate_synthetic_topics -- documents
are sparse Dirichlet mixtures of K0 latent topics with their own word rankings -- and the generator hands out the
ground truth (plsa_synthetic_dominant_topics).  The reference's own notebook makes the same kind of check by eye on
20-Newsgroups (notebooks/EnsTop with 20-Newsgroups.ipynb: newsgroup labels against the embedding)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

K0 = 8


@pytest.fixture(scope="module")
def amd():
    import enstop_amd
    return enstop_amd


@pytest.fixture(scope="module")
def planted(amd):
    with amd.Engine() as eng:
        eng.generate_synthetic(6000, 3000, 330_000, seed=3, topics=K0, alpha=0.05, background=0.1)
        X = eng.download_active_csr()
        labels = eng.synthetic_dominant_topics()
    X = X.astype(np.int64)                     # counts: the estimators leave integer input unnormalised (utils.py:276-280)
    assert labels.shape == (X.shape[0],) and set(np.unique(labels)) == set(range(K0))
    return X, labels


def matched_accuracy(assign, labels, n_found):
    """documents whose found topic is the one matched (Hungarian, maximum agreement) to their planted topic"""
    from scipy.optimize import linear_sum_assignment
    C = np.zeros((n_found, K0), np.int64)
    np.add.at(C, (assign, labels), 1)
    r, c = linear_sum_assignment(-C)
    return C[r, c].sum() / float(len(labels)), dict(zip(r.tolist(), c.tolist()))


def test_plsa_recovers_the_planted_topics(amd, planted):
    X, labels = planted
    model = amd.PLSA(n_components=K0, n_iter=100, random_state=0)
    emb = model.fit_transform(X)
    assert emb.shape == (X.shape[0], K0) and model.components_.shape == (K0, X.shape[1])
    np.testing.assert_allclose(emb.sum(axis=1), 1.0, atol=1e-4)
    acc, match = matched_accuracy(emb.argmax(axis=1), labels, K0)
    assert acc > 0.85, acc
    # every found topic's heaviest words are words the documents of its planted topic actually use most
    Xc = X.tocsc()
    for z, t in match.items():
        top = np.argsort(-model.components_[z])[:20]
        docs_t = np.flatnonzero(labels == t)
        mass_in = np.asarray(Xc[docs_t][:, top].sum()) / float(Xc[docs_t].sum())
        mass_all = np.asarray(Xc[:, top].sum()) / float(Xc.sum())
        assert mass_in > 2.0 * mass_all, (z, t, mass_in, mass_all)
    # held-out documents (transform = plsa_refit against the fitted topics, plsa.py:1184-1220)
    acc_t, _ = matched_accuracy(model.transform(X[:1500]).argmax(axis=1), labels[:1500], K0)
    assert acc_t > 0.85, acc_t


@pytest.mark.parametrize("combination", ["hellinger", "kl_divergence"])
def test_ensemble_topics_finds_the_planted_topics(amd, planted, combination):
    """EnsembleTopics end to end (enstop_.py:417-584): 12 bootstrapped fits with MORE topics than planted (12 > 8), the
    all-pairs divergence matrix, the HDBSCAN* leaf clusters, the cluster representatives, the refit of the documents.
    The stable topics must cover all planted topics, one cluster each (no planted topic split or merged)."""
    X, labels = planted
    et = amd.EnsembleTopics(n_components=12, n_starts=12, topic_combination=combination, n_iter=60, min_samples=3,
                            min_cluster_size=5, n_jobs=4, random_state=1)
    emb = et.fit_transform(X)
    m_found = et.n_components_
    assert et.components_.shape == (m_found, X.shape[1]) and emb.shape == (X.shape[0], m_found)
    np.testing.assert_allclose(et.components_.sum(axis=1), 1.0, atol=1e-3)
    assert K0 <= m_found <= K0 + 3, m_found
    acc, match = matched_accuracy(emb.argmax(axis=1), labels, m_found)
    assert len(set(match.values())) == K0 and acc > 0.8, (acc, match)


@pytest.mark.parametrize("combination", ["hellinger", "kl_divergence"])
def test_baseline_ensemble_configuration_on_a_20ng_shaped_corpus(amd, combination):
    """BASELINE.json configs[3] through the estimator itself: `EnsembleTopics(n_components=20, n_starts=32)` on a corpus of
    20-Newsgroups' shape (18 846 x 173 762, 2.9 M non-zeros) -- here with 20 PLANTED topics instead of newsgroups, so there
    is an answer to check: 32 bootstrapped 50-iteration fits, the 640 x 640 divergence matrix over 173 762 words, the
    HDBSCAN* leaf clusters, their representatives, the refit of all documents (enstop_.py:417-584).  Exactly the 20 planted
    topics come back, none split or merged, and > 90 % of the documents sit on their planted topic."""
    K = 20
    with amd.Engine() as eng:
        eng.generate_synthetic(18846, 173762, 2_950_000, seed=5, topics=K, alpha=0.05, background=0.1)
        X = eng.download_active_csr().astype(np.int64)
        labels = eng.synthetic_dominant_topics()
    et = amd.EnsembleTopics(n_components=K, n_starts=32, topic_combination=combination, n_iter=50, n_jobs=4, random_state=1)
    emb = et.fit_transform(X)
    found = et.n_components_
    assert et.components_.shape == (found, X.shape[1]) and emb.shape == (X.shape[0], found)
    np.testing.assert_allclose(et.components_.sum(axis=1), 1.0, atol=1e-3)
    from scipy.optimize import linear_sum_assignment
    C = np.zeros((found, K), np.int64)
    np.add.at(C, (emb.argmax(axis=1), labels), 1)
    r, c = linear_sum_assignment(-C)
    acc = C[r, c].sum() / float(len(labels))
    assert found == K and len(set(c.tolist())) == K and acc > 0.9, (found, acc)
