"""Host-side topic combination of the EnsembleTopics shell (parity unpinned: the reference's hdbscan /
umap stage cannot run here).  Checks the divergence matrices against direct definitions and that
planted topic clusters are recovered."""
import numpy as np
import pytest

from enstop_amd import ensemble


def _planted(n_base=4, copies=8, m=60, seed=0):
    rs = np.random.RandomState(seed)
    base = rs.dirichlet(np.full(m, 0.05), size=n_base)
    topics = []
    for b in base:
        for _ in range(copies):
            t = b * np.exp(0.05 * rs.randn(m)) + 1e-6
            topics.append(t / t.sum())
    return base, np.array(topics, np.float32)


def test_hellinger_matches_definition():
    _, T = _planted()
    D = ensemble.all_pairs_hellinger_distance(T)
    rs = np.random.RandomState(1)
    for _ in range(20):
        i, j = rs.randint(0, T.shape[0], 2)
        x, y = T[i].astype(np.float64), T[j].astype(np.float64)
        ref = np.sqrt(max(0.0, 1 - np.sum(np.sqrt(x * y)) / np.sqrt(x.sum() * y.sum()))) if i != j else 0.0
        assert abs(D[i, j] - ref) < 1e-9
    assert np.allclose(D, D.T) and np.all(np.diag(D) == 0)
    Z = np.vstack([T[:2], np.zeros((1, T.shape[1]), np.float32)])
    DZ = ensemble.all_pairs_hellinger_distance(Z)
    assert DZ[2, 2] == 0.0 and DZ[0, 2] == 1.0 and DZ[2, 1] == 1.0


def test_kl_matches_definition():
    _, T = _planted(seed=3)
    T = T.copy(); T[0, :5] = 0; T[0] /= T[0].sum()
    K = ensemble.all_pairs_kl_divergence(T)
    for i, j in ((0, 1), (1, 0), (5, 17), (3, 3)):
        a, b = T[i].astype(np.float64), T[j].astype(np.float64)
        mask = (a > 0) & (b > 0)
        ref = np.sum(a[mask] * (np.log2(a[mask]) - np.log2(b[mask])))
        assert abs(K[i, j] - ref) < 1e-9


@pytest.mark.parametrize("how", ["hellinger", "kl_divergence"])
def test_planted_clusters_recovered(how):
    base, T = _planted(n_base=4, copies=8)
    stable = ensemble._topic_combiner[how](T, 3, 4)
    assert stable.shape == (4, T.shape[1]) and stable.dtype == np.float32
    np.testing.assert_allclose(stable.sum(1), 1.0, atol=1e-5)
    D = ensemble.all_pairs_hellinger_distance(np.vstack([base, stable]))[:4, 4:]
    assert sorted(D.argmin(axis=1)) == [0, 1, 2, 3] and D.min(axis=1).max() < 0.08


def test_umap_combiner_reports_missing_dependency():
    try:
        import umap  # noqa: F401
        pytest.skip("umap is installed")
    except ImportError:
        pass
    _, T = _planted()
    with pytest.raises(ImportError, match="umap"):
        ensemble.generate_combined_topics_hellinger_umap(T)


def test_constructor_signature_matches_reference():
    p = ensemble.EnsembleTopics().get_params()
    ref = dict(n_components=10, model="plsa", init="random", n_starts=16, min_samples=3, min_cluster_size=5,
               n_jobs=8, parallelism="dask", topic_combination="hellinger_umap", bootstrap=True, n_iter=80,
               n_iter_per_test=10, tolerance=0.001, e_step_thresh=1e-32, lift_factor=1, beta_loss=1, alpha=0.0,
               solver="mu", transform_random_seed=42, random_state=None)
    for k, v in ref.items():
        assert p[k] == v
    with pytest.raises(ValueError, match="topic_combination"):
        ensemble.ensemble_fit(np.eye(4), topic_combination="nope")


# ---- reference-generated goldens (tests/golden/combine_t24.npz, make_golden.py::gen_combine) ----------
def _golden():
    from conftest import load_golden
    return load_golden("combine_t24")


def test_host_kl_matches_reference_golden():
    g = _golden()
    D = ensemble.all_pairs_kl_divergence(g["topics"])
    assert np.abs(D - g["kl_f64_input"]).max() < 1e-12
    assert np.abs(D - g["kl_f32_input"]).max() < 1e-5 * np.abs(D).max()


def test_mutual_reachability_matches_reference_golden():
    """enstop_.py:283-296, captured at the reference's call of mst_linkage_core: bit-exact."""
    g = _golden()
    mr = ensemble.mutual_reachability_from_divergences(g["kl_f32_input"], int(g["min_samples"]))
    np.testing.assert_array_equal(mr, g["mutual_reachability"])


def test_cluster_representatives_match_reference_golden():
    """enstop_.py:299-308, 340-345, 385-393 given labels / membership strengths: bit-exact on the host."""
    g = _golden()
    for key, w in (("rep_kl", None), ("rep_hellinger", None), ("rep_umap", g["probabilities"])):
        np.testing.assert_array_equal(ensemble._cluster_representatives(g["topics"], g["labels"], w), g[key])


def test_kl_pipeline_recovers_planted_clusters():
    base, T = _planted(n_base=4, copies=8, seed=11)
    stable = ensemble.generate_combined_topics_kl(T, min_samples=3, min_cluster_size=4)
    assert stable.shape[0] == 4
    D = ensemble.all_pairs_hellinger_distance(np.vstack([base, stable]))[:4, 4:]
    assert np.all(D.min(axis=1) < 0.1) and len(set(D.argmin(axis=1))) == 4


def test_nmf_model_is_delegated_to_sklearn():
    import scipy.sparse as sp
    from enstop_amd.enstop_ import ensemble_of_topics, nmf_topics
    X = sp.random(80, 50, density=0.2, format="csr", random_state=0)
    V = nmf_topics(X, 4, random_state=1, init="nndsvda")
    assert V.shape == (4, 50) and np.allclose(V.sum(axis=1), 1.0)
    S = ensemble_of_topics(X, 4, model="nmf", n_runs=3, random_state=np.random.RandomState(2), init="nndsvda")
    assert S.shape == (12, 50)
    with pytest.raises(ValueError):
        ensemble_of_topics(X, 4, model="lda")
