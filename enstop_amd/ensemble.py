"""EnsembleTopics on top of the GPU ensemble path (a "next" row).

On the MI355X engine: the bootstrapped pLSA fits (`ensemble_of_topics`), the all-pairs Hellinger and
KL matrices of the stacked topics (`plsa_all_pairs_hellinger`, `plsa_all_pairs_kl`), the cluster
representatives (`plsa_cluster_representatives`) and the final refit of the document vectors
(`plsa_refit`).  On the host: the tree step of HDBSCAN over the few hundred topic vectors.

Parity status.  Pinned by reference-generated goldens (tests/golden/combine_t24.npz, produced by the
reference's own statements, tests/golden/make_golden.py::gen_combine): `all_pairs_kl_divergence`
(enstop_.py:234-253), the mutual-reachability matrix of `generate_combined_topics_kl` (:283-296), and
the cluster representatives of all three combiners given labels / membership strengths (:299-308,
:340-345, :385-393).  **Unpinned by a run of the package itself**: what `hdbscan` computes (MST, single
linkage, condensed-tree leaf clusters) and what `umap` computes (the Hellinger metric, the embedding);
neither is installable in the build image.  The HDBSCAN* tree step is the product's own
(`hdbscan_tree.py`: Prim order, interval dendrogram, leaf clusters, hdbscan's conventions for core
distances and cluster numbering) for "kl_divergence" and "hellinger", checked against a restatement of
hdbscan's published routines and against scikit-learn's public estimator (tests/test_hdbscan_tree.py);
no private scikit-learn module is imported.  "hellinger_umap" needs
`umap-learn` and raises ImportError when it is absent.  The all-pairs Hellinger matrix follows
umap.distances.hellinger's published definition (umap-learn >= 0.3.8) and is checked against that
definition, not against a reference run.
"""
import numpy as np
from scipy.sparse import csr_matrix, issparse
from sklearn.base import BaseEstimator, TransformerMixin
from sklearn.utils import check_array, check_random_state

from .enstop_ import ensemble_of_topics
from .plsa import _TopicMetricsMixin, plsa_refit
from .utils import _check_sample_weight


def all_pairs_hellinger_distance(distributions):
    """sqrt(1 - BC(p, q) / sqrt(|p|_1 |q|_1)) for all pairs (umap.distances.hellinger's definition,
    which enstop_.py:258-266 applies pairwise); one small GEMM."""
    P = np.asarray(distributions, dtype=np.float64)
    root = np.sqrt(P)
    l1 = P.sum(axis=1)
    denom = np.sqrt(np.outer(l1, l1))
    with np.errstate(divide="ignore", invalid="ignore"):
        ratio = np.where(denom > 0, (root @ root.T) / denom, 0.0)
    D = np.sqrt(np.clip(1.0 - ratio, 0.0, None))
    both_empty = np.outer(l1 == 0, l1 == 0)
    one_empty = np.logical_xor.outer(l1 == 0, l1 == 0)
    D[both_empty] = 0.0
    D[one_empty] = 1.0
    np.fill_diagonal(D, 0.0)
    return D


def all_pairs_kl_divergence(distributions):
    """KL(p_i || p_j) in bits over the common support (enstop_.py:234-253), host NumPy float64
    definition; `ensemble_fit` uses the device kernel (`Engine.all_pairs_kl`) instead."""
    P = np.asarray(distributions, dtype=np.float64)
    with np.errstate(divide="ignore"):
        L = np.where(P > 0, np.log2(np.where(P > 0, P, 1.0)), 0.0)
    pos = (P > 0).astype(np.float64)
    # sum_w p_i[w] (log p_i[w] - log p_j[w]) restricted to p_i > 0 and p_j > 0
    return (P * L) @ pos.T - P @ L.T


def mutual_reachability_from_divergences(divergence_matrix, min_samples):
    """enstop_.py:283-296: the core divergence of a topic is its `min_samples`-th smallest divergence
    (row-wise sort, position `min_samples`, the zero self-divergence included); the mutual
    reachability is the elementwise maximum of D, D^T and both core divergences."""
    D = np.asarray(divergence_matrix, dtype=np.float64)
    core = np.sort(D, axis=1)[:, min_samples]
    tiled = np.tile(core, (core.shape[0], 1))
    return np.dstack([D, D.T, tiled, tiled.T]).max(axis=-1)


def _cluster_representatives(all_topics, labels, weights=None, engine=None):
    """enstop_.py:299-308 / 385-393.  With `engine` the means are formed on the device
    (`plsa_cluster_representatives`), else in NumPy with the reference's own expressions."""
    labels = np.asarray(labels)
    n_clusters = int(labels.max()) + 1 if labels.size else 0
    if engine is not None and n_clusters > 0:
        if weights is not None:     # np.average raises ZeroDivisionError on all-zero weights (enstop_.py:387)
            for i in range(n_clusters):
                if not np.any(np.asarray(weights)[labels == i] != 0):
                    raise ZeroDivisionError("Weights sum to zero, can't be normalized")
        return engine.cluster_representatives(all_topics, labels, weights)
    all_topics = np.asarray(all_topics)
    result = np.empty((max(n_clusters, 0), all_topics.shape[1]), dtype=np.float32)
    for i in range(n_clusters):
        mask = labels == i
        if weights is None:
            result[i] = np.mean(np.sqrt(all_topics[mask]), axis=0) ** 2
        else:
            result[i] = np.average(np.sqrt(all_topics[mask]), axis=0, weights=np.asarray(weights)[mask]) ** 2
        result[i] /= result[i].sum()
    return result


def labels_from_mutual_reachability(mutual_reachability, min_cluster_size):
    """enstop_.py:291-298 (hdbscan's `mst_linkage_core`, `label`, `_tree_to_labels(..., "leaf")`):
    (labels, membership strengths) from the product's own tree step, `hdbscan_tree.py`."""
    from .hdbscan_tree import labels_from_mutual_reachability as tree_step
    return tree_step(mutual_reachability, min_cluster_size)


def generate_combined_topics_kl(all_topics, min_samples=5, min_cluster_size=5, engine=None):
    """enstop_.py:256-310.  `engine`: run the all-pairs KL matrix and the representatives on the GPU."""
    D = engine.all_pairs_kl(all_topics) if engine is not None else all_pairs_kl_divergence(all_topics)
    mr = mutual_reachability_from_divergences(D, min_samples)
    labels, _ = labels_from_mutual_reachability(mr, min_cluster_size)
    return _cluster_representatives(all_topics, labels, engine=engine)


def generate_combined_topics_hellinger(all_topics, min_samples=5, min_cluster_size=5, distance_fn=None,
                                       engine=None):
    """enstop_.py:313-347.  `engine` / `distance_fn`: all-pairs Hellinger on the device
    (`Engine.all_pairs_hellinger`: 32 x 20 topics over 174 k words take 0.8 s in NumPy float64,
    milliseconds on the GPU); standalone calls on host arrays use the NumPy definition above."""
    if distance_fn is None:
        distance_fn = engine.all_pairs_hellinger if engine is not None else all_pairs_hellinger_distance
    D = distance_fn(all_topics)
    # hdbscan.HDBSCAN(metric="precomputed", cluster_selection_method="leaf").fit_predict (enstop_.py:340-345);
    # hdbscan's min_samples convention (core distance = min_samples-th smallest entry of a row, self included)
    from .hdbscan_tree import hdbscan_precomputed_leaf
    labels, _ = hdbscan_precomputed_leaf(D, min_samples, min_cluster_size)
    return _cluster_representatives(all_topics, labels, engine=engine)


def generate_combined_topics_hellinger_umap(all_topics, min_samples=5, min_cluster_size=5,
                                            n_neighbors=15, reduced_dim=5, engine=None):
    try:
        import umap
    except ImportError as e:
        raise ImportError('topic_combination="hellinger_umap" needs the umap-learn package; '
                          'use topic_combination="hellinger" instead') from e
    from sklearn.cluster import HDBSCAN
    embedding = umap.UMAP(n_neighbors=n_neighbors, n_components=reduced_dim,
                          metric="hellinger").fit_transform(all_topics)
    # scikit-learn's public estimator on the embedding; its min_samples counts the point itself, hdbscan's
    # (enstop_.py:388-392) does not: + 1 gives the same core distances
    clusterer = HDBSCAN(min_samples=min_samples + 1, min_cluster_size=min_cluster_size,
                        cluster_selection_method="leaf", allow_single_cluster=True).fit(embedding)
    return _cluster_representatives(all_topics, clusterer.labels_, clusterer.probabilities_, engine=engine)


_topic_combiner = {
    "kl_divergence": generate_combined_topics_kl,
    "hellinger": generate_combined_topics_hellinger,
    "hellinger_umap": generate_combined_topics_hellinger_umap,
}


def ensemble_fit(X, estimated_n_topics=10, model="plsa", init="random", min_samples=3,
                 min_cluster_size=4, n_starts=16, n_jobs=1, parallelism="dask",
                 topic_combination="hellinger_umap", bootstrap=True, n_iter=100, n_iter_per_test=10,
                 tolerance=0.001, e_step_thresh=1e-16, lift_factor=1, beta_loss=1, alpha=0.0,
                 solver="mu", random_state=None, device=None):
    """Stable topics from an ensemble of bootstrapped pLSA fits, then document vectors against them
    (enstop_.py:417-584).  Returns (doc_vectors [n_docs, M], stable_topics [M, n_words])."""
    if model not in ("plsa", "nmf"):
        raise ValueError('Model must be one of "plsa" or "nmf"')
    if topic_combination not in _topic_combiner:
        raise ValueError("topic_combination must be one of {}".format(tuple(_topic_combiner.keys())))
    X = check_array(X, accept_sparse="csr", dtype=np.float32)
    X = X.tocsr() if issparse(X) else csr_matrix(X, dtype=np.float32)
    all_topics = ensemble_of_topics(X, estimated_n_topics, model, n_jobs, n_starts, parallelism,
                                    init=init, n_iter=n_iter, n_iter_per_test=n_iter_per_test,
                                    tolerance=tolerance, e_step_thresh=e_step_thresh, bootstrap=bootstrap,
                                    random_state=random_state, device=device,
                                    **(dict(beta_loss=beta_loss, alpha=alpha, solver=solver) if model == "nmf" else {}))
    from .engine import get_engine
    stable_topics = _topic_combiner[topic_combination](all_topics, min_samples, min_cluster_size,
                                                       engine=get_engine(device))
    if stable_topics.shape[0] == 0:
        raise ValueError("topic combination found no stable topic cluster; lower min_samples / "
                         "min_cluster_size or raise n_starts")
    if lift_factor != 1:
        stable_topics = stable_topics.astype(np.float64) ** lift_factor
        stable_topics = (stable_topics / stable_topics.sum(axis=1, keepdims=True)).astype(np.float32)
    if model == "nmf":                          # enstop_.py:570-579, host scikit-learn like the reference
        from sklearn.decomposition import non_negative_factorization
        doc_vectors, _, _ = non_negative_factorization(
            X, H=np.asarray(stable_topics, dtype=X.dtype), n_components=stable_topics.shape[0],
            update_H=False, beta_loss=beta_loss, alpha_W=alpha, solver=solver)
        return doc_vectors, stable_topics
    sample_weight = _check_sample_weight(None, X, dtype=np.float32)
    doc_vectors = plsa_refit(X, stable_topics, sample_weight, e_step_thresh=e_step_thresh,
                             random_state=random_state, device=device)
    return doc_vectors, stable_topics


class EnsembleTopics(_TopicMetricsMixin, BaseEstimator, TransformerMixin):
    """Ensemble topic modelling with the reference estimator's constructor, methods and fitted
    attributes (enstop_.py:587-927): `components_`, `embedding_`, `training_data_`,
    `n_components_`.  `transform` passes unit sample weights to `plsa_refit` (the reference omits
    the argument and raises TypeError, enstop_.py:847-854)."""

    def __init__(self, n_components=10, model="plsa", init="random", n_starts=16, min_samples=3,
                 min_cluster_size=5, n_jobs=8, parallelism="dask", topic_combination="hellinger_umap",
                 bootstrap=True, n_iter=80, n_iter_per_test=10, tolerance=0.001, e_step_thresh=1e-32,
                 lift_factor=1, beta_loss=1, alpha=0.0, solver="mu", transform_random_seed=42,
                 random_state=None, device=None):
        self.n_components = n_components
        self.model = model
        self.init = init
        self.n_starts = n_starts
        self.min_samples = min_samples
        self.min_cluster_size = min_cluster_size
        self.n_jobs = n_jobs
        self.parallelism = parallelism
        self.topic_combination = topic_combination
        self.bootstrap = bootstrap
        self.n_iter = n_iter
        self.n_iter_per_test = n_iter_per_test
        self.tolerance = tolerance
        self.e_step_thresh = e_step_thresh
        self.lift_factor = lift_factor
        self.beta_loss = beta_loss
        self.alpha = alpha
        self.solver = solver
        self.transform_random_seed = transform_random_seed
        self.random_state = random_state
        self.device = device

    def fit(self, X, y=None):
        self.fit_transform(X)
        return self

    def fit_transform(self, X, y=None, **fit_params):
        X = check_array(X, accept_sparse="csr")
        if not issparse(X):
            X = csr_matrix(X)
        U, V = ensemble_fit(X, self.n_components, self.model, self.init, self.min_samples,
                            self.min_cluster_size, self.n_starts, self.n_jobs, self.parallelism,
                            self.topic_combination, self.bootstrap, self.n_iter, self.n_iter_per_test,
                            self.tolerance, self.e_step_thresh, self.lift_factor, self.beta_loss,
                            self.alpha, self.solver, self.random_state, device=self.device)
        self.components_ = V
        self.embedding_ = U
        self.training_data_ = X
        self.n_components_ = self.components_.shape[0]
        return U

    def transform(self, X, y=None):
        X = check_array(X, accept_sparse="csr")
        random_state = check_random_state(self.transform_random_seed)
        X = csr_matrix(X) if not issparse(X) else X.tocsr()
        sample_weight = _check_sample_weight(None, X, dtype=np.float32)
        return plsa_refit(X, self.components_, sample_weight, n_iter=50, n_iter_per_test=5,
                          tolerance=0.001, random_state=random_state, device=self.device)
