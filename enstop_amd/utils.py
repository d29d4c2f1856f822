"""Host-side helpers with the reference's names (enstop/utils.py): input standardisation and
sample-weight validation used by the estimators.  Not hot-path code."""
import numpy as np
from sklearn.preprocessing import normalize as _sk_normalize

from .engine import host_normalize_rows


def normalize(ndarray, axis=0):
    """L1-normalise a 2-D float64 array IN PLACE along `axis` (enstop/utils.py:8-41): sequential
    float64 marginal, division only where the marginal is positive."""
    if ndarray.ndim != 2 or axis not in (0, 1):
        raise ValueError("axis must be 0 or 1")
    if axis == 1 and ndarray.dtype == np.float64 and ndarray.flags.c_contiguous:
        host_normalize_rows(ndarray)
        return
    work = np.ascontiguousarray(ndarray.T if axis == 0 else ndarray, dtype=np.float64)
    host_normalize_rows(work)
    ndarray[...] = work.T if axis == 0 else work


def standardize_input(input_matrix):
    """Float inputs are L1 row-normalised, integer counts pass through (enstop/utils.py:276-280).
    The reference tests `dtype in (np.float32, np.float64, np.float, np.double)`; `np.float` no
    longer exists in NumPy >= 1.24, the intended float test is applied here."""
    if input_matrix.dtype in (np.float32, np.float64):
        return _sk_normalize(input_matrix, norm="l1")
    return input_matrix


# sample-weight validation: scikit-learn's own validator, as the reference imports it by default
# (enstop/plsa.py:9, enstop_.py:8; its vendored copy enstop/utils.py:285-335 is only a fallback for
# scikit-learn versions that predate the function)
try:
    from sklearn.utils.validation import _check_sample_weight  # noqa: E402,F401
except ImportError:          # a private name: keep working if a scikit-learn release moves it
    def _check_sample_weight(sample_weight, X, dtype=None):
        """Fallback with the contract of enstop/utils.py:285-335: None -> ones, a number -> constant vector,
        otherwise a 1-D float array of length n_samples."""
        n = X.shape[0]
        dtype = np.float64 if dtype is None else dtype
        if sample_weight is None:
            return np.ones(n, dtype=dtype)
        if isinstance(sample_weight, (int, float, np.integer, np.floating)):
            return np.full(n, sample_weight, dtype=dtype)
        sw = np.asarray(sample_weight, dtype=dtype)
        if sw.ndim != 1:
            raise ValueError("Sample weights must be 1D array or scalar")
        if sw.shape != (n,):
            raise ValueError("sample_weight.shape == {}, expected {}!".format(sw.shape, (n,)))
        return sw


# ------------------------------------------------------------------------------------------------
# topic-quality metrics (enstop/utils.py:44-273): host-side NumPy, not hot-path code
# ------------------------------------------------------------------------------------------------
def _empirical_probs(data):
    p = np.asarray(data.sum(axis=0)).squeeze().astype(np.float64)
    return p / p.sum()


def _log_lift(topics, z, empirical_probs, n=-1):
    words = np.arange(topics.shape[1]) if n <= 0 else np.argsort(topics[z])[-n:]
    ok = empirical_probs[words] > 0
    total = np.sum(topics[z, words][ok] * 1.0 / empirical_probs[words][ok])
    return np.log(total * 1.0 / len(words))


def log_lift(topics, z, data, n_words=-1):
    """log of the mean lift P(w|z) / P(w) over the topic's top `n_words` words (all when <= 0)."""
    t = np.array(topics, dtype=np.float64)
    normalize(t, axis=1)
    return _log_lift(t.astype(np.asarray(topics).dtype), z, _empirical_probs(data), n_words)


def mean_log_lift(topics, data, n_words=-1):
    """Mean over topics; like the reference it scores the topics AS GIVEN (enstop/utils.py:144 passes
    the un-normalised array) -- identical for the normalised topics the estimators produce."""
    e = _empirical_probs(data)
    return np.mean([_log_lift(np.asarray(topics), z, e, n_words) for z in range(np.asarray(topics).shape[0])])


def _coherence(topics, z, n, B, n_docs_per_word):
    top = np.argsort(topics[z])[-n:]
    sub = B[:, top]
    co = np.asarray((sub.T @ sub).todense(), dtype=np.float64)        # co-document counts of the top words
    total = 0.0
    for i in range(n - 1):
        w = top[i]
        if n_docs_per_word[w] == 0:
            continue
        total += np.sum(np.log((co[i, i + 1:] + 1.0) / n_docs_per_word[w]))
    return total


def _binarised(data):
    from scipy.sparse import csc_matrix, issparse
    B = data.tocsc() if issparse(data) else csc_matrix(data)
    B = B.copy()
    B.data = np.ones_like(B.data, dtype=np.float64)       # stored entries count as occurrences, as the
    n_docs = np.asarray((data > 0).sum(axis=0)).squeeze()  # reference's index intersection does
    return B, n_docs


def coherence(topics, z, data, n_words=20):
    """UMass-style coherence of topic z over its top `n_words` words (enstop/utils.py:155-197)."""
    B, n_docs = _binarised(data)
    return _coherence(np.asarray(topics), z, n_words, B, n_docs)


def mean_coherence(topics, data, n_words=20):
    B, n_docs = _binarised(data)
    topics = np.asarray(topics)
    return np.mean([_coherence(topics, z, n_words, B, n_docs) for z in range(topics.shape[0])])
