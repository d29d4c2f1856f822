"""Host-side helpers with the reference's names (enstop/utils.py): input standardisation and
sample-weight validation used by the estimators.  Not hot-path code."""
import numbers

import numpy as np
from sklearn.preprocessing import normalize as _sk_normalize
from sklearn.utils.validation import check_array

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


def _check_sample_weight(sample_weight, X, dtype=None):
    """sklearn's validator (the reference falls back to a vendored copy, enstop/utils.py:285-335):
    None -> ones, scalar -> full, array -> checked 1-D of length n_samples."""
    n_samples = X.shape[0]
    if dtype is not None and dtype not in (np.float32, np.float64):
        dtype = np.float64
    if sample_weight is None:
        return np.ones(n_samples, dtype=dtype)
    if isinstance(sample_weight, numbers.Number):
        return np.full(n_samples, sample_weight, dtype=dtype)
    sample_weight = check_array(sample_weight, accept_sparse=False, ensure_2d=False,
                                dtype=[np.float64, np.float32] if dtype is None else dtype, order="C")
    if sample_weight.ndim != 1:
        raise ValueError("Sample weights must be 1D array or scalar")
    if sample_weight.shape != (n_samples,):
        raise ValueError("sample_weight.shape == {}, expected {}!".format(sample_weight.shape, (n_samples,)))
    return sample_weight
