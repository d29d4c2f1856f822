"""Function-level drop-in for enstop/block_parallel_plsa.py (`plsa_fit`, :339-421).

The reference cuts X into n_row_blocks x n_col_blocks padded COO tiles so that numba threads can own
tiles (:373-403) and sums per-tile partial factors (:182-185).  On MI355X the ownership units are the
document row and the vocabulary column item, which is the same arithmetic with
a different summation tree; the tile counts are accepted and ignored (the reference's np.uint16 tile
sizes, :359-360, silently wrap above 65 535 rows per tile -- not reproduced).  Reproduced semantics:
no sample weights anywhere in the fit, stop test `change / |cur| < tolerance` only (:329-331).
"""
import numpy as np

from .engine import PLSA_STOP_NO_ZERO_ARM, default_flags
from .plsa import BlockParallelPLSA, plsa_fit as _plsa_fit  # noqa: F401


def plsa_fit(X, k, n_row_blocks=8, n_col_blocks=8, init="random", n_iter=100, n_iter_per_test=10,
             tolerance=0.001, e_step_thresh=1e-32, random_state=None, device=None, flags=None,
             return_info=False):
    flags = (default_flags() if flags is None else flags) | PLSA_STOP_NO_ZERO_ARM
    return _plsa_fit(X, k, np.ones(X.shape[0], np.float32), init, n_iter, n_iter_per_test, tolerance,
                     e_step_thresh, random_state, device=device, flags=flags, return_info=return_info)
