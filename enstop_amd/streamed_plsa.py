"""Function-level drop-in for enstop/streamed_plsa.py (`plsa_fit` :606-699, `plsa_refit` :959-1039).

The reference bounds memory by materialising P(z|w,d) for `block_size` non-zeros at a time
(:341-375); the fused HIP schedule keeps the responsibilities in registers, so `block_size` is
accepted and has no effect.  Reproduced semantics: sample weights as in plsa.py, stop test
`change / |cur| < tolerance` only (:596-597).
"""
from .engine import PLSA_FUSED, PLSA_STOP_NO_ZERO_ARM
from .plsa import StreamedPLSA, plsa_fit as _plsa_fit, plsa_refit as _plsa_refit  # noqa: F401


def plsa_fit(X, k, sample_weight, init="random", block_size=65536, n_iter=100, n_iter_per_test=10,
             tolerance=0.001, e_step_thresh=1e-32, random_state=None, device=None, return_info=False):
    return _plsa_fit(X, k, sample_weight, init, n_iter, n_iter_per_test, tolerance, e_step_thresh,
                     random_state, device=device, flags=PLSA_FUSED | PLSA_STOP_NO_ZERO_ARM,
                     return_info=return_info)


def plsa_refit(X, topics, sample_weight, block_size=65536, n_iter=50, n_iter_per_test=10,
               tolerance=0.005, e_step_thresh=1e-32, random_state=None, device=None):
    return _plsa_refit(X, topics, sample_weight, n_iter, n_iter_per_test, tolerance, e_step_thresh,
                       random_state, device=device, flags=PLSA_FUSED)
