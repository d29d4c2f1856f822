"""enstop_amd -- MI355X-native pLSA EM engine behind the lmcinnes/enstop interface.

    from enstop_amd import PLSA, plsa_fit, plsa_refit, plsa_topics, ensemble_of_topics

Importing is possible anywhere; any numerical call needs libplsa_hip.so (python -m
enstop_amd.build) and a gfx950 device -- there is no CPU code path in this package.
"""
from .plsa import (PLSA, StreamedPLSA, BlockParallelPLSA, GPUPLSA, DistributedPLSA, log_likelihood, plsa_e_step, plsa_fit, plsa_fit_inner, plsa_init,
                   plsa_m_step, plsa_m_step_w_sample_weight, plsa_refit, plsa_refit_inner,
                   plsa_refit_m_step)
from .enstop_ import ensemble_of_topics, plsa_topics
from .ensemble import EnsembleTopics, ensemble_fit
from .sharded import sharded_plsa_fit
from .engine import Engine, DeviceError, PLSA_FUSED, PLSA_REFERENCE_SUMS, PLSA_REFERENCE_LL
from .utils import log_lift, mean_log_lift, coherence, mean_coherence
from . import comm, distributed, engine       # enstop_amd.distributed.init() works without a separate import

__all__ = ["PLSA", "StreamedPLSA", "BlockParallelPLSA", "GPUPLSA", "DistributedPLSA", "log_lift", "mean_log_lift", "coherence", "mean_coherence", "plsa_fit", "plsa_refit", "plsa_fit_inner", "plsa_refit_inner", "plsa_init",
           "plsa_e_step", "plsa_m_step", "plsa_m_step_w_sample_weight", "plsa_refit_m_step",
           "log_likelihood", "plsa_topics", "ensemble_of_topics", "EnsembleTopics", "ensemble_fit", "sharded_plsa_fit", "Engine", "DeviceError",
           "PLSA_FUSED", "PLSA_REFERENCE_SUMS", "PLSA_REFERENCE_LL"]
