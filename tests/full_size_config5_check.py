#!/usr/bin/env python3
"""BASELINE.json configs[4] at full size (5 M docs x 200 k vocab, 500 M nnz, k = 128; the 256 GB
materialised P array included) through the size-independent properties of
test_hip_parity.py::test_full_size_properties.  Takes minutes and ~270 GB of HBM: run by hand on the
GPU box (python tests/full_size_config5_check.py), not collected by pytest."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import enstop_amd                                              # noqa: E402
from test_hip_parity import test_full_size_properties         # noqa: E402

t0 = time.time()
test_full_size_properties(enstop_amd, (5_000_000, 200_000, 500_000_000, 128))
print("config 5 full-size properties ok (%.0f s)" % (time.time() - t0))
