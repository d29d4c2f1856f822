"""CPU-only checks of the drop-in boundary: the C-ABI library loads, exports every symbol that
include/plsa_hip.h declares, and the host-side logic (initialisation, COO->CSR staging, estimator
validation, ensemble seeding) behaves like the reference.  No device computation here."""
import os
import re

import numpy as np
import pytest
import scipy.sparse as sp

from conftest import ROOT, load_golden, golden_csr


def _declared_symbols(headers=("plsa_hip.h", "plsa_hip_diag.h")):
    """include/plsa_hip.h is the drop-in boundary; include/plsa_hip_diag.h holds diagnostics / measurement / test plumbing of
    the same library (round 6 split).  Both are bound by enstop_amd/_lib.py and exported by libplsa_hip.so."""
    out = set()
    for h in headers:
        text = open(os.path.join(ROOT, "include", h)).read()
        text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
        out |= set(re.findall(r"\b(plsa_[a-z0-9_]+)\s*\(", text))
    return sorted(out)


def test_the_drop_in_header_is_small_and_free_of_diagnostics():
    """What a maintainer binds for the enstop_.py:52-53 seam: at most 40 entries, none of them a probe, a timer, a generator
    or a schedule report (those live in plsa_hip_diag.h), and the two headers do not overlap."""
    main, diag = _declared_symbols(("plsa_hip.h",)), _declared_symbols(("plsa_hip_diag.h",))
    assert len(main) <= 40, len(main)
    assert not set(main) & set(diag)
    for word in ("timing", "measure", "synthetic", "schedule_info", "placement", "hw_queues", "marginals", "device_info"):
        assert not [s for s in main if word in s], word
    for needed in ("plsa_create", "plsa_upload_csr", "plsa_set_factors", "plsa_fit", "plsa_refit", "plsa_get_factors",
                   "plsa_e_step", "plsa_m_step", "plsa_log_likelihood", "plsa_bootstrap", "plsa_set_arithmetic",
                   "plsa_comm_init", "plsa_comm_allgather_stack_to", "plsa_destroy", "plsa_last_error"):
        assert needed in main, needed


def test_header_symbols_exported_and_bound():
    from enstop_amd import _lib
    lib = _lib.load()
    declared = _declared_symbols()
    assert len(declared) >= 24
    for name in declared:
        assert hasattr(lib, name), "libplsa_hip.so does not export %s" % name
        assert name in _lib.SIGNATURES, "ctypes binding lacks %s" % name
    assert sorted(_lib.SIGNATURES) == declared


def test_every_entry_point_is_documented_with_the_interface_it_replaces():
    """INTEGRATION.md names every symbol of include/plsa_hip.h next to the reference interface it stands for, and every
    symbol of include/plsa_hip_diag.h as plumbing / measurement without a counterpart; the library links RCCL directly."""
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    missing = [s for s in _declared_symbols() if s not in doc and s.replace("plsa_timing_", "_") not in doc]
    assert not missing, missing
    import subprocess
    needed = subprocess.run(["readelf", "-d", os.path.join(ROOT, "enstop_amd", "libplsa_hip.so")],
                            capture_output=True, text=True).stdout
    assert "librccl.so" in needed and "libamdhip64.so" in needed


def test_library_has_gfx950_code_object():
    data = open(os.path.join(ROOT, "enstop_amd", "libplsa_hip.so"), "rb").read()
    assert b"gfx950" in data and b"k_e_step" in data and b"k_row_pass" in data


def test_python_loader_sets_a_default_hardware_queue_count_the_library_itself_never_does():
    """enstop_amd/_lib.py asks the HIP runtime for 8 hardware queues before it loads the library (concurrent ensemble
    members on one device must not serialise on the default 4) -- unless GPU_MAX_HW_QUEUES is already set or the user
    opted out with ENSTOP_AMD_HW_QUEUES=0.  Loading libplsa_hip.so by itself (a C host) leaves the environment alone
    (rounds 3-4 set the variable from a constructor of the library)."""
    import subprocess
    import sys
    probe = ("libc = ctypes.CDLL(None); libc.getenv.restype = ctypes.c_char_p; v = libc.getenv(b'GPU_MAX_HW_QUEUES'); "
             "print(v.decode() if v else 'unset')")
    code = ("import ctypes, os, sys; sys.path.insert(0, %r); from enstop_amd import _lib; _lib.load(); " % ROOT) + probe + \
           "; print(_lib.hw_queues()['set_by'], _lib.hw_queues()['value'])"
    raw = ("import ctypes; L = ctypes.CDLL(%r); " % os.path.join(ROOT, "enstop_amd", "libplsa_hip.so")) + probe + \
          "; print(L.plsa_hw_queues())"
    env = {k: v for k, v in os.environ.items() if k not in ("GPU_MAX_HW_QUEUES", "ENSTOP_AMD_HW_QUEUES")}

    def run(c, e):
        return subprocess.run([sys.executable, "-c", c], env=e, capture_output=True, text=True).stdout.split()

    assert run(code, env) == ["8", "enstop_amd", "8"]
    assert run(code, dict(env, GPU_MAX_HW_QUEUES="2")) == ["2", "user", "2"]
    assert run(code, dict(env, ENSTOP_AMD_HW_QUEUES="0")) == ["unset", "opt-out", "4"]
    assert run(code, dict(env, ENSTOP_AMD_HW_QUEUES="6")) == ["6", "enstop_amd", "6"]
    assert run(raw, env) == ["unset", "4"]


def test_no_cpu_fallback_without_device():
    """Without a GPU the product must fail loudly, not compute on the host."""
    from enstop_amd import _lib, plsa_fit
    from enstop_amd.engine import DeviceError
    import ctypes as C
    cnt = C.c_int(0)
    _lib.load().plsa_device_count(C.byref(cnt))
    if cnt.value > 0:
        pytest.skip("a HIP device is present")
    X = sp.random(20, 30, density=0.2, format="csr", random_state=0)
    with pytest.raises(DeviceError):
        plsa_fit(X, 4, np.ones(20, np.float32), n_iter=2, random_state=0)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "enstop_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in src.lower(), "%s mentions the oracle" % f


@pytest.mark.parametrize("case", ["fit_k8_tol0", "fit_k6_tupleinit", "fit_k20_50it"])
def test_plsa_init_matches_reference(case):
    """plsa_init + float32 casts (plsa.py:707-710) reproduce the reference's initial factors."""
    from enstop_amd import plsa_init
    g = load_golden(case)
    X = golden_csr(g)
    if "U_init" in g:
        U, V = plsa_init(X, int(g["k"]), init=(g["U_init"], g["V_init"]))
    else:
        U, V = plsa_init(X, int(g["k"]), init="random", rng=np.random.RandomState(int(g["fit_seed"])))
    assert U.dtype == np.float64 and V.dtype == np.float64
    np.testing.assert_array_equal(U.astype(np.float32), g["U0"])
    np.testing.assert_array_equal(V.astype(np.float32), g["V0"])


def test_plsa_init_nndsvd_matches_reference():
    """plsa_init(init="nndsvd") against the reference's own loop (plsa.py:458-491) on the same randomized SVD: both
    sides call scikit-learn's public randomized_svd on NumPy's global stream, seeded alike."""
    from enstop_amd import plsa_init
    g = load_golden("init_nndsvd_k6")
    X = golden_csr(g).astype(np.float64)             # the generator handed the reference a float64 matrix
    np.random.seed(int(g["numpy_seed"]))
    U, V = plsa_init(X, int(g["k"]), init="nndsvd")
    np.testing.assert_allclose(U, g["U"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(V, g["V"], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(U.sum(axis=1), 1.0, atol=1e-12)


def test_no_private_sklearn_modules_in_the_product():
    """scikit-learn's underscore modules are not an interface (`_hdbscan._linkage`, `_nmf._initialize_nmf`)."""
    import re
    pkg = os.path.join(ROOT, "enstop_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            bad = re.findall(r"(?:from|import)\s+sklearn(?:\.\w+)*\._\w+", src)
            assert not bad, (f, bad)


def test_plsa_init_errors():
    from enstop_amd import plsa_init
    X = sp.random(5, 7, density=0.5, format="csr", random_state=0)
    with pytest.raises(ValueError, match="Unrecognized init"):
        plsa_init(X, 3, init="bogus")
    with pytest.raises(ValueError):
        plsa_init(X, 3, init=(np.ones((4, 3)), np.ones((3, 7))))


def test_normalize_matches_oracle(oracle):
    from enstop_amd.utils import normalize
    rs = np.random.RandomState(3)
    a = rs.rand(37, 53)
    a[5] = 0.0
    b = a.copy()
    normalize(a, axis=1)
    oracle.normalize(b, axis=1)
    np.testing.assert_array_equal(a, b)
    c = rs.rand(11, 9)
    d = c.copy()
    normalize(c, axis=0)
    np.testing.assert_allclose(c.sum(axis=0), 1.0, rtol=1e-12)
    np.testing.assert_allclose(c, d / d.sum(axis=0), rtol=1e-12)


def test_coo_staging_sorted_and_unsorted():
    from enstop_amd.plsa import _coo_to_csr
    X = sp.random(30, 40, density=0.2, format="csr", random_state=1)
    A = X.tocoo()
    csr, order = _coo_to_csr(A.row, A.col, A.data, 30, 40)
    assert order is None
    assert (csr != X.astype(np.float32)).nnz == 0
    perm = np.random.RandomState(0).permutation(A.nnz)
    csr2, order2 = _coo_to_csr(A.row[perm], A.col[perm], A.data[perm], 30, 40)
    assert order2 is not None
    assert (csr2 != X.astype(np.float32)).nnz == 0
    np.testing.assert_array_equal(A.row[perm][order2], np.sort(A.row))


def test_estimator_validation_like_reference():
    from enstop_amd import PLSA
    X = np.abs(np.random.RandomState(0).randn(6, 5))
    X[2, 3] = -1.0
    with pytest.raises(ValueError, match="non-negative"):
        PLSA(n_components=2).fit(sp.csr_matrix(X.astype(np.int64) * 0 + np.where(X < 0, -1, 1)))
    params = PLSA().get_params()
    for key, val in dict(n_components=10, init="random", n_iter=100, n_iter_per_test=10, tolerance=0.001,
                         e_step_thresh=1e-32, transform_random_seed=42, random_state=None).items():
        assert params[key] == val


def test_estimators_follow_the_scikit_learn_protocol():
    """What pipelines, grid searches and joblib do to an estimator of the reference (BaseEstimator + TransformerMixin,
    plsa.py:1000, enstop_.py:587): `clone` rebuilds it from `get_params`, and it pickles -- unfitted and with fitted
    attributes; no device handle lives on the object."""
    import pickle
    from sklearn.base import clone
    import enstop_amd
    for cls in (enstop_amd.PLSA, enstop_amd.EnsembleTopics, enstop_amd.StreamedPLSA, enstop_amd.BlockParallelPLSA,
                enstop_amd.GPUPLSA):
        est = cls(n_components=7)
        assert clone(est).get_params() == est.get_params()
        assert pickle.loads(pickle.dumps(est)).get_params() == est.get_params()
        assert hasattr(est, "fit_transform") and hasattr(est, "transform")
    fitted = enstop_amd.PLSA(n_components=3)
    fitted.components_ = np.full((3, 5), 0.2, np.float32)
    fitted.embedding_ = np.full((4, 3), 1 / 3, np.float32)
    back = pickle.loads(pickle.dumps(fitted))
    np.testing.assert_array_equal(back.components_, fitted.components_)


def test_standardize_input():
    from enstop_amd.utils import standardize_input
    Xi = sp.csr_matrix(np.arange(12).reshape(3, 4))
    assert standardize_input(Xi) is Xi
    Xf = sp.csr_matrix(np.arange(12, dtype=np.float64).reshape(3, 4) + 1)
    np.testing.assert_allclose(np.asarray(standardize_input(Xf).sum(axis=1)).ravel(), 1.0)


def test_row_ranges_by_nnz():
    from enstop_amd.sharded import row_ranges_by_nnz
    indptr = np.array([0, 10, 10, 30, 35, 80, 100], np.int32)
    for parts in (1, 2, 3, 4, 6, 9):
        r = row_ranges_by_nnz(indptr, parts)
        assert len(r) == parts and r[0][0] == 0 and r[-1][1] == 6
        assert all(r[i][1] == r[i + 1][0] for i in range(parts - 1)) and all(a <= b for a, b in r)
        if parts <= 6:
            assert all(a < b for a, b in r)
    r = row_ranges_by_nnz(indptr, 2)
    assert abs((indptr[r[0][1]] - indptr[r[0][0]]) - 50) <= 45
    # every shard gets at least one row whenever there are enough rows (a rank without rows would leave
    # its peers waiting in the all-reduce): one row holding nearly all non-zeros, more parts than "mass"
    skew = np.array([0, 1, 2, 1000, 1001, 1002, 1003], np.int32)
    for parts in (2, 3, 4, 6):
        r = row_ranges_by_nnz(skew, parts)
        assert all(b > a for a, b in r) and r[0][0] == 0 and r[-1][1] == 6, r
    # fewer rows than parts: some range is necessarily empty (sharded_plsa_fit raises before any exchange)
    r = row_ranges_by_nnz(np.array([0, 5, 9], np.int32), 4)
    assert len(r) == 4 and any(b <= a for a, b in r)


def test_topic_metrics_match_reference():
    """coherence / log_lift (host-side) against values computed by the reference's enstop/utils.py."""
    from enstop_amd import utils
    g = load_golden("metrics")
    X = golden_csr(g)
    T = g["topics"]
    for z in range(T.shape[0]):
        np.testing.assert_allclose(utils.coherence(T, z, X, n_words=10), g["coherence"][z], rtol=1e-10)
        np.testing.assert_allclose(utils.log_lift(T, z, X, n_words=10), g["log_lift"][z], rtol=1e-6)
        np.testing.assert_allclose(utils.log_lift(T, z, X), g["log_lift_allwords"][z], rtol=1e-6)
    np.testing.assert_allclose(utils.mean_coherence(T, X, n_words=10), g["mean_coherence"], rtol=1e-10)
    np.testing.assert_allclose(utils.mean_log_lift(T, X, n_words=10), g["mean_log_lift"], rtol=1e-6)
    from enstop_amd import PLSA
    model = PLSA(n_components=T.shape[0])
    model.components_, model.training_data_ = T, X
    np.testing.assert_allclose(model.coherence(n_words=10), g["mean_coherence"], rtol=1e-10)
    np.testing.assert_allclose(model.log_lift(2, n_words=10), g["log_lift"][2], rtol=1e-6)
    with pytest.raises(ValueError):
        model.coherence(topic_num=99)
    with pytest.raises(ValueError):
        model.log_lift(topic_num="a")


@pytest.mark.parametrize("seed,advance,log2_blocks", [(42, 0, 0), (42, 1000, 1), (7, 5000, 3), (123, 700, 11), (5, 624, 17)])
def test_mt19937_block_jump_matches_numpy(seed, advance, log2_blocks):
    """csrc/mt_jump.hpp: the characteristic polynomial recovered by Berlekamp-Massey and the jump
    polynomials x^(624 * 2^b) mod phi move a RandomState key exactly as generating the outputs does."""
    import ctypes as C
    from enstop_amd import _lib
    lib = _lib.load()
    rs = np.random.RandomState(seed)
    if advance:
        rs.bytes(4 * advance)
    state = rs.get_state()
    key = np.array(state[1], dtype=np.uint32)
    assert lib.plsa_host_mt19937_jump(key.ctypes.data_as(C.POINTER(C.c_uint32)), log2_blocks) == 0
    chunk = 624 * 4 * 4096
    left = 624 * 4 * (1 << log2_blocks)
    while left:
        take = min(left, chunk)
        rs.bytes(take)
        left -= take
    after = rs.get_state()
    assert after[2] == state[2]
    want = np.array(after[1], dtype=np.uint32)
    np.testing.assert_array_equal(key[1:], want[1:])
    assert (key[0] ^ want[0]) & 0x80000000 == 0        # only the top bit of word 0 is generator state
    assert lib.plsa_host_mt19937_jump(key.ctypes.data_as(C.POINTER(C.c_uint32)), 41) == 1


def test_headers_are_plain_c():
    """Both headers compile as C (a cgo / JNI / ctypes-generator style consumer never sees C++): gcc -fsyntax-only, -Wall -Werror."""
    import subprocess
    for h in ("plsa_hip.h", "plsa_hip_diag.h"):
        out = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c",
                              os.path.join(ROOT, "include", h)], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr


def test_arithmetic_keyword_and_environment_map_to_the_flags(monkeypatch):
    """`arithmetic=` / ENSTOP_AMD_ARITHMETIC -> PLSA_REFERENCE_SUMS (| PLSA_REFERENCE_LL); the values are the header's."""
    from enstop_amd import engine, PLSA, StreamedPLSA, BlockParallelPLSA
    hdr = open(os.path.join(ROOT, "include", "plsa_hip.h")).read()
    assert re.search(r"PLSA_REFERENCE_SUMS\s*=\s*256", hdr) and re.search(r"PLSA_REFERENCE_LL\s*=\s*512", hdr)
    assert engine.PLSA_REFERENCE_SUMS == 256 and engine.PLSA_REFERENCE_LL == 512
    assert engine.arithmetic_flags(None) == 0 == engine.arithmetic_flags("default")
    assert engine.arithmetic_flags("reference") == 256 and engine.arithmetic_flags("reference_source") == 768
    assert engine.arithmetic_flags(512) == 512
    with pytest.raises(ValueError):
        engine.arithmetic_flags("fast")
    with pytest.raises(ValueError):
        engine.arithmetic_flags(1)
    monkeypatch.delenv("ENSTOP_AMD_ARITHMETIC", raising=False)
    monkeypatch.delenv("ENSTOP_AMD_MATERIALISE", raising=False)
    assert engine.default_flags() == engine.PLSA_FUSED
    monkeypatch.setenv("ENSTOP_AMD_ARITHMETIC", "reference")
    assert engine.default_flags() == engine.PLSA_FUSED | 256
    monkeypatch.delenv("ENSTOP_AMD_ARITHMETIC")
    assert PLSA(arithmetic="reference")._flags() & 256 and not PLSA()._flags() & 256
    assert StreamedPLSA(arithmetic="reference_source")._flags() & 768 == 768
    assert BlockParallelPLSA(arithmetic="reference").get_params()["arithmetic"] == "reference"

