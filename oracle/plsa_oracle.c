/*
 * plsa_oracle.c -- CPU restatement of the reference's pLSA EM hot path.
 *
 * TEST INFRASTRUCTURE, NOT PRODUCT.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this library; enstop_amd/ never does.
 *
 * Parity status: PINNED.  Every function below is checked in tests/test_oracle_golden.py
 * against the .npz fixtures under tests/golden/ produced by running the reference's own
 * enstop/plsa.py + enstop/enstop_.py + enstop/streamed_plsa.py in the build container
 * (tests/golden/make_golden.py).
 * The E-step, both M-steps and the refit M-step reproduce those fixtures bit-for-bit when the
 * library is built strict (no -ffast-math) and run on one thread; the log-likelihood agrees to
 * float32 rounding (NumPy's float32 log and libm logf differ in the last ulp).
 *
 * Each function cites the reference lines it follows (paths relative to /root/reference).
 * Layouts are the reference's: COO rows/cols int32, vals float32, V = P(w|z) float32 [k,m]
 * C-order, U = P(z|d) float32 [n,k] C-order, P = P(z|w,d) float32 [nnz,k].
 *
 * Two builds of this one file (oracle/Makefile):
 *   liboracle_plsa.so       -O2, strict IEEE, OpenMP      -> the checker
 *   liboracle_plsa_fast.so  -O3 -ffast-math, OpenMP       -> the timed "port" CPU baseline
 *     (fastmath=True + parallel=True of the numba decorators, enstop/plsa.py:35-37)
 * Two further DIAGNOSTIC builds (not the reference's arithmetic; used by tests/test_parity_at_scale.py
 * to show where the float32 reference itself is the inaccurate side at BASELINE sizes):
 *   liboracle_plsa_n64.so   as the checker, but norm_pwz (plsa.py:193, one float32 running sum over
 *                           ALL nnz per topic) and the log-likelihood accumulator (plsa.py:322) in
 *                           float64 (-DORACLE_WIDE_NORMS)
 *   liboracle_plsa_wide.so  additionally the P(w|z) / P(z|d) accumulators and norm_pdz in float64
 *                           (-DORACLE_WIDE_FACTORS): the exact-arithmetic limit of the same algorithm
 * Thread structure mirrors the reference: E-step and log-likelihood are parallel over nnz
 * (numba.prange, plsa.py:91,375), the M-step scatter is a single serial loop (plsa.py:182-194),
 * the M-step normalisation is parallel over topics (plsa.py:196).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#if defined(ORACLE_WIDE_FACTORS)
typedef double acc_t;   /* P(w|z), P(z|d) accumulators and norm_pdz */
typedef double norm_t;  /* norm_pwz and the log-likelihood accumulator */
#define ORACLE_WIDE_ACC 1
#elif defined(ORACLE_WIDE_NORMS)
typedef float acc_t;
typedef double norm_t;
#define ORACLE_WIDE_ACC 0
#else
typedef float acc_t;    /* the reference's types: everything float32 (plsa.py:27-34, 112-119) */
typedef float norm_t;
#define ORACLE_WIDE_ACC 0
#endif

void oracle_set_threads(int t) {
#ifdef _OPENMP
    if (t > 0) omp_set_num_threads(t);
#else
    (void)t;
#endif
}

/* The log-likelihood reduction alone on ONE thread whatever the thread count of the E-step (which is independent per
 * non-zero: threads cannot change it; the M-step scatter is serial in every build): the reference's source read literally --
 * `result` one float32 running sum over the non-zeros in order, plsa.py:322, 375-384 -- without paying a single-threaded
 * E-step for it.  Results are those of oracle_set_threads(1), bit for bit (tests/test_oracle_golden.py). */
static int g_ll_sequential = 0;
void oracle_set_ll_sequential(int on) { g_ll_sequential = on != 0; }

int oracle_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

/* enstop/plsa.py:89-105  plsa_e_step */
void oracle_e_step(const int32_t *rows, const int32_t *cols, int64_t nnz,
                   const float *V, const float *U, float *P,
                   int64_t m, int64_t k, float thresh) {
#pragma omp parallel for schedule(static)
    for (int64_t nz = 0; nz < nnz; nz++) {
        const int64_t d = rows[nz], w = cols[nz];
        float *p = P + nz * k;
        float norm = 0.0f;
        for (int64_t z = 0; z < k; z++) {
            float v = V[z * m + w] * U[d * k + z];
            if (v > thresh) {
                p[z] = v;
                norm += p[z];
            } else {
                p[z] = 0.0f;
            }
        }
        for (int64_t z = 0; z < k; z++)
            if (norm > 0.0f) p[z] /= norm;
    }
}

/* Both M-steps.  sw == NULL: enstop/plsa.py:172-204 (plsa_m_step); sw != NULL: enstop/plsa.py:277-310
 * (plsa_m_step_w_sample_weight: t = s * sample_weight[d] feeds P(w|z) and norm_pwz, P(z|d) and
 * norm_pdz take the unweighted s, plsa.py:292-300).  In the checker build acc_t = norm_t = float and
 * the sums run in place in V / U exactly like the reference; the diagnostic builds widen them. */
static int m_step_impl(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                       float *V, float *U, const float *P, const float *sw,
                       float *norm_pwz_out, float *norm_pdz_out, int64_t n, int64_t m, int64_t k) {
    acc_t *Va, *Ua, *npdz;
    norm_t *npwz = (norm_t *)calloc((size_t)k + 1, sizeof(norm_t));
    if (ORACLE_WIDE_ACC) {
        Va = (acc_t *)calloc((size_t)(k * m) + 1, sizeof(acc_t));
        Ua = (acc_t *)calloc((size_t)(n * k) + 1, sizeof(acc_t));
        npdz = (acc_t *)calloc((size_t)n + 1, sizeof(acc_t));
    } else {
        Va = (acc_t *)(void *)V; Ua = (acc_t *)(void *)U; npdz = (acc_t *)(void *)norm_pdz_out;
        memset(V, 0, sizeof(float) * (size_t)(k * m));             /* plsa.py:176-180 */
        memset(U, 0, sizeof(float) * (size_t)(n * k));
        memset(norm_pdz_out, 0, sizeof(float) * (size_t)n);
    }
    if (!npwz || !Va || !Ua || !npdz) return -1;
    for (int64_t nz = 0; nz < nnz; nz++) {              /* serial: plsa.py:182 is range() */
        const int64_t d = rows[nz], w = cols[nz];
        const float x = vals[nz];
        const float *p = P + nz * k;
        for (int64_t z = 0; z < k; z++) {
            float s = x * p[z];                                     /* plsa.py:188 */
            float t = sw ? s * sw[d] : s;                           /* plsa.py:294 */
            Va[z * m + w] += t;
            Ua[d * k + z] += s;
            npwz[z] += t;
            npdz[d] += s;
        }
    }
    /* plsa.py:196-202 (and 302-308) */
#pragma omp parallel for schedule(static)
    for (int64_t z = 0; z < k; z++) {
        if (npwz[z] > 0)
            for (int64_t w = 0; w < m; w++) V[z * m + w] = (float)(Va[z * m + w] / npwz[z]);
        else if (ORACLE_WIDE_ACC)
            for (int64_t w = 0; w < m; w++) V[z * m + w] = (float)Va[z * m + w];
        for (int64_t d = 0; d < n; d++)
            U[d * k + z] = npdz[d] > 0 ? (float)(Ua[d * k + z] / npdz[d]) : (float)Ua[d * k + z];
    }
    for (int64_t z = 0; z < k; z++) norm_pwz_out[z] = (float)npwz[z];
    if (ORACLE_WIDE_ACC) {
        for (int64_t d = 0; d < n; d++) norm_pdz_out[d] = (float)npdz[d];
        free(Va); free(Ua); free(npdz);
    }
    free(npwz);
    return 0;
}

/* enstop/plsa.py:172-204  plsa_m_step */
void oracle_m_step(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                   float *V, float *U, const float *P, float *norm_pwz, float *norm_pdz,
                   int64_t n, int64_t m, int64_t k) {
    (void)m_step_impl(rows, cols, vals, nnz, V, U, P, NULL, norm_pwz, norm_pdz, n, m, k);
}

/* enstop/plsa.py:277-310  plsa_m_step_w_sample_weight */
void oracle_m_step_w(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                     float *V, float *U, const float *P, const float *sw,
                     float *norm_pwz, float *norm_pdz, int64_t n, int64_t m, int64_t k) {
    (void)m_step_impl(rows, cols, vals, nnz, V, U, P, sw, norm_pwz, norm_pdz, n, m, k);
}

/* enstop/plsa.py:795-816  plsa_refit_m_step (sample_weight is accepted and unused there) */
void oracle_refit_m_step(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                         float *U, const float *P, float *norm_pdz, int64_t n, int64_t k) {
    (void)cols;
    memset(U, 0, sizeof(float) * (size_t)(n * k));
    memset(norm_pdz, 0, sizeof(float) * (size_t)n);
    for (int64_t nz = 0; nz < nnz; nz++) {
        const int64_t d = rows[nz];
        const float x = vals[nz];
        const float *p = P + nz * k;
        for (int64_t z = 0; z < k; z++) {
            float s = x * p[z];
            U[d * k + z] += s;
            norm_pdz[d] += s;
        }
    }
    for (int64_t z = 0; z < k; z++)
        for (int64_t d = 0; d < n; d++)
            if (norm_pdz[d] > 0.0f) U[d * k + z] /= norm_pdz[d];
}

/* enstop/plsa.py:372-386  log_likelihood (float32 accumulator `result`, plsa.py:322) */
float oracle_log_likelihood(const int32_t *rows, const int32_t *cols, const float *vals,
                            int64_t nnz, const float *V, const float *U, const float *sw,
                            int64_t m, int64_t k) {
    norm_t result = 0;                                   /* float32 in the reference, plsa.py:322 */
#pragma omp parallel for schedule(static) reduction(+ : result) if (!g_ll_sequential)
    for (int64_t nz = 0; nz < nnz; nz++) {
        const int64_t d = rows[nz], w = cols[nz];
        const float x = vals[nz];
        float p_w_given_d = 0.0f;
        for (int64_t z = 0; z < k; z++) p_w_given_d += V[z * m + w] * U[d * k + z];
        result += x * logf(p_w_given_d) * sw[d];
    }
    return (float)result;
}

/* enstop/plsa.py:583-640  plsa_fit_inner.
 * ll_trace (nullable) receives every log-likelihood evaluated (the one before the loop first);
 * *iters receives the number of EM iterations executed.  Returns 0, or -1 on allocation failure. */
int oracle_fit_inner(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                     float *V, float *U, const float *sw, int64_t n, int64_t m, int64_t k,
                     int32_t n_iter, int32_t n_iter_per_test, double tolerance, float thresh,
                     int32_t use_sample_weights, float *ll_trace, int32_t *n_ll, int32_t *iters) {
    float *P = (float *)calloc((size_t)(nnz * k) + 1, sizeof(float));          /* plsa.py:586 */
    float *norm_pwz = (float *)calloc((size_t)k + 1, sizeof(float));           /* plsa.py:588 */
    float *norm_pdz = (float *)calloc((size_t)n + 1, sizeof(float));           /* plsa.py:589 */
    if (!P || !norm_pwz || !norm_pdz) { free(P); free(norm_pwz); free(norm_pdz); return -1; }
    int32_t nll = 0, it = 0;
    float prev = oracle_log_likelihood(rows, cols, vals, nnz, V, U, sw, m, k);  /* plsa.py:591 */
    if (ll_trace) ll_trace[nll] = prev;
    nll++;
    for (int32_t i = 0; i < n_iter; i++) {
        oracle_e_step(rows, cols, nnz, V, U, P, m, k, thresh);
        if (use_sample_weights)
            oracle_m_step_w(rows, cols, vals, nnz, V, U, P, sw, norm_pwz, norm_pdz, n, m, k);
        else
            oracle_m_step(rows, cols, vals, nnz, V, U, P, norm_pwz, norm_pdz, n, m, k);
        it++;
        if (i % n_iter_per_test == 0) {                                         /* plsa.py:630 */
            float cur = oracle_log_likelihood(rows, cols, vals, nnz, V, U, sw, m, k);
            if (ll_trace) ll_trace[nll] = cur;
            nll++;
            float change = fabsf(cur - prev);
            /* plsa.py:635; the ratio is float32, the comparison with `tolerance` float64 */
            if (change == 0.0f || (double)(change / fabsf(cur)) < tolerance) break;
            prev = cur;
        }
    }
    if (n_ll) *n_ll = nll;
    if (iters) *iters = it;
    free(P); free(norm_pwz); free(norm_pdz);
    return 0;
}

/* enstop/plsa.py:884-920  plsa_refit_inner.  The stop test only acts when the log-likelihood is
 * positive (plsa.py:913), which never happens for probabilities: all n_iter iterations run. */
int oracle_refit_inner(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                       const float *topics, float *U, const float *sw, int64_t n, int64_t m,
                       int64_t k, int32_t n_iter, int32_t n_iter_per_test, double tolerance,
                       float thresh, float *ll_trace, int32_t *n_ll, int32_t *iters) {
    float *P = (float *)calloc((size_t)(nnz * k) + 1, sizeof(float));
    float *norm_pdz = (float *)calloc((size_t)n + 1, sizeof(float));
    if (!P || !norm_pdz) { free(P); free(norm_pdz); return -1; }
    int32_t nll = 0, it = 0;
    float prev = oracle_log_likelihood(rows, cols, vals, nnz, topics, U, sw, m, k);
    if (ll_trace) ll_trace[nll] = prev;
    nll++;
    for (int32_t i = 0; i < n_iter; i++) {
        oracle_e_step(rows, cols, nnz, topics, U, P, m, k, thresh);
        oracle_refit_m_step(rows, cols, vals, nnz, U, P, norm_pdz, n, k);
        it++;
        if (i % n_iter_per_test == 0) {
            float cur = oracle_log_likelihood(rows, cols, vals, nnz, topics, U, sw, m, k);
            if (ll_trace) ll_trace[nll] = cur;
            nll++;
            if (cur > 0.0f) {
                float change = fabsf(cur - prev);
                if ((double)(change / fabsf(cur)) < tolerance) break;
                prev = cur;
            }
        }
    }
    if (n_ll) *n_ll = nll;
    if (iters) *iters = it;
    free(P); free(norm_pdz);
    return 0;
}

/* ---------------------------------------------------------------------------------------------------
 * enstop/streamed_plsa.py -- the block-streamed loops.  The checker build is float32 everywhere and pinned
 * bit for bit by tests/golden/stream*.npz.  The FIT loop also honours the diagnostic builds (round 5:
 * norm_pwz in norm_t, and under ORACLE_WIDE_FACTORS the P(w|z) / P(z|d) sums and norm_pdz in float64
 * side buffers) so that BASELINE config 5 -- whose nnz x k array cannot exist on any host -- has an
 * exact-arithmetic comparison; the refit loop stays float32.
 * ------------------------------------------------------------------------------------------------- */

/* streamed_plsa.py:102-119  plsa_e_step_on_a_block: P block row (nz - block_start) */
static void streamed_e_block(const int32_t *rows, const int32_t *cols, const float *V, const float *U,
                             float *Pb, int64_t b0, int64_t b1, int64_t m, int64_t k, float thresh) {
#pragma omp parallel for schedule(static)
    for (int64_t nz = b0; nz < b1; nz++) {
        const int64_t d = rows[nz], w = cols[nz];
        float *p = Pb + (nz - b0) * k;
        float norm = 0.0f;
        for (int64_t z = 0; z < k; z++) {
            float v = V[z * m + w] * U[d * k + z];
            if (v > thresh) { p[z] = v; norm += v; } else { p[z] = 0.0f; }
        }
        for (int64_t z = 0; z < k; z++)
            if (norm > 0.0f) p[z] /= norm;
    }
}

/* streamed_plsa.py:204-219 (plsa_partial_m_step_on_a_block), :304-320 (..._w_sample_weight; sw != NULL),
 * (the refit variant :774-785 follows below).  In the checker build acc_t = norm_t = float and
 * nV / nU / norm_pdz ARE the reference's float32 arrays. */
static void streamed_m_block(const int32_t *rows, const int32_t *cols, const float *vals, acc_t *nV, acc_t *nU,
                             const float *Pb, const float *sw, norm_t *norm_pwz, acc_t *norm_pdz,
                             int64_t b0, int64_t b1, int64_t m, int64_t k) {
    for (int64_t nz = b0; nz < b1; nz++) {
        const int64_t d = rows[nz], w = cols[nz];
        const float x = vals[nz];
        const float *p = Pb + (nz - b0) * k;
        for (int64_t z = 0; z < k; z++) {
            float s = x * p[z];
            float t = sw ? s * sw[d] : s;
            if (nV) { nV[z * m + w] += t; norm_pwz[z] += t; }
            nU[d * k + z] += s;
            norm_pdz[d] += s;
        }
    }
}

/* streamed_plsa.py:469-603  plsa_fit_inner_blockwise with :322-391 / :394-465 (plsa_em_step*) inlined:
 * `n_blocks = nnz // block_size + 1` (:342), factors double-buffered (prev zeroed and handed back as the
 * next "next", :386-391), stop test WITHOUT the `change == 0` arm (:596-597).  V, U receive the result. */
int oracle_streamed_fit_inner(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                              float *V, float *U, const float *sw, int64_t n, int64_t m, int64_t k,
                              int64_t block_size, int32_t n_iter, int32_t n_iter_per_test, double tolerance,
                              float thresh, int32_t use_sample_weights, float *ll_trace, int32_t *n_ll,
                              int32_t *iters) {
    float *Pb = (float *)calloc((size_t)(block_size * k) + 1, sizeof(float));      /* :543 */
    norm_t *norm_pwz = (norm_t *)calloc((size_t)k + 1, sizeof(norm_t));
    acc_t *norm_pdz = (acc_t *)calloc((size_t)n + 1, sizeof(acc_t));
    float *bufV = (float *)calloc((size_t)(k * m) + 1, sizeof(float));             /* :552-553 */
    float *bufU = (float *)calloc((size_t)(n * k) + 1, sizeof(float));
    /* diagnostic wide build only: float64 sums beside the float32 "next" factors */
    acc_t *wV = ORACLE_WIDE_ACC ? (acc_t *)calloc((size_t)(k * m) + 1, sizeof(acc_t)) : NULL;
    acc_t *wU = ORACLE_WIDE_ACC ? (acc_t *)calloc((size_t)(n * k) + 1, sizeof(acc_t)) : NULL;
    if (!Pb || !norm_pwz || !norm_pdz || !bufV || !bufU || (ORACLE_WIDE_ACC && (!wV || !wU))) {
        free(Pb); free(norm_pwz); free(norm_pdz); free(bufV); free(bufU); free(wV); free(wU); return -1;
    }
    float *pV = V, *pU = U, *nV = bufV, *nU = bufU;
    int32_t nll = 0, it = 0;
    float prev = oracle_log_likelihood(rows, cols, vals, nnz, pV, pU, sw, m, k);   /* :548 */
    if (ll_trace) ll_trace[nll] = prev;
    nll++;
    const int64_t n_blocks = nnz / block_size + 1;
    for (int32_t i = 0; i < n_iter; i++) {
        memset(norm_pdz, 0, sizeof(acc_t) * (size_t)n);                             /* :345-346 */
        memset(norm_pwz, 0, sizeof(norm_t) * (size_t)k);
        acc_t *aV = ORACLE_WIDE_ACC ? wV : (acc_t *)(void *)nV;
        acc_t *aU = ORACLE_WIDE_ACC ? wU : (acc_t *)(void *)nU;
        if (ORACLE_WIDE_ACC) {
            memset(wV, 0, sizeof(acc_t) * (size_t)(k * m));
            memset(wU, 0, sizeof(acc_t) * (size_t)(n * k));
        }
        for (int64_t b = 0; b < n_blocks; b++) {
            const int64_t b0 = b * block_size;
            const int64_t b1 = nnz < b0 + block_size ? nnz : b0 + block_size;
            streamed_e_block(rows, cols, pV, pU, Pb, b0, b1, m, k, thresh);
            streamed_m_block(rows, cols, vals, aV, aU, Pb, use_sample_weights ? sw : NULL, norm_pwz, norm_pdz,
                             b0, b1, m, k);
        }
        for (int64_t z = 0; z < k; z++) {                                           /* :378-384 */
            if (ORACLE_WIDE_ACC) {
                for (int64_t w = 0; w < m; w++)
                    nV[z * m + w] = (float)(norm_pwz[z] > 0 ? aV[z * m + w] / norm_pwz[z] : aV[z * m + w]);
                for (int64_t d = 0; d < n; d++)
                    nU[d * k + z] = (float)(norm_pdz[d] > 0 ? aU[d * k + z] / norm_pdz[d] : aU[d * k + z]);
            } else {
                if (norm_pwz[z] > 0)
                    for (int64_t w = 0; w < m; w++) nV[z * m + w] = (float)(nV[z * m + w] / norm_pwz[z]);
                for (int64_t d = 0; d < n; d++)
                    if (norm_pdz[d] > 0) nU[d * k + z] = (float)(nU[d * k + z] / norm_pdz[d]);
            }
        }
        memset(pV, 0, sizeof(float) * (size_t)(k * m));                             /* :388-389 */
        memset(pU, 0, sizeof(float) * (size_t)(n * k));
        { float *t = pV; pV = nV; nV = t; t = pU; pU = nU; nU = t; }               /* :391, :558-590 */
        it++;
        if (i % n_iter_per_test == 0) {
            float cur = oracle_log_likelihood(rows, cols, vals, nnz, pV, pU, sw, m, k);
            if (ll_trace) ll_trace[nll] = cur;
            nll++;
            float change = fabsf(cur - prev);
            if ((double)(change / fabsf(cur)) < tolerance) break;                   /* :596-597 */
            prev = cur;
        }
    }
    if (pV != V) memcpy(V, pV, sizeof(float) * (size_t)(k * m));
    if (pU != U) memcpy(U, pU, sizeof(float) * (size_t)(n * k));
    if (n_ll) *n_ll = nll;
    if (iters) *iters = it;
    free(Pb); free(norm_pwz); free(norm_pdz); free(bufV); free(bufU); free(wV); free(wU);
    return 0;
}

/* streamed_plsa.py:774-785  plsa_partial_refit_m_step_on_a_block: float32 in every build */
static void streamed_refit_m_block(const int32_t *rows, const float *vals, float *nU, const float *Pb,
                                   float *norm_pdz, int64_t b0, int64_t b1, int64_t k) {
    for (int64_t nz = b0; nz < b1; nz++) {
        const int64_t d = rows[nz];
        const float x = vals[nz];
        const float *p = Pb + (nz - b0) * k;
        for (int64_t z = 0; z < k; z++) {
            float s = x * p[z];
            nU[d * k + z] += s;
            norm_pdz[d] += s;
        }
    }
}

/* streamed_plsa.py:851-956  plsa_refit_inner_blockwise with :788-847 (plsa_refit_em_step) inlined.  The caller's
 * e_step_thresh is NOT forwarded (:932-943): the E-step runs with plsa_refit_em_step's default 1e-32 (:798);
 * the stop test only acts on a positive log-likelihood (:949) -- all n_iter iterations run. */
int oracle_streamed_refit_inner(const int32_t *rows, const int32_t *cols, const float *vals, int64_t nnz,
                                const float *topics, float *U, const float *sw, int64_t n, int64_t m, int64_t k,
                                int64_t block_size, int32_t n_iter, int32_t n_iter_per_test, double tolerance,
                                float thresh_ignored, float *ll_trace, int32_t *n_ll, int32_t *iters) {
    (void)thresh_ignored;
    float *Pb = (float *)calloc((size_t)(block_size * k) + 1, sizeof(float));
    float *norm_pdz = (float *)calloc((size_t)n + 1, sizeof(float));
    float *bufU = (float *)calloc((size_t)(n * k) + 1, sizeof(float));
    if (!Pb || !norm_pdz || !bufU) { free(Pb); free(norm_pdz); free(bufU); return -1; }
    float *pU = U, *nU = bufU;
    int32_t nll = 0, it = 0;
    float prev = oracle_log_likelihood(rows, cols, vals, nnz, topics, pU, sw, m, k);
    if (ll_trace) ll_trace[nll] = prev;
    nll++;
    const int64_t n_blocks = nnz / block_size + 1;
    for (int32_t i = 0; i < n_iter; i++) {
        memset(norm_pdz, 0, sizeof(float) * (size_t)n);
        for (int64_t b = 0; b < n_blocks; b++) {
            const int64_t b0 = b * block_size;
            const int64_t b1 = nnz < b0 + block_size ? nnz : b0 + block_size;
            streamed_e_block(rows, cols, topics, pU, Pb, b0, b1, m, k, 1e-32f);
            streamed_refit_m_block(rows, vals, nU, Pb, norm_pdz, b0, b1, k);
        }
        for (int64_t z = 0; z < k; z++)
            for (int64_t d = 0; d < n; d++)
                if (norm_pdz[d] > 0.0f) nU[d * k + z] /= norm_pdz[d];
        memset(pU, 0, sizeof(float) * (size_t)(n * k));
        { float *t = pU; pU = nU; nU = t; }
        it++;
        if (i % n_iter_per_test == 0) {
            float cur = oracle_log_likelihood(rows, cols, vals, nnz, topics, pU, sw, m, k);
            if (ll_trace) ll_trace[nll] = cur;
            nll++;
            if (cur > 0.0f) {
                float change = fabsf(cur - prev);
                if ((double)(change / fabsf(cur)) < tolerance) break;
                prev = cur;
            }
        }
    }
    if (pU != U) memcpy(U, pU, sizeof(float) * (size_t)(n * k));
    if (n_ll) *n_ll = nll;
    if (iters) *iters = it;
    free(Pb); free(norm_pdz); free(bufU);
    return 0;
}

/* enstop/utils.py:22-41  normalize(ndarray, axis=1): float64, in place, sequential marginal */
void oracle_normalize_rows(double *a, int64_t rows, int64_t cols) {
    for (int64_t i = 0; i < rows; i++) {
        double marginal = 0.0;
        for (int64_t j = 0; j < cols; j++) marginal += a[i * cols + j];
        for (int64_t j = 0; j < cols; j++)
            if (marginal > 0.0) a[i * cols + j] /= marginal;
    }
}
