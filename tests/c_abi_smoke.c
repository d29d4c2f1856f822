/* Plain-C consumer of the drop-in boundary (include/plsa_hip.h): no Python, no C++ types.  Built by
 * tests/test_hip_parity.py::test_c_abi_from_plain_c with gcc and run on the GPU box.
 * Round 5: (1) a VALUE check across the boundary -- the reference's own fit of tests/golden/fit_k8_tol0.npz
 * (enstop/plsa.py plsa_fit; inputs and outputs embedded by tests/golden/make_c_fixture.py) reproduced through the C ABI
 * within the north-star tolerances (factors 1e-4 of the largest entry, log-likelihood 1e-5 relative), both
 * schedules; (2) the header's input contract answered with a status code, not a GPU fault. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "plsa_hip.h"
#include "plsa_hip_diag.h"   /* plsa_comm_barrier / plsa_comm_allreduce_f64 only: the one-rank communicator check below */
#include "golden/fit_k8_tol0_fixture.h"

static unsigned long long s = 88172645463325252ull;
static double rnd(void) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0; }

#define CHECK(call) do { if (call) { fprintf(stderr, "%s failed: %s\n", #call, plsa_last_error(ctx)); return 2; } } while (0)

static double peak_rel(const float *a, const float *b, size_t cnt) {
    double worst = 0, peak = 0;
    for (size_t i = 0; i < cnt; i++) { if (fabs(a[i] - b[i]) > worst) worst = fabs(a[i] - b[i]); if (fabs(b[i]) > peak) peak = fabs(b[i]); }
    return worst / peak;
}

/* the reference's outputs for the embedded fixture, through the C ABI alone */
static int value_check(plsa_ctx *ctx) {
    static float U[FX_N * FX_K], V[FX_K * FX_M], trace[FX_N_ITER + 2];
    for (int sched = 0; sched < 2; sched++) {
        int32_t iters = 0, n_ll = 0;
        CHECK(plsa_upload_csr(ctx, fx_indptr, fx_indices, fx_data, FX_N, FX_M, FX_NNZ));
        CHECK(plsa_set_factors(ctx, fx_U0, fx_V0, FX_N, FX_M, FX_K));
        CHECK(plsa_fit(ctx, NULL, FX_N_ITER, FX_N_ITER_PER_TEST, 0.0, FX_THRESH, (sched ? 0 : PLSA_FUSED) | PLSA_TRACE_LL,
                       &iters, trace, &n_ll));
        CHECK(plsa_get_factors(ctx, U, V));
        const double eu = peak_rel(U, fx_U, FX_N * FX_K), ev = peak_rel(V, fx_V, FX_K * FX_M);
        double el = 0;
        for (int i = 0; i < FX_N_LL && i < n_ll; i++) {
            const double e = fabs(trace[i] - fx_ll_trace[i]) / fabs(fx_ll_trace[i]);
            if (e > el) el = e;
        }
        if (iters != FX_ITERS || n_ll != FX_N_LL || !(eu <= 1e-4) || !(ev <= 1e-4) || !(el <= 1e-5)) {
            fprintf(stderr, "value check (%s): iters=%d (want %d) n_ll=%d (want %d) U %.3g V %.3g LL %.3g\n",
                    sched ? "materialised" : "fused", iters, FX_ITERS, n_ll, FX_N_LL, eu, ev, el);
            return 5;
        }
        printf("c-abi values (%s) vs the reference: U %.2e  V %.2e  LL %.2e\n", sched ? "materialised" : "fused", eu, ev, el);
    }
    return 0;
}

/* "indices must be in [0, m), indptr non-decreasing" (plsa_hip.h): a violation is a status + message, the context
 * stays usable and holds no corpus */
static int contract_check(plsa_ctx *ctx) {
    static int32_t ip[FX_N + 1], ix[FX_NNZ];
    int64_t n = 0;
    memcpy(ip, fx_indptr, sizeof ip); memcpy(ix, fx_indices, sizeof ix);
    ix[FX_NNZ / 2] = FX_M;                                   /* one column index == m */
    if (!plsa_upload_csr(ctx, ip, ix, fx_data, FX_N, FX_M, FX_NNZ) || !strstr(plsa_last_error(ctx), "column index")) return 6;
    ix[FX_NNZ / 2] = -1;
    if (!plsa_upload_csr(ctx, ip, ix, fx_data, FX_N, FX_M, FX_NNZ)) return 6;
    memcpy(ix, fx_indices, sizeof ix);
    { const int32_t t = ip[10]; ip[10] = ip[11]; ip[11] = t; }   /* indptr decreasing somewhere */
    if (ip[10] != ip[11] && (!plsa_upload_csr(ctx, ip, ix, fx_data, FX_N, FX_M, FX_NNZ) || !strstr(plsa_last_error(ctx), "indptr"))) return 7;
    CHECK(plsa_active_shape(ctx, &n, NULL, NULL));
    if (n != 0 || !plsa_set_factors(ctx, fx_U0, fx_V0, FX_N, FX_M, FX_K)) return 8;   /* nothing resident after a refusal */
    return 0;
}

int main(void) {
    const int64_t n = 300, m = 200;
    const int32_t k = 12;
    int32_t *indptr = malloc(sizeof(int32_t) * (n + 1));
    int32_t *indices = malloc(sizeof(int32_t) * n * 40);
    float *data = malloc(sizeof(float) * n * 40);
    int64_t nnz = 0;
    indptr[0] = 0;
    for (int64_t d = 0; d < n; d++) {
        int len = 5 + (int)(rnd() * 30), w = 0;
        for (int j = 0; j < len && w < m; j++) {
            w += 1 + (int)(rnd() * 5);
            if (w >= m) break;
            indices[nnz] = w; data[nnz] = (float)(1 + (int)(rnd() * 4)); nnz++;
        }
        indptr[d + 1] = (int32_t)nnz;
    }
    float *U = malloc(sizeof(float) * n * k), *V = malloc(sizeof(float) * k * m);
    for (int64_t d = 0; d < n; d++) { double t = 0; for (int z = 0; z < k; z++) { U[d * k + z] = (float)rnd(); t += U[d * k + z]; }
                                      for (int z = 0; z < k; z++) U[d * k + z] /= (float)t; }
    for (int z = 0; z < k; z++) { double t = 0; for (int64_t w = 0; w < m; w++) { V[z * m + w] = (float)rnd(); t += V[z * m + w]; }
                                  for (int64_t w = 0; w < m; w++) V[z * m + w] /= (float)t; }
    plsa_ctx *ctx = NULL;
    if (plsa_create(0, &ctx)) { fprintf(stderr, "plsa_create: %s\n", plsa_last_error(NULL)); return 1; }
    { int rc = value_check(ctx); if (rc) return rc; }
    { int rc = contract_check(ctx); if (rc) { fprintf(stderr, "contract check failed (%d): %s\n", rc, plsa_last_error(ctx)); return rc; } }
    CHECK(plsa_upload_csr(ctx, indptr, indices, data, n, m, nnz));
    CHECK(plsa_set_factors(ctx, U, V, n, m, k));
    double ll0 = 0, ll1 = 0;
    CHECK(plsa_log_likelihood(ctx, NULL, &ll0));
    int32_t iters = 0, n_ll = 0;
    float trace[64];
    CHECK(plsa_fit(ctx, NULL, 20, 5, 0.0, 1e-32f, PLSA_FUSED | PLSA_TRACE_LL, &iters, trace, &n_ll));
    CHECK(plsa_log_likelihood(ctx, NULL, &ll1));
    CHECK(plsa_get_factors(ctx, U, V));
    double worst = 0;
    for (int64_t d = 0; d < n; d++) { double t = 0; for (int z = 0; z < k; z++) t += U[d * k + z]; if (indptr[d + 1] > indptr[d] && fabs(t - 1) > worst) worst = fabs(t - 1); }
    for (int z = 0; z < k; z++) { double t = 0; for (int64_t w = 0; w < m; w++) t += V[z * m + w]; if (fabs(t - 1) > worst) worst = fabs(t - 1); }
    /* the multi-GPU exchange from plain C: a one-rank RCCL communicator (rank 0 makes the id, every rank
     * joins), all-gather of the topics, the doc-sharded loop with its in-stream all-reduce */
    unsigned char id[PLSA_COMM_ID_BYTES];
    int32_t rank = -1, world = -1;
    CHECK(plsa_comm_unique_id(id));
    CHECK(plsa_comm_init(ctx, id, 0, 1));
    CHECK(plsa_comm_info(ctx, &rank, &world));
    /* two members of an ensemble in the device stack, one grouped all-gather, one copy to pinned host memory */
    void *base = NULL;
    float *pinned = NULL;
    CHECK(plsa_stack_reserve(ctx, 2, m, k, &base));
    CHECK(plsa_copy_components_to_device(ctx, base));
    CHECK(plsa_copy_components_to_device(ctx, (float *)base + (size_t)k * m));
    CHECK(plsa_comm_allgather_stack(ctx, 2, m, k, &pinned));
    double gather_err = 0;
    for (int64_t i = 0; i < (int64_t)k * m; i++) { if (fabs(pinned[i] - V[i]) > gather_err) gather_err = fabs(pinned[i] - V[i]);
                                                   if (fabs(pinned[(int64_t)k * m + i] - V[i]) > gather_err) gather_err = fabs(pinned[(int64_t)k * m + i] - V[i]); }
    int32_t iters2 = 0;
    CHECK(plsa_fit(ctx, NULL, 3, 5, 0.0, 1e-32f, PLSA_FUSED | PLSA_SHARDED, &iters2, NULL, NULL));
    double red[2] = {1.5, -2.0};
    CHECK(plsa_comm_allreduce_f64(ctx, red, 2, 0));
    CHECK(plsa_comm_barrier(ctx));
    CHECK(plsa_comm_destroy(ctx));
    if (rank != 0 || world != 1 || gather_err != 0.0 || iters2 != 3 || red[0] != 1.5) {
        fprintf(stderr, "comm: rank=%d world=%d gather_err=%g iters2=%d\n", rank, world, gather_err, iters2);
        return 4;
    }
    plsa_destroy(ctx);
    if (iters != 20 || n_ll != 5 || !(ll1 > ll0) || worst > 1e-4 || fabs(trace[0] - (float)ll0) > 1e-3 * fabs(ll0)) {
        fprintf(stderr, "unexpected: iters=%d n_ll=%d ll0=%g ll1=%g worst=%g\n", iters, n_ll, ll0, ll1, worst);
        return 3;
    }
    printf("c-abi ok: %d iterations, LL %.4f -> %.4f, row-sum error %.2e\n", iters, ll0, ll1, worst);
    return 0;
}
