/*
 * plsa_hip.h -- C ABI of the MI355X-native pLSA EM engine (libplsa_hip.so).
 *
 * This is the drop-in boundary for the reference's pLSA hot path.  The reference (lmcinnes/enstop,
 * pure Python + numba) has no FFI of its own; its operator boundary is the set of Python functions
 * below, and its own accelerator plug point is the `plsa_fit` swap at enstop/enstop_.py:52-53,92-114.
 * Each entry point names the reference interface it replaces (paths relative to the reference root).
 * enstop_amd/_lib.py is the ctypes binding; INTEGRATION.md shows the stub a maintainer would add.
 * Diagnostics, measurement hooks, test plumbing and the synthetic-corpus generators of the same library -- nothing a
 * maintainer binding the seam needs -- are declared in include/plsa_hip_diag.h.
 *
 * Conventions
 *   - Every function returns 0 on success and a non-zero status on failure; the message is available
 *     from plsa_last_error().  No exception crosses this boundary.
 *   - Host arrays are borrowed for the duration of the call only, C-contiguous, in the reference's
 *     layouts: U = P(z|d) float32 [n,k]; V = P(w|z) float32 [k,m]; P = P(z|w,d) float32 [nnz,k];
 *     CSR int32 indptr[n+1] / int32 indices[nnz] / float32 data[nnz].
 *   - Device buffers are owned by the opaque context.  Internally V is held word-major ([m,kp],
 *     kp = k rounded up to 4) so that one non-zero touches one contiguous k-vector of each factor.
 *   - One context = one device + one HIP stream.  A context is not thread-safe; distinct contexts
 *     are independent (one host thread or process per GPU, like the reference's nogil thread pool,
 *     enstop/enstop_.py:209-217).
 */
#ifndef PLSA_HIP_H
#define PLSA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct plsa_ctx plsa_ctx;

/* plsa_fit()/plsa_refit() `flags` */
enum {
    PLSA_FUSED     = 1, /* never materialise P(z|w,d) (idea of enstop/streamed_plsa.py:341-375)     */
    PLSA_TRACE_LL  = 4, /* also evaluate the log-likelihood test of the last iteration when it
                           cannot change the result (only to fill ll_trace like plsa.py:631)        */
    PLSA_SW_LL_ONLY = 8, /* plsa_fit: `sw` enters the log-likelihood only, the M-step is the unweighted
                           one: plsa_fit_inner(use_sample_weights=False) with non-unit weights,
                           enstop/plsa.py:591, 606-628, 631                                         */
    PLSA_STOP_NO_ZERO_ARM = 16, /* plsa_fit: stop test of enstop/block_parallel_plsa.py:329-331
                           (`change / |cur| < tolerance` only, no `change == 0` arm)               */
    PLSA_GRAPH     = 128, /* plsa_fit (fused): replay the iterations between two likelihood tests from a hipGraph
                           of two iterations captured from the same launch sequence; results identical           */
    PLSA_REFERENCE_SUMS = 256, /* plsa_fit / plsa_refit: THE REFERENCE'S ROUNDING.  Every sum of the E- and M-step is one float32
                           accumulator added in the order the reference's loops add: the E-step norm over z = 0 .. k-1
                           (enstop/plsa.py:96-105, true division), s = x * P(z|w,d) rounded before it is added (:188), P(w|z) columns
                           and P(z|d) rows entry by entry in COO order (:190-191), norm_pdz entry-major / topic-minor (:194) and
                           norm_pwz[z] as ONE chain over all non-zeros (:193) -- the sum that puts the reference 1e-2 from exact
                           arithmetic at 3 M non-zeros.  Results are the bits of the reference's source executed statement by
                           statement (the fixtures under tests/golden; the numba-compiled reference sits 2e-5 from them at config 1).  The
                           kernel sequence is the reference's (P(z|w,d) materialised; PLSA_FUSED is ignored); a parity mode, 20-110 ms
                           per iteration at the BASELINE sizes, not available with PLSA_SHARDED.  The log-likelihood stays the
                           float64-accumulated one (which is what the compiled reference's vectorised reduction delivers to 1e-7) unless: */
    PLSA_REFERENCE_LL = 512, /* the log-likelihood as ONE float32 running sum over the non-zeros in COO order, its inner product a float32
                           sum over the topics in order (enstop/plsa.py:322, 378-384 read literally: one thread, no SIMD).  3.4e-3 from
                           the exact value at config 1 -- enough to move a default-tolerance stop from iteration 61 to 71      */
    PLSA_SHARDED   = 64 /* plsa_fit: the context's rows are ONE SHARD of the corpus; the P(w|z)
                           accumulator and the log-likelihood are all-reduced over the context's RCCL
                           communicator every iteration (plsa_comm_init); every rank passes the same
                           arguments and ends with its own P(z|d) rows and the full P(w|z)            */
};

/* ---- lifetime / errors -----------------------------------------------------------------------
 * Several contexts may live on one device (the reference's thread pool of ensemble members, enstop_.py:209-217: one
 * context per thread).  The HIP runtime multiplexes every stream of a process onto GPU_MAX_HW_QUEUES (default 4) hardware
 * queues and members that share a queue run in submission order (20NG shape, four contexts: +10 % ensemble throughput with
 * 8).  The library does NOT touch the environment (rounds 3-4 did, from a constructor): a host that fits members
 * concurrently sets GPU_MAX_HW_QUEUES=8 itself before its first HIP call -- enstop_amd/_lib.py does so unless
 * ENSTOP_AMD_HW_QUEUES=0 or the variable is already set; plsa_hw_queues() returns the value this process runs with. */
int plsa_device_count(int *count);
int plsa_create(int device, plsa_ctx **out);
void plsa_destroy(plsa_ctx *ctx);
/* ctx may be NULL: returns the last error of a failed plsa_create()/plsa_device_count(). */
const char *plsa_last_error(const plsa_ctx *ctx);
int plsa_synchronize(plsa_ctx *ctx);

/* ---- corpus ------------------------------------------------------------------------------------
 * plsa_upload_csr: the doc-term matrix X, replaces `A = X.tocoo().astype(np.float32)`
 *   (enstop/plsa.py:714, 975).  The uploaded matrix becomes both the *base* corpus and the *active*
 *   matrix the EM kernels iterate over.  indices must be in [0, m), indptr non-decreasing from 0 to nnz: checked on
 *   the device after the copy (one streaming pass); a violation returns a non-zero status and leaves NO corpus
 *   resident (later calls answer "no corpus uploaded").
 * plsa_bootstrap: active := base[idx, :] gathered on the device; replaces
 *   `B = A[bootstrap_sample_indices]` (enstop/enstop_.py:87-88).  idx == NULL restores active := base.
 * plsa_active_shape: dimensions of the active matrix (plsa_download_active_csr: plsa_hip_diag.h).            */
int plsa_upload_csr(plsa_ctx *ctx, const int32_t *indptr, const int32_t *indices, const float *data,
                    int64_t n, int64_t m, int64_t nnz);
int plsa_bootstrap(plsa_ctx *ctx, const int64_t *idx, int64_t n_out);
int plsa_active_shape(plsa_ctx *ctx, int64_t *n, int64_t *m, int64_t *nnz);

/* ---- factors -----------------------------------------------------------------------------------
 * plsa_set_factors: current estimates (p_z_given_d [n,k], p_w_given_z [k,m]) as produced by
 *   plsa_init + the float32 casts (enstop/plsa.py:708-710) or a warm start.  n must equal the active
 *   matrix' row count.  V may be NULL to keep the current topics (refit of new documents).
 * plsa_get_factors: copy back in the reference's layouts; either pointer may be NULL.
 * plsa_copy_components_to_device: D2D copy of P(w|z) as [k,m] float32 into caller-owned device
 *   memory (e.g. a buffer handed to an RCCL all-gather: the np.vstack of enstop/enstop_.py:231).  */
int plsa_set_factors(plsa_ctx *ctx, const float *U, const float *V, int64_t n, int64_t m, int32_t k);
int plsa_get_factors(plsa_ctx *ctx, float *U, float *V);
/* plsa_init(X, k, init="random", rng) + the float32 casts (enstop/plsa.py:455-456, 510-511, 709-710)
 * evaluated on the device WITH the reference's random stream: state_io holds the 624 MT19937 key
 * words followed by the position, i.e. numpy.random.RandomState.get_state()[1:3]; k*m + n*k doubles
 * are drawn exactly as rng.rand(k, m) then rng.rand(n, k) would, and the advanced state is written
 * back (set_state it to leave the host generator where the reference would).  Bit-identical to
 * the host path, ~7x faster at config 3.                                                        */
int plsa_init_factors_mt19937(plsa_ctx *ctx, int32_t k, uint32_t *state_io);
/* The initialisation of plsa_refit (enstop/plsa.py:979-981): p_z_given_d = rng.rand(n, k), L1 row
 * normalised in float64, cast to float32 -- drawn on the device from the same stream -- against the
 * given fixed topics V [k, m] (host, float32).                                                      */
int plsa_refit_init_mt19937(plsa_ctx *ctx, const float *V, int64_t m, int32_t k, uint32_t *state_io /*[625]*/);
int plsa_copy_components_to_device(plsa_ctx *ctx, void *dst_device_km);

/* ---- kernel-level operators ---------------------------------------------------------------------
 * plsa_e_step           <- plsa_e_step                  enstop/plsa.py:39-107 (signature :26)
 *   materialises P(z|w,d) [nnz,k] in HBM; P_out (nullable) receives a host copy.
 * plsa_m_step           <- plsa_m_step                  enstop/plsa.py:124-204 (sw == NULL)
 *                       <- plsa_m_step_w_sample_weight  enstop/plsa.py:221-310 (sw != NULL)
 *                       <- plsa_refit_m_step            enstop/plsa.py:746-816 (update_v == 0)
 *   consumes the device-resident P; overwrites the factors; norms (nullable) receive norm_pwz[k]
 *   and norm_pdz[n].  P(w|z) is updated by column ownership over a CSC copy of X (no atomics,
 *   bit-reproducible); a float-atomic scatter was measured 17x slower on MI355X.
 * plsa_log_likelihood   <- log_likelihood               enstop/plsa.py:329-386
 *   sw == NULL means all-ones.  Accumulated in float64 on the device; the reference returns float32
 *   (callers cast).
 * plsa_set_arithmetic   arithmetic of the four operators above and of every later driver call on this context:
 *   0 (default) the engine's own summation orders (float64 norm_pwz / likelihood), or a combination of PLSA_REFERENCE_SUMS
 *   and PLSA_REFERENCE_LL (see the flags): plsa_e_step / plsa_m_step then return the reference's bits -- P(z|w,d), both
 *   factors, norm_pwz and norm_pdz -- and plsa_log_likelihood the float32 running sum.                          */
int plsa_set_arithmetic(plsa_ctx *ctx, int32_t mode);
int plsa_e_step(plsa_ctx *ctx, float thresh, float *P_out);
int plsa_m_step(plsa_ctx *ctx, const float *sw, int32_t update_v, float *norm_pwz, float *norm_pdz);
int plsa_log_likelihood(plsa_ctx *ctx, const float *sw, double *ll);

/* ---- EM drivers ---------------------------------------------------------------------------------
 * plsa_fit   <- plsa_fit_inner   enstop/plsa.py:517-640   (and the cuda seam enstop/cuda_plsa.py:157)
 * plsa_refit <- plsa_refit_inner enstop/plsa.py:820-920   (topics frozen, stop test as plsa.py:913)
 *   sw: sample weights [n] or NULL (== all ones; the weighted M-step is used iff sw != NULL, the
 *       caller applies `np.any(sample_weight != 1.0)`, enstop/plsa.py:712).
 *   tolerance is float64, thresh float32, exactly as the reference's argument types.
 *   iters_done  <- number of EM iterations the reference would have executed.
 *   ll_trace    (nullable, capacity >= n_iter + 2) every log-likelihood evaluated, float32;
 *   n_ll        (nullable) how many were written.                                                 */
int plsa_fit(plsa_ctx *ctx, const float *sw, int32_t n_iter, int32_t n_iter_per_test,
             double tolerance, float thresh, int32_t flags, int32_t *iters_done, float *ll_trace,
             int32_t *n_ll);
int plsa_refit(plsa_ctx *ctx, const float *sw, int32_t n_iter, int32_t n_iter_per_test,
               double tolerance, float thresh, int32_t flags, int32_t *iters_done, float *ll_trace,
               int32_t *n_ll);

/* ---- doc-sharded single fit ----------------------------------------------------------------------
 * Replacement for the tile reduction of enstop/distributed_plsa.py:99-131 (dask.delayed tiles summed
 * with da.dstack(...).sum) and enstop/block_parallel_plsa.py:182-185: every rank (GPU) holds a row
 * range of X, its rows of P(z|d) and the full P(w|z).  One EM iteration =
 *   plsa_em_accumulate   fused E+M over the local rows: local P(z|d) rows are final, the
 *                        un-normalised P(w|z) sums of the local rows are left in the accumulator;
 *                        ll_partial (nullable) = log-likelihood of the CURRENT factors, local rows
 *   <all-reduce(sum) of the accumulator across ranks -- the caller's RCCL call on the pointer from
 *    plsa_accumulator_device, or plsa_accumulator_get/set through the host>
 *   plsa_em_finish       norm_pwz, division, buffer swap (identical on every rank)
 * Skipping plsa_em_finish discards the iteration (late stop decision, see plsa_fit).
 * Stream contract: `sw` has been copied when plsa_em_accumulate returns (the host buffer may be freed); the kernels
 * of plsa_em_accumulate / plsa_em_finish are only ENQUEUED on the context's stream -- plsa_allreduce_accumulator and
 * plsa_accumulator_get/set are ordered behind them on that stream; a caller that touches plsa_accumulator_device's
 * buffer from a stream of its own calls plsa_synchronize first (ll_partial != NULL implies that synchronisation). */
int plsa_em_accumulate(plsa_ctx *ctx, const float *sw, float thresh, double *ll_partial);

/* Doc-block tiling of the MATERIALISED schedule (enstop/block_parallel_plsa.py:373-403: X cut into row blocks, the
 * responsibilities of a block computed and consumed block by block, the blocks' partial P(w|z) summed, :182-185).  One
 * context per doc block; per EM iteration every block calls
 *   plsa_em_accumulate_materialised   the reference's kernel sequence over the block's rows -- plsa_e_step into P(z|w,d),
 *                                     the M-step from it (plsa.py:39-107, 124-204) -- leaving the block's rows of P(z|d) final
 *                                     and its un-normalised P(w|z) sums in the accumulator, exactly like plsa_em_accumulate;
 *                                     P(z|w,d) is dead when the call returns (the call waits for its own kernels: the
 *                                     buffer may be lent to the next block at once)
 * followed by the same sum over blocks and plsa_em_finish as the doc-sharded fit.  The blocks of ONE device may share one
 * P(z|w,d) buffer sized for the largest block: plsa_p_reserve on one context (own allocation of at least `bytes`, address
 * in *device_ptr), plsa_p_borrow on the others (the context then never allocates or frees P itself; NULL ends the loan; the
 * lender outlives the loan; sharers do not run materialising calls concurrently).  Config 5 (500 M non-zeros, k = 128:
 * 256 GB of P(z|w,d) untiled) runs this schedule in 8 blocks of 32 GB. */
int plsa_em_accumulate_materialised(plsa_ctx *ctx, const float *sw, float thresh, double *ll_partial);
int plsa_p_reserve(plsa_ctx *ctx, int64_t bytes, void **device_ptr);
int plsa_p_borrow(plsa_ctx *ctx, void *device_ptr, int64_t bytes);
int plsa_em_finish(plsa_ctx *ctx);
/* Makes the per-document weights (`sample_weight`, enstop/plsa.py:208, :314) resident in HBM: one copy + one host
 * wait here, after which every call that passes sw = NULL (plsa_em_accumulate in a per-iteration loop, plsa_fit,
 * plsa_log_likelihood ...) uses them without touching the host.  sw = NULL clears them; uploading another matrix
 * with a different number of documents makes the next use an error.                                          */
int plsa_set_sample_weight(plsa_ctx *ctx, const float *sw /* [n] or NULL */);
int plsa_accumulator_device(plsa_ctx *ctx, void **ptr, int64_t *n_floats);

/* ---- multi-GPU exchange: RCCL over xGMI, one process per GPU -------------------------------------------
 * Replaces the two exchange steps of the reference: np.vstack of the members' topics
 * (enstop/enstop_.py:231) and the per-iteration sum of partial factors over tiles / workers
 * (enstop/distributed_plsa.py:116-131, enstop/block_parallel_plsa.py:182-185).
 *   plsa_comm_unique_id   rank 0 creates the 128-byte RCCL id (ncclGetUniqueId); the caller ships it to
 *                         the other ranks (file, environment, any side channel: enstop_amd/comm.py)
 *   plsa_comm_init        ncclCommInitRank on the context's device; collective over all ranks
 *   plsa_stack_reserve    device block [slots][k][m] for the topic matrices of the members THIS process fits
 *                         (the per-thread results that enstop_.py:209-231 collects); a member's P(w|z) is stored
 *                         into a slot with plsa_copy_components_to_device(member_ctx, base + slot * k * m)
 *   plsa_comm_allgather_stack   the np.vstack itself: slot s of every rank -> [s][rank] in ONE grouped
 *                         ncclAllGather launch on the context's stream, then one copy into a page-locked host
 *                         buffer owned by the context.  Run r of the ensemble is fitted by rank r % world in
 *                         slot r / world, so *host is the stack in run order, [slots * world][k][m] (valid until
 *                         the next plsa_comm_allgather_stack on ctx or plsa_destroy; plsa_release_scratch keeps
 *                         it).  Without a communicator: the local stack.
 *   plsa_allreduce_accumulator       in-place ncclAllReduce(sum) of the un-normalised P(w|z) accumulator,
 *                         stream-ordered between plsa_em_accumulate and plsa_em_finish; plsa_fit with
 *                         PLSA_SHARDED issues the same call itself -- every collective of a communicator goes on
 *                         the context's one stream, in the same program order on every rank
 *   plsa_comm_allgather_host / _broadcast_host
 *                         small host payloads staged through HBM (seeds, P(z|d) row blocks of a doc-sharded fit);
 *                         barrier and float64 all-reduce (timings): plsa_hip_diag.h
 * Without a communicator (world = 1) every call degenerates to the identity.                          */
#define PLSA_COMM_ID_BYTES 128
int plsa_comm_unique_id(void *id128);
int plsa_comm_init(plsa_ctx *ctx, const void *id128, int32_t rank, int32_t world);
int plsa_comm_destroy(plsa_ctx *ctx);
int plsa_comm_info(plsa_ctx *ctx, int32_t *rank, int32_t *world);
int plsa_stack_reserve(plsa_ctx *ctx, int64_t slots, int64_t m, int32_t k, void **base_device);
int plsa_comm_allgather_stack(plsa_ctx *ctx, int64_t slots, int64_t m, int32_t k, float **host);
/* the same gather with the caller's own host array as the destination ([slots * world][k][m] floats, e.g. the NumPy array
 * that np.vstack would have returned, enstop_.py:231): one pass instead of page-locked buffer + copy */
int plsa_comm_allgather_stack_to(plsa_ctx *ctx, int64_t slots, int64_t m, int32_t k, float *dst);
int plsa_comm_allgather_host(plsa_ctx *ctx, const void *send, int64_t bytes, void *recv /* world * bytes */);
int plsa_comm_broadcast_host(plsa_ctx *ctx, void *buf, int64_t bytes, int32_t root);
int plsa_allreduce_accumulator(plsa_ctx *ctx);

/* frees the large scratch buffers (materialised P, column-pass partials, the ensemble member stack and its DEVICE gather
 * buffers); they are re-created on demand.  The page-locked host buffer whose address plsa_comm_allgather_stack handed
 * out stays valid. */
int plsa_release_scratch(plsa_ctx *ctx);

/* ---- topic combination --------------------------------------------------
 * plsa_all_pairs_hellinger <- enstop/enstop_.py:258-266: the all-pairs Hellinger distance matrix of the
 *   stacked ensemble topics (umap.distances.hellinger's definition), topics [t, m] float32 on the host,
 *   D [t, t] float64 on the host.  Rows with zero mass: distance 1 to any other row, 0 to each other.  */
int plsa_all_pairs_hellinger(plsa_ctx *ctx, const float *topics, int64_t t, int64_t m, double *D);
/* plsa_all_pairs_kl <- all_pairs_kl_divergence, enstop/enstop_.py:234-253: D[i, j] = KL(topic i || topic j)
 *   in bits over the words where both are positive; D [t, t] float64 on the host, not symmetric.      */
int plsa_all_pairs_kl(plsa_ctx *ctx, const float *topics, int64_t t, int64_t m, double *D);
/* plsa_cluster_representatives <- enstop/enstop_.py:299-308, 340-345 (weights == NULL) and 385-393
 *   (weights = HDBSCAN membership strengths [t]): for each cluster c in [0, n_clusters) the weighted mean
 *   of the square-rooted member topics (labels[i] == c; negative labels are noise), squared and
 *   L1-normalised; out [n_clusters, m] float32 on the host.                                           */
int plsa_cluster_representatives(plsa_ctx *ctx, const float *topics, int64_t t, int64_t m,
                                 const int32_t *labels, const double *weights, int32_t n_clusters, float *out);

#ifdef __cplusplus
}
#endif
#endif /* PLSA_HIP_H */
