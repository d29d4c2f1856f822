/*
 * plsa_hip_diag.h -- diagnostics, measurement hooks, test plumbing and synthetic corpora of libplsa_hip.so.
 *
 * NOT part of the drop-in: a maintainer binding the reference's seam (enstop/enstop_.py:52-53) needs include/plsa_hip.h
 * only.  What is declared here has no counterpart in the reference; it exists for bench.py (timings, bandwidth ceilings,
 * the synthetic corpora of SURVEY.md section 8d), for the parity tests (read-backs, a given P(z|w,d), host emulation of
 * the all-reduce, NumPy-identity of the device initialisation) and for reports (schedule, placement, queues).
 * Same conventions as plsa_hip.h: status codes, plsa_last_error(), borrowed host arrays, opaque context.
 */
#ifndef PLSA_HIP_DIAG_H
#define PLSA_HIP_DIAG_H

#include "plsa_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* the GPU_MAX_HW_QUEUES value this process runs with (4 = the HIP runtime's default; see plsa_hip.h, lifetime) */
int plsa_hw_queues(void);
/* 64-char device name ("AMD Instinct MI355X"), gcnArchName, CU count and HBM bytes. */
int plsa_device_info(plsa_ctx *ctx, char *name64, char *arch64, int *cus, int64_t *hbm_bytes);

/* read back the ACTIVE matrix (the uploaded corpus, a bootstrap resample, a synthetic corpus) in the layout of plsa_upload_csr */
int plsa_download_active_csr(plsa_ctx *ctx, int32_t *indptr, int32_t *indices, float *data);

/* Throughput-mode alternative to plsa_init(random) + plsa_set_factors: uniform draws from a
 * counter-based generator, rows L1-normalised, entirely on the device.  NOT the reference's NumPy
 * MT19937 stream -- use plsa_set_factors for seed-for-seed parity (enstop/plsa.py:455-456).      */
int plsa_init_factors_device(plsa_ctx *ctx, int32_t k, uint64_t seed);

/* Diagnostics: the k float64 topic marginals (enstop/utils.py:24-29, `marginal[i] += ndarray[i, j]` left to right) that the
 * last plsa_init_factors_mt19937 call divided by.  The device evaluates that sequential sum from per-chunk parity pairs
 * (csrc/plsa_kernels.hpp: k_mt_chunk_pairs); the tests pin it bit for bit to numpy's own sequential accumulation.   */
int plsa_mt_marginals(plsa_ctx *ctx, double *out, int32_t k);

/* uploads a host P(z|w,d) [nnz,k] in place of plsa_e_step's output (lets plsa_m_step be tested in isolation against
 * enstop/plsa.py:124-204) */
int plsa_set_p(plsa_ctx *ctx, const float *P);

/* host copies of the un-normalised P(w|z) accumulator [m, kp] of the doc-sharded fit (plsa_accumulator_device): a host-side
 * emulation of the all-reduce for tests without a communicator */
int plsa_accumulator_get(plsa_ctx *ctx, float *host);
int plsa_accumulator_set(plsa_ctx *ctx, const float *host);

/* RCCL's own text for the last failure in this process (ncclGetLastError); ctx may be NULL.  No reference
 * counterpart (dask / joblib raise Python exceptions, enstop_.py:209-217); read by enstop_amd/comm.py::report_failure
 * so that a failed multi-GPU start says which stage and why. */
int plsa_comm_last_error(plsa_ctx *ctx, char *buf, int64_t cap);

/* barrier and float64 all-reduce (op 0 sum, 1 max) over the context's communicator, staged through HBM: the bracketing and the
 * max-over-ranks of bench.py's timed regions.  Identity without a communicator. */
int plsa_comm_barrier(plsa_ctx *ctx);
int plsa_comm_allreduce_f64(plsa_ctx *ctx, double *inout, int64_t count, int32_t op);

/* PLSA_REFERENCE_SUMS: norm_pwz[z] -- the reference's one float32 running sum over all non-zeros, enstop/plsa.py:193 -- is evaluated
 * from per-chunk (parity -> increment) pairs and a walk that checks every chunk; chunks that fail the check (binade crossings, a
 * chain that has drifted from the real sums) are added addend by addend.  Reports, over the walks finished so far on this
 * context: chunks that took the slow way, chunks in all, and whether the context has gone back to the serial chain for the
 * current corpus (more than a quarter slow; PLSA_REF_CHAIN=pairs / serial pins either).  Same bits whichever way.  Waits for
 * the context's streams.  Any pointer may be NULL. */
int plsa_reference_chain_info(plsa_ctx *ctx, int64_t *slow_chunks, int64_t *chunks, int32_t *serial_now);

/* The materialised P array is placed by probing: up to PLSA_PLACEMENT_CANDIDATES (default 4)
 * allocations are streamed through once and the fastest is kept (HBM placement alone moves the
 * E-step by ~15 %).  Reports the last probe: candidates tried and the fill
 * bandwidth of the kept / the worst candidate (0 when no probing took place).                    */
int plsa_placement_info(plsa_ctx *ctx, int32_t *candidates, double *best_gbps, double *worst_gbps);

/* Schedule of the column pass for the current structure (diagnostics; bench.py reports it): the visiting list is
 * walked in chunks, XCD x takes the chunks [xcd_lo[x], xcd_lo[x+1]); the boundaries are MEASURED -- timed launches
 * of the pass itself, stretches resized until the eight XCDs finish together (csrc/plsa_hip.hip::ensure_balance).
 * xcd_end_us: per-XCD finish times of the last timed launch (0 when none ran: small corpora, PLSA_BALANCE=0).
 * Results never depend on the boundaries.  No counterpart in the reference (its loops are per-thread ranges of
 * numba.prange, enstop/plsa.py:91).  Any pointer may be NULL.                                               */
int plsa_schedule_info(plsa_ctx *ctx, int32_t *xcd_lo /*[9]*/, double *xcd_end_us /*[8]*/, int32_t *timed_launches,
                       int32_t *item_len, int64_t *n_items);

/* ---- measurement --------------------------------------------------------------------------------
 * HIP events on the context's own stream around every kernel launch (bench.py roofline figures).  */
int plsa_timing_enable(plsa_ctx *ctx, int32_t on);
int plsa_timing_reset(plsa_ctx *ctx);
/* total milliseconds and launch count of kernels whose name starts with `prefix`. */
int plsa_timing_get(plsa_ctx *ctx, const char *prefix, double *total_ms, int64_t *launches);
/* newline-separated "name launches total_ms" report into buf. */
int plsa_timing_report(plsa_ctx *ctx, char *buf, int64_t cap);
/* achievable streaming bandwidth of this device, GB/s, over `bytes` of scratch HBM:
 * kind 0 = fill with non-temporal stores, 1 = fill with plain stores, 2 = copy (bytes read + bytes
 * written counted), 3 = read-only stream (small sizes probe the L2 / Infinity-Cache service rate),
 * 4 / 5 / 6 = non-temporal fill in the E-step's store order (each wave writes 16 / 4 / 64 consecutive
 * 1-KB rows before moving on).
 * The practical ceiling the E-step's P write is compared with.   */
int plsa_measure_stream_bandwidth(plsa_ctx *ctx, int64_t bytes, int32_t kind, int32_t reps, double *gbps);

/* ---- host helper ---------------------------------------------------------------------------------
 * plsa_host_normalize_rows <- enstop/utils.py:8-41 normalize(ndarray, axis=1): float64, in place,
 *   sequential marginal, used by the factor initialisation (enstop/plsa.py:510-511, 980).          */
void plsa_host_normalize_rows(double *a, int64_t rows, int64_t cols);

/* plsa_host_mt19937_jump: advance a numpy.random.RandomState key (624 words) by 624 * 2^log2_blocks
 *   outputs with the jump polynomial the device initialisation uses (csrc/mt_jump.hpp); the position
 *   inside the block is unaffected by a whole-block jump.  Host-only (tests pin the polynomial
 *   arithmetic against NumPy without a GPU).  Returns 0, or 1 if log2_blocks is outside [0, 40].   */
int plsa_host_mt19937_jump(uint32_t *key /*[624]*/, int32_t log2_blocks);

/* synthetic bag-of-words CSR generated on the device (bench.py / large-size tests; not part of the
 * reference): lognormal document lengths, Zipf(s) word ids, the stored count of a (doc, word) pair is its
 * multiplicity among the document's token draws (a multinomial bag of words).  The result
 * becomes base + active matrix.  nnz_target is approximate; the exact nnz is returned.            */
int plsa_generate_synthetic(plsa_ctx *ctx, int64_t n, int64_t m, int64_t nnz_target, double zipf_s,
                            uint64_t seed, int64_t *nnz_out);
/* the same with TOPICAL structure (round 5; the corpus above draws every token independently -- no co-occurrence, unlike
 * text such as the reference's 20-Newsgroups, notebooks/EnsTop with 20-Newsgroups.ipynb:49): document d draws a topic
 * mixture theta_d ~ Dirichlet(alpha) over k0 latent topics, every topic has its own Zipf(s) ranking of the vocabulary,
 * a token comes from the shared ranking with probability `background` and otherwise from topic t ~ theta_d: the
 * generative model pLSA assumes.  1 <= k0 <= 256, alpha > 0, 0 <= background <= 1; deterministic in all arguments. */
int plsa_generate_synthetic_topics(plsa_ctx *ctx, int64_t n, int64_t m, int64_t nnz_target, double zipf_s,
                                   uint64_t seed, int32_t k0, double alpha, double background, int64_t *nnz_out);
/* ground truth of the topical corpus currently held as the base matrix: out[d] = the latent topic with the largest share
 * of document d's mixture theta_d (tests: does a fit recover the planted structure; experiments: document orderings). */
int plsa_synthetic_dominant_topics(plsa_ctx *ctx, int32_t *out /* [n], host */);

#ifdef __cplusplus
}
#endif
#endif /* PLSA_HIP_DIAG_H */
