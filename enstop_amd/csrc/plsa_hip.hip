// plsa_hip.hip -- host side of libplsa_hip.so: context, HBM layout, kernel dispatch, EM drivers and
// the C ABI declared in include/plsa_hip.h (the drop-in) and include/plsa_hip_diag.h (diagnostics, measurement, test
// plumbing).  gfx950 only; no other back-end exists.
//
// HBM layout per context (n docs, m words, k topics, kp = 4*ceil(k/4)):
//   base CSR     indptr i32[n+1], col i32[nnz], val f32[nnz]      the uploaded corpus
//   active CSR   same arrays for the matrix the EM runs on (== base, or a bootstrap resample)
//   rowidx       i32[nnz]   COO row ids of the active matrix (nnz-parallel E-step)
//   CSC copy     colptr i32[m+1], csc_row/csc_pos i32[nnz], csc_val f32[nnz], column items
//                (built lazily; not needed with PLSA_ATOMIC_V)
//   U[2(+1)]     f32[n,kp]  P(z|d), double-buffered (a rejected iteration is simply not swapped in); a third buffer while a
//                fit of a small corpus runs one iteration ahead of its likelihood tests (plsa_fit)
//   Vt[2(+1)], Vacc  f32[m,kp]  P(w|z) word-major, buffered like U, + the un-normalised accumulator
//   P            f32[nnz,kp] materialised responsibilities (only when not PLSA_FUSED)
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <rccl/rccl.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <thread>
#include <type_traits>
#include <vector>

#include "../../include/plsa_hip.h"
#include "../../include/plsa_hip_diag.h"
#include "mt_jump.hpp"
#include "plsa_kernels.hpp"
#include "plsa_ref_kernels.hpp"
#include "plsa_synth.hpp"

using plsa::i64;

namespace {

thread_local std::string g_err;  // errors before a context exists

// Ensemble members of a small corpus are fitted concurrently on several contexts of ONE device (two streams each).  The
// HIP runtime multiplexes all streams of a process onto GPU_MAX_HW_QUEUES = 4 hardware queues by default, and streams that
// share a queue run in submission order: a member's initialisation chain then stalls another member's EM kernels (20NG
// shape, four contexts: 7 060 -> 7 770 ... 8 500 fits/min through ensemble_of_topics; single fits unchanged).  Rounds 3-4
// set GPU_MAX_HW_QUEUES=8 from a constructor of this library; a drop-in must not edit its host's environment behind its
// back, so since round 5 the LIBRARY never does: enstop_amd/_lib.py (the Python host layer) sets it before loading the
// library unless ENSTOP_AMD_HW_QUEUES=0, and a C host that wants concurrent members calls
// setenv("GPU_MAX_HW_QUEUES", "8", 0) itself before its first HIP call.  plsa_hw_queues() reports what
// this process runs with.

struct DevBuf {
    void *p = nullptr;
    size_t cap = 0;
    template <class T> T *as() const { return reinterpret_cast<T *>(p); }
};

struct Timed {
    int name_id;
    hipEvent_t a, b;
};

}  // namespace

struct plsa_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipStream_t stream2 = nullptr;   // column-side chain of the fused iteration (overlaps the document pass)
    hipStream_t ls = nullptr;        // stream the kernel wrappers currently launch on
    hipEvent_t ev_fork = nullptr, ev_join = nullptr;
    hipEvent_t ev_row = nullptr, ev_tail = nullptr;   // pipelined small-corpus iteration: document pass done / column chain done
    bool overlap = true;
    double overlap_full_limit = 2e9;   // nnz * kp below which both passes run side by side (PLSA_OVERLAP_FULL_LIMIT)
    std::string err;
    hipDeviceProp_t prop;
    int grid_cap = 2048;

    // corpus
    i64 bn = 0, bm = 0, bnnz = 0;
    DevBuf b_indptr, b_col, b_val;
    bool active_is_base = true;
    i64 n = 0, m = 0, nnz = 0;
    DevBuf a_indptr, a_col, a_val, rowidx;
    const int *indptr = nullptr, *col = nullptr;
    const float *val = nullptr;
    bool rowidx_valid = false;

    // CSC copy + column items
    bool csc_valid = false;
    int seg = 256, seg_override = 0, struct_lpn = 0;   // column item length: adaptive unless PLSA_COL_SEG is set
    i64 n_items = 0;
    DevBuf colptr, csc_row, csc_val, csc_pos, item_first, item_col, item_start, item_order, partial, heavy_cols;
    bool use_item_order = true, xcd_split = true;
    int chunks_per_lane = 2;
    int row_lpn = 1, row_ch = 1;     // lane shape of the DOCUMENT pass (may differ from lpn / ch: see set_shape)
    bool row_shape_8x2 = true;       // PLSA_ROW_SHAPE=0: document pass in the common shape
    bool p_borrowed = false;       // P(z|w,d) lives in memory lent by plsa_p_borrow (never freed, never re-allocated here)
    bool p_lent = false;           // plsa_p_reserve handed this context's P(z|w,d) address out: it must not move (no regrowth) until plsa_release_scratch
    int e_rows = -1;               // E-step traversal: 1 document-owned, 0 one group per non-zero, -1 by size (PLSA_E_ROWS)
    int mt_streams = 256;          // pieces the MT19937 init stream is cut into (PLSA_MT_STREAMS; 1 = sequential)
    i64 mt_min_blocks = 4096;      // ... once it is at least this many 624-word blocks long (PLSA_MT_MIN_BLOCKS)
    int heavy_items = 32, n_heavy = 0;

    // row items (documents cut into pieces) for corpora with few / very uneven rows
    bool force_wide = false;           // PLSA_FORCE_WIDE: 64-bit gather addresses whatever the table size
    bool ritems_valid = false, use_ritems = false;
    int rseg = 64, rseg_override = 0, ritems_mode = -1;   // ritems_mode: -1 auto, 0 never, 1 always (PLSA_ROW_ITEMS); rseg: entries per row item (PLSA_ROW_SEG, 0 = by size)
    i64 n_ritems = 0;
    DevBuf ritem_first, ritem_row, ritem_start, rpartial;

    // items of the document-owned E-step: documents cut into pieces of eseg entries (balanced: the four
    // groups of a wave all run the same number of gather/store bursts; measured 5.27 -> 4.62 ms at config 3)
    bool eitems_valid = false;
    int eseg = 0, eseg_override = -1;      // PLSA_E_SEG: -1 auto, 0 whole documents, N pieces of N entries
    i64 n_eitems = 0;
    DevBuf eitem_row, eitem_start;

    // rows in descending-length order (row-owned kernels: groups of a wave finish together)
    bool sort_rows = true, roworder_valid = false;
    DevBuf row_order;
    // parameters of the last topical corpus generated on this context (plsa_synthetic_dominant_topics)
    i64 syn_n = 0; int syn_k0 = 0; double syn_alpha = 0.0; uint64_t syn_seed = 0;
    bool row_xcd = false;            // PLSA_ROW_XCD=1 (experiment): XCD x walks the x-th eighth of the documents (k_row_pass)
    int roworder_range = 0;          // documents per range the current row_order was built for (0: plain length order)

    // factors
    int k = 0, kp = 0, lpn = 1, ch = 1;
    DevBuf U[3], Vt[3], Vacc;      // [2]: third buffers of the speculating fit loop (plsa_fit); cu, cv stay in {0, 1} outside it
    bool rot3 = false;             // inside that loop: the output buffers are (cu + 1) % 3, (cv + 1) % 3
    int speculate = -1;            // PLSA_SPECULATE: -1 small corpora only, 0 never, 1 whenever the loop allows
    hipEvent_t ev_ll = nullptr;    // likelihood of a test has reached the host buffer
    int cu = 0, cv = 0;
    i64 fac_n = 0, fac_m = 0;
    DevBuf P;
    // arithmetic of the kernel-level operators and drivers (plsa_set_arithmetic; PLSA_REFERENCE_SUMS / PLSA_REFERENCE_LL of plsa_fit):
    // ref_sums: every factor sum one float32 accumulator in the reference's loop order (plsa_ref_kernels.hpp)
    // ref_ll:   the log-likelihood one float32 running sum over the non-zeros (plsa.py:322 read literally)
    bool ref_sums = false, ref_ll = false;
    DevBuf ref_terms;              // float32 [nnz]: per-entry log-likelihood terms of the sequential sum
    // norm_pwz of the reference arithmetic from per-chunk parity pairs (k_ref_pair_*): PLSA_REF_CHAIN = pairs | serial | auto
    // (auto, default: pairs from 4096 non-zeros on, back to the serial chain for the rest of a corpus' fits once more than a
    // quarter of the chunks of an iteration took the walk's slow way -- chains that drift too far from the real sums)
    int ref_chain_mode = 0;        // 0 auto, 1 always pairs, 2 always the serial chain
    bool ref_pairs_off = false;    // auto mode: the current corpus went back to the serial chain
    DevBuf ref_csum, ref_pairs, ref_exps, ref_pairs2, ref_exps2, ref_stats, ref_ll_neg, ref_heavy, ref_tsum;
    // tile sums of x * P(z|w,d) [* sw] left by the last reference-arithmetic E-step (valid for THAT P and THOSE weights only)
    bool ref_tsum_valid = false;
    const float *ref_tsum_sw = nullptr, *ref_e_sw = nullptr;   // weights the sums were formed with / the next E-step should use
    int ref_tsum_tj = 0;
    bool ref_e_no_sums = false;      // the E-steps of a refit: no norm_pwz chain follows
    bool ref_heavy_valid = false;
    int n_ref_heavy = 0, ref_heavy_min = 0;
    unsigned long long *h_ref_stats = nullptr;   // pinned [2]: chunks that took the slow way / chunks, of the last finished walk
    hipEvent_t ev_ref_stats = nullptr;
    bool ref_stats_pending = false;
    unsigned long long ref_slow_total = 0, ref_chunks_total = 0;   // accumulated over the walks read back so far (plsa_reference_chain_info)
    int placement_candidates = 4, placement_tried = 0;
    double placement_gbps[2] = {0.0, 0.0};
    size_t p_shift = 0;   // experiment knob: byte offset of P inside its allocation (PLSA_P_OFFSET_KB)
    bool p_valid = false;

    // small buffers
    DevBuf sw, ll_partials, ll_out, colsum_partials, norm_pwz, norm_pdz, tmp0, tmp1, tmp2, cubtmp;
    DevBuf mt_seq;                                // chunk sums / parity pairs / binade guesses of the topic marginals
    bool mt_chain = false;                        // PLSA_MT_CHAIN
    DevBuf mt_words, mt_state, mt_fin, mt_poly;   // MT19937 initialisation scratch: kept between calls (an ensemble member per call:
                                                  // four hipMalloc + four hipFree per member cost more than the generator kernels)
    double *h_ll = nullptr;  // pinned

    DevBuf item_end, colsum_rows, colsum_rows2;
    // column-pass schedule: visiting-order item records, chunk boundaries per XCD (measured, see ensure_balance)
    DevBuf item_rec, xcd_lo, t_end;
    bool pipeline = true;            // PLSA_PIPELINE=0: fork/join form of the small-corpus iteration (A/B)
    bool graph = false;              // PLSA_GRAPH=1: hipGraph replay of the iterations between two likelihood tests
    int order_band = -1;             // PLSA_ORDER_BAND: documents per band of the visiting order (-1 auto, 0 first-document order)
    int balance = -1;                // PLSA_BALANCE: -1 auto (large problems), 0 equal stretches, 1 always measure
    bool bal_valid = false;          // xcd_lo matches the current structure
    bool bal_have_frac = false;      // bal_frac holds measured boundaries (kept across bootstrap resamples as the start)
    double bal_frac[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    int bal_lo[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    i64 bal_chunks = -1;
    int bal_launches = 0;            // timed tuning launches spent on the current structure
    double bal_end_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // per-XCD finish times of the last timed launch
    int small_grid = 0;              // PLSA_SMALL_GRID: workgroups per CU of the column pass on small corpora (0 = no cap)
    int colsum_rows_used = 0;        // rows of colsum_rows written by the last column pass

    // multi-GPU exchange: one RCCL communicator per context (one process per GPU), collectives are
    // enqueued on the context's own streams
    ncclComm_t comm = nullptr;
    int comm_rank = 0, comm_world = 1;
    bool sharded = false;            // PLSA_SHARDED fit in progress: accumulators / likelihoods are all-reduced
    bool sw_resident = false;   // plsa_set_sample_weight: sw_res holds weights that apply whenever a call passes sw = NULL
    i64 sw_n = 0;               // (their own buffer: a call that passes explicit weights stages them in c->sw and leaves these alone)
    DevBuf sw_res;
    DevBuf comm_send, comm_recv, comm_small, comm_stack;   // comm_stack: the member stack (plsa_stack_reserve)
    float *comm_host = nullptr;      // pinned landing buffer of plsa_comm_allgather_stack
    size_t comm_host_cap = 0;

    // timing
    bool timing = false;
    std::vector<std::string> names;
    std::vector<Timed> timed;
    std::vector<hipEvent_t> pool;
    std::vector<double> acc_ms;
    std::vector<i64> acc_n;
};

namespace {

inline float *p_base(plsa_ctx *c) { return reinterpret_cast<float *>(reinterpret_cast<char *>(c->P.p) + c->p_shift); }

int fail(plsa_ctx *c, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf; else g_err = buf;
    return 1;
}

#define HIPCHK(c, expr)                                                                            \
    do {                                                                                           \
        hipError_t e_ = (expr);                                                                    \
        if (e_ != hipSuccess)                                                                      \
            return fail((c), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__,      \
                        __LINE__);                                                                 \
    } while (0)

#define CHK(expr)                                                                                  \
    do {                                                                                           \
        int r_ = (expr);                                                                           \
        if (r_) return r_;                                                                         \
    } while (0)

#define NCCLCHK(c, expr)                                                                           \
    do {                                                                                           \
        ncclResult_t n_ = (expr);                                                                  \
        if (n_ != ncclSuccess)                                                                     \
            return fail((c), "%s failed: %s (%s:%d)", #expr, ncclGetErrorString(n_), __FILE__,     \
                        __LINE__);                                                                 \
    } while (0)

int grid_for(plsa_ctx *c, i64 work_items, int items_per_block);
bool g_contig = false;   // PLSA_CONTIG=1: ask for physically contiguous HBM for large buffers

int ensure(plsa_ctx *c, DevBuf &b, size_t bytes) {
    if (bytes == 0) bytes = 16;
    if (b.cap >= bytes) return 0;
    // a buffer that GROWS is one whose size follows the data of the moment (the non-zeros of a bootstrap resample vary by
    // a fraction of a percent from member to member): 6 % head-room ends the hipFree + hipMalloc pairs (device-wide
    // synchronisations) after the first few members.  First allocations and buffers of 1 GB or more stay exact.
    if (b.p && bytes < ((size_t)1 << 30)) bytes += bytes / 16;
    if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    if (g_contig && bytes >= ((size_t)64 << 20)) {
        if (hipExtMallocWithFlags(&b.p, bytes, hipDeviceMallocContiguous) == hipSuccess) { b.cap = bytes; return 0; }
        (void)hipGetLastError();
        b.p = nullptr;
    }
    HIPCHK(c, hipMalloc(&b.p, bytes));
    b.cap = bytes;
    return 0;
}

// HBM placement matters for the streamed P array: the same kernel on the same data ran 6.3 ... 7.4 ms
// depending only on which physical pages hipMalloc happened to hand out (tools/p_offset_probe*.py;
// the start offset inside one allocation is irrelevant).  For that one buffer the engine therefore
// allocates a few candidates, streams a non-temporal fill through each (~5 ms per 25 GB) and keeps
// the fastest.  Candidates are held simultaneously so the allocator cannot return the same pages.
int ensure_best_placement(plsa_ctx *c, DevBuf &b, size_t bytes, int max_candidates, double *gbps, int *n_tried) {
    if (b.cap >= bytes) return 0;
    if (gbps) { gbps[0] = gbps[1] = 0.0; }
    if (n_tried) *n_tried = 0;
    if (b.p) { HIPCHK(c, hipFree(b.p)); b.p = nullptr; b.cap = 0; }
    size_t free_b = 0, total_b = 0;
    int ncand = 1;
    if (max_candidates > 1 && hipMemGetInfo(&free_b, &total_b) == hipSuccess)
        ncand = (int)std::min<size_t>((size_t)max_candidates, (size_t)((double)free_b * 0.6) / std::max<size_t>(bytes, 1));
    if (ncand < 2 || bytes < ((size_t)256 << 20)) return ensure(c, b, bytes);
    std::vector<void *> cand;
    std::vector<float> ms;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const i64 n4 = (i64)(bytes / 16);
    const int grid = grid_for(c, n4, 256);
    for (int i = 0; i < ncand; ++i) {
        void *p = nullptr;
        if (hipMalloc(&p, bytes) != hipSuccess) { (void)hipGetLastError(); break; }
        cand.push_back(p);
        hipLaunchKernelGGL((plsa::k_probe_fill<true>), dim3(grid), dim3(256), 0, c->stream, (float *)p, n4);   // first touch
        (void)hipEventRecord(e0, c->stream);
        hipLaunchKernelGGL((plsa::k_probe_fill<true>), dim3(grid), dim3(256), 0, c->stream, (float *)p, n4);
        hipLaunchKernelGGL((plsa::k_probe_fill<true>), dim3(grid), dim3(256), 0, c->stream, (float *)p, n4);
        (void)hipEventRecord(e1, c->stream);
        (void)hipStreamSynchronize(c->stream);
        float t = 0.f;
        (void)hipEventElapsedTime(&t, e0, e1);
        ms.push_back(t / 2.f);
    }
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (cand.empty()) return ensure(c, b, bytes);
    size_t best = 0, worst = 0;
    for (size_t i = 1; i < cand.size(); ++i) { if (ms[i] < ms[best]) best = i; if (ms[i] > ms[worst]) worst = i; }
    for (size_t i = 0; i < cand.size(); ++i) if (i != best) (void)hipFree(cand[i]);
    b.p = cand[best]; b.cap = bytes;
    if (gbps) { gbps[0] = bytes / 1e9 / (ms[best] / 1e3); gbps[1] = bytes / 1e9 / (ms[worst] / 1e3); }
    if (n_tried) *n_tried = (int)cand.size();
    return 0;
}

// ---------------------------------------------------------------------------------------------
// Large host <-> device copies.  The boundary takes ordinary (pageable) host arrays -- NumPy's -- and a plain hipMemcpy
// of such memory ran at 25 GB/s up / 10 GB/s down on the GPU box (profiles/r05_pcie_inclusive_plsa_fit_from_host.jsonl: 60 of
// the 72 ms a config-3 fit spends outside its iterations).  From STAGE_MIN bytes on, a copy is cut into 8-MB chunks that
// STAGE_THREADS helper threads move through page-locked slots of their own (two each, one HIP stream each): the host-side
// memcpy of one chunk (and the first-touch page faults of a fresh destination array) overlaps the PCIe transfer of the
// others.  The slots are process-wide (64 MB page-locked once), copies of different contexts take turns.
// ---------------------------------------------------------------------------------------------
constexpr size_t STAGE_CHUNK = (size_t)8 << 20;
constexpr int STAGE_THREADS_MAX = 8;
int g_stage_threads = 4;          // helper threads in use (PLSA_STAGE_THREADS, 1 .. 8)
constexpr size_t STAGE_MIN = (size_t)48 << 20;
struct HostStage {
    std::mutex mu;
    int device = -1;
    void *slot[STAGE_THREADS_MAX][2] = {};
    hipStream_t stream[STAGE_THREADS_MAX] = {};
    hipEvent_t ev[STAGE_THREADS_MAX][2] = {};
    bool ok = false, tried = false;
} g_stage;

bool stage_ready(int device) {        // (g_stage.mu held)
    if (g_stage.tried && g_stage.device == device) return g_stage.ok;
    if (g_stage.tried) {              // another device: rebuild the streams / events there, keep the host slots
        for (int t = 0; t < STAGE_THREADS_MAX; ++t) {
            if (g_stage.stream[t]) (void)hipStreamDestroy(g_stage.stream[t]);
            for (int b = 0; b < 2; ++b) if (g_stage.ev[t][b]) (void)hipEventDestroy(g_stage.ev[t][b]);
            g_stage.stream[t] = nullptr; g_stage.ev[t][0] = g_stage.ev[t][1] = nullptr;
        }
    }
    g_stage.tried = true; g_stage.device = device; g_stage.ok = true;
    const char *off = getenv("PLSA_STAGED_COPIES");
    if (off && atoi(off) == 0) { g_stage.ok = false; return false; }
    const char *nt = getenv("PLSA_STAGE_THREADS");
    if (nt) g_stage_threads = std::max(1, std::min(STAGE_THREADS_MAX, atoi(nt)));
    for (int t = 0; t < g_stage_threads && g_stage.ok; ++t) {
        for (int b = 0; b < 2; ++b) {
            if (!g_stage.slot[t][b] && hipHostMalloc(&g_stage.slot[t][b], STAGE_CHUNK, hipHostMallocDefault) != hipSuccess) g_stage.ok = false;
            if (hipEventCreateWithFlags(&g_stage.ev[t][b], hipEventDisableTiming) != hipSuccess) g_stage.ok = false;
        }
        if (hipStreamCreateWithFlags(&g_stage.stream[t], hipStreamNonBlocking) != hipSuccess) g_stage.ok = false;
    }
    if (!g_stage.ok) (void)hipGetLastError();
    return g_stage.ok;
}

// dev <- host (to_device) or host <- dev, `bytes` contiguous on both sides.  The caller has synchronised whatever produced
// the source; on return the data is in place (every helper stream drained).  Returns false if the staged path is unavailable.
bool staged_copy(plsa_ctx *c, void *dev, void *host, size_t bytes, bool to_device) {
    if (bytes < STAGE_MIN) return false;
    std::lock_guard<std::mutex> lock(g_stage.mu);
    if (!stage_ready(c->device)) return false;
    const size_t n_chunks = (bytes + STAGE_CHUNK - 1) / STAGE_CHUNK;
    const int STAGE_THREADS = g_stage_threads;
    bool failed[STAGE_THREADS_MAX] = {};
    auto worker = [&](int t) {
        if (hipSetDevice(c->device) != hipSuccess) { failed[t] = true; return; }
        hipStream_t st = g_stage.stream[t];
        if (to_device) {
            int b = 0;
            for (size_t i = t; i < n_chunks; i += STAGE_THREADS, b ^= 1) {
                const size_t off = i * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
                if (hipEventSynchronize(g_stage.ev[t][b]) != hipSuccess) { failed[t] = true; return; }   // the slot's previous transfer
                memcpy(g_stage.slot[t][b], (const char *)host + off, len);
                if (hipMemcpyAsync((char *)dev + off, g_stage.slot[t][b], len, hipMemcpyHostToDevice, st) != hipSuccess ||
                    hipEventRecord(g_stage.ev[t][b], st) != hipSuccess) { failed[t] = true; return; }
            }
        } else {
            // chunk j+1 is on its way into the other slot while chunk j is copied out to the caller's pages
            size_t i = t;
            int b = 0;
            auto issue = [&](size_t ci, int slot) {
                const size_t off = ci * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
                return hipMemcpyAsync(g_stage.slot[t][slot], (const char *)dev + off, len, hipMemcpyDeviceToHost, st) == hipSuccess &&
                       hipEventRecord(g_stage.ev[t][slot], st) == hipSuccess;
            };
            if (i < n_chunks && !issue(i, b)) { failed[t] = true; return; }
            for (; i < n_chunks; i += STAGE_THREADS, b ^= 1) {
                const size_t nxt = i + STAGE_THREADS;
                if (nxt < n_chunks && !issue(nxt, b ^ 1)) { failed[t] = true; return; }
                if (hipEventSynchronize(g_stage.ev[t][b]) != hipSuccess) { failed[t] = true; return; }
                const size_t off = i * STAGE_CHUNK, len = std::min(STAGE_CHUNK, bytes - off);
                memcpy((char *)host + off, g_stage.slot[t][b], len);
            }
        }
        if (hipStreamSynchronize(st) != hipSuccess) failed[t] = true;
    };
    std::thread th[STAGE_THREADS_MAX];
    for (int t = 1; t < STAGE_THREADS; ++t) th[t] = std::thread(worker, t);
    worker(0);
    for (int t = 1; t < STAGE_THREADS; ++t) th[t].join();
    (void)hipSetDevice(c->device);
    for (int t = 0; t < STAGE_THREADS; ++t)
        if (failed[t]) { (void)hipGetLastError(); return false; }      // the caller repeats the copy the plain way
    return true;
}

// host -> device on the context's stream order: everything enqueued on c->stream so far is complete when the staged path
// writes (it waits), and the data is in place when this returns, so later work on c->stream sees it
int copy_to_device(plsa_ctx *c, void *dev, const void *host, size_t bytes) {
    if (bytes >= STAGE_MIN) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        if (staged_copy(c, dev, const_cast<void *>(host), bytes, true)) return 0;
    }
    HIPCHK(c, hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, c->stream));
    return 0;
}

// device -> host; the data is in `host` on return
int copy_to_host(plsa_ctx *c, void *host, const void *dev, size_t bytes) {
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bytes >= STAGE_MIN && staged_copy(c, const_cast<void *>(dev), host, bytes, false)) return 0;
    HIPCHK(c, hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

void release(DevBuf &b) {
    if (b.p) (void)hipFree(b.p);
    b.p = nullptr;
    b.cap = 0;
}

int name_id(plsa_ctx *c, const char *name) {
    for (size_t i = 0; i < c->names.size(); ++i)
        if (c->names[i] == name) return (int)i;
    c->names.emplace_back(name);
    c->acc_ms.push_back(0.0);
    c->acc_n.push_back(0);
    return (int)c->names.size() - 1;
}

hipEvent_t get_event(plsa_ctx *c) {
    if (!c->pool.empty()) { hipEvent_t e = c->pool.back(); c->pool.pop_back(); return e; }
    hipEvent_t e;
    (void)hipEventCreate(&e);
    return e;
}

// folds finished event pairs into the per-kernel totals (requires the stream to be idle)
int timing_flush(plsa_ctx *c) {
    if (c->timed.empty()) return 0;
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    for (auto &t : c->timed) {
        float ms = 0.f;
        HIPCHK(c, hipEventElapsedTime(&ms, t.a, t.b));
        c->acc_ms[t.name_id] += ms;
        c->acc_n[t.name_id] += 1;
        c->pool.push_back(t.a);
        c->pool.push_back(t.b);
    }
    c->timed.clear();
    return 0;
}

struct Scope {  // brackets one kernel launch with events when timing is on
    plsa_ctx *c;
    Timed t;
    bool on;
    Scope(plsa_ctx *c_, const char *name) : c(c_), on(c_->timing) {
        if (on) {
            t.name_id = name_id(c, name);
            t.a = get_event(c);
            t.b = get_event(c);
            (void)hipEventRecord(t.a, c->ls);
        }
    }
    ~Scope() {
        if (on) {
            (void)hipEventRecord(t.b, c->ls);
            c->timed.push_back(t);
        }
    }
};

template <class Fn>
int dispatch_shape(plsa_ctx *c, Fn &&fn) {
    const bool full = c->kp == 4 * c->lpn * c->ch;
#define PLSA_SHAPE(L, H)                                                                           \
    if (c->lpn == L && c->ch == H) {                                                               \
        if (full) fn(plsa::Shape<L, H, true>{}); else fn(plsa::Shape<L, H, false>{});              \
        return 0;                                                                                  \
    }
    PLSA_SHAPE(1, 1) PLSA_SHAPE(2, 1) PLSA_SHAPE(4, 1) PLSA_SHAPE(8, 1) PLSA_SHAPE(16, 1)
    PLSA_SHAPE(32, 1) PLSA_SHAPE(64, 1) PLSA_SHAPE(64, 2) PLSA_SHAPE(64, 4)
    PLSA_SHAPE(16, 2) PLSA_SHAPE(32, 2)
#undef PLSA_SHAPE
    return fail(c, "unsupported topic count k=%d (max 1024)", c->k);
}

// The two fused passes gather rows of a factor table by index with 32-bit byte offsets (plsa_kernels.hpp: gather_row).
// A table of 4 GB or more (rows * kp * 4 >= 2^32: e.g. 20 M documents at k = 64) takes the WIDE instantiations instead:
// 64-bit row addresses, run-time kp -- same arithmetic, same results.  PLSA_FORCE_WIDE=1 selects them for any size (tests).
bool table_is_wide(plsa_ctx *c, i64 rows) { return c->force_wide || (double)rows * c->kp * 4.0 >= 4294967296.0; }

template <class Fn>
int dispatch_shape_gather(plsa_ctx *c, bool wide, Fn &&fn) {
    if (!wide) return dispatch_shape(c, fn);
#define PLSA_SHAPE(L, H)                                                                           \
    if (c->lpn == L && c->ch == H) { fn(plsa::Shape<L, H, false, true>{}); return 0; }
    PLSA_SHAPE(1, 1) PLSA_SHAPE(2, 1) PLSA_SHAPE(4, 1) PLSA_SHAPE(8, 1) PLSA_SHAPE(16, 1)
    PLSA_SHAPE(32, 1) PLSA_SHAPE(64, 1) PLSA_SHAPE(64, 2) PLSA_SHAPE(64, 4)
    PLSA_SHAPE(16, 2) PLSA_SHAPE(32, 2)
#undef PLSA_SHAPE
    return fail(c, "unsupported topic count k=%d (max 1024)", c->k);
}

// lane shape of the document pass (k_row_pass, k_row_reduce); it gathers P(w|z) rows: m of them
template <class Fn>
int dispatch_shape_row(plsa_ctx *c, Fn &&fn) {
    const bool wide = table_is_wide(c, c->m);
    if (c->row_lpn == 8 && c->row_ch == 2 && c->kp == 64) {
        if (wide) fn(plsa::Shape<8, 2, false, true>{}); else fn(plsa::Shape<8, 2, true>{});
        return 0;
    }
    return dispatch_shape_gather(c, wide, fn);
}

int launch_check(plsa_ctx *c, const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, "launch of %s failed: %s", what, hipGetErrorString(e));
    return 0;
}

// buffers the passes write (the alternates of the current factors)
inline int out_u(const plsa_ctx *c) { return c->rot3 ? (c->cu + 1) % 3 : 1 - c->cu; }
inline int out_v(const plsa_ctx *c) { return c->rot3 ? (c->cv + 1) % 3 : 1 - c->cv; }

int grid_for(plsa_ctx *c, i64 work_items, int items_per_block) {
    i64 need = (work_items + items_per_block - 1) / items_per_block;
    if (need < 1) need = 1;
    return (int)std::min<i64>(need, c->grid_cap);
}

// ---------------------------------------------------------------------------------------------
// corpus helpers
// ---------------------------------------------------------------------------------------------
void set_active_pointers(plsa_ctx *c) {
    if (c->active_is_base) {
        c->indptr = c->b_indptr.as<int>(); c->col = c->b_col.as<int>(); c->val = c->b_val.as<float>();
        c->n = c->bn; c->m = c->bm; c->nnz = c->bnnz;
    } else {
        c->indptr = c->a_indptr.as<int>(); c->col = c->a_col.as<int>(); c->val = c->a_val.as<float>();
    }
    c->rowidx_valid = false;
    c->csc_valid = false;
    c->roworder_valid = false;
    c->ritems_valid = false;
    c->eitems_valid = false;
    c->p_valid = false;
    c->ref_pairs_off = false;
    c->ref_heavy_valid = false;
    c->ref_tsum_valid = false;
}

int ensure_rowidx(plsa_ctx *c) {
    if (c->rowidx_valid) return 0;
    CHK(ensure(c, c->rowidx, sizeof(int) * (size_t)c->nnz));
    if (c->n > 0) {
        Scope s(c, "k_expand_rows");
        hipLaunchKernelGGL(plsa::k_expand_rows, dim3(grid_for(c, c->n, 4)), dim3(256), 0, c->stream,
                           c->indptr, (int)c->n, c->rowidx.as<int>());
    }
    CHK(launch_check(c, "k_expand_rows"));
    c->rowidx_valid = true;
    return 0;
}

// row ids sorted by descending row length (stable), or nullptr when sorting is disabled
// documents per range of the XCD-contiguous document schedule for the current corpus and lane shape (0: not in use)
int row_xcd_range(const plsa_ctx *c) {
    const i64 gpb = 256 / std::max(1, c->row_lpn);
    if (!c->row_xcd || c->n < 64 * gpb) return 0;
    return (int)(((c->n + gpb - 1) / gpb + 7) / 8 * gpb);
}

int ensure_roworder(plsa_ctx *c, const int **out) {
    *out = nullptr;
    if (!c->sort_rows) return 0;
    const int range = row_xcd_range(c);
    if (!c->roworder_valid || c->roworder_range != range) {
        c->roworder_range = range;
        const i64 n = c->n;
        CHK(ensure(c, c->row_order, sizeof(int) * (size_t)n));
        CHK(ensure(c, c->tmp0, sizeof(int) * (size_t)n * 2));
        CHK(ensure(c, c->tmp1, sizeof(int) * (size_t)n));
        int *len = c->tmp0.as<int>(), *len_sorted = c->tmp0.as<int>() + n, *ids = c->tmp1.as<int>();
        hipLaunchKernelGGL(plsa::k_row_lengths, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->indptr, (int)n, len, ids, range);
        CHK(launch_check(c, "k_row_lengths"));
        size_t bytes = 0;
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairsDescending(nullptr, bytes, len, len_sorted, ids,
                                                               c->row_order.as<int>(), (int)n, 0, 32, c->stream));
        CHK(ensure(c, c->cubtmp, bytes));
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairsDescending(c->cubtmp.p, bytes, len, len_sorted, ids,
                                                               c->row_order.as<int>(), (int)n, 0, 32, c->stream));
        c->roworder_valid = true;
    }
    *out = c->row_order.as<int>();
    return 0;
}

int exclusive_sum_int(plsa_ctx *c, const int *in, int *out, i64 count);

// Decide whether the document pass should run over row items, and build them.  Row ownership needs
// enough rows to fill 256 CUs x 32 waves x (64/LPN) groups, and rows of comparable length.
int ensure_ritems(plsa_ctx *c) {
    if (c->ritems_valid) return 0;
    const i64 n = c->n;
    const i64 group_slots = (i64)c->prop.multiProcessorCount * 32 * (64 / std::max(1, c->row_lpn));
    const double avg = (double)c->nnz / (double)std::max<i64>(n, 1);
    // the decision is made for 64-entry items (documents averaging more than 128 entries: config 2's 100-entry documents
    // stay whole -- items cost it 15 % in the two-stream schedule); the item LENGTH then follows the size of the corpus:
    // about one item per group slot, a power of two in [16, 64] (20NG shape: 45 entries per slot -> 32; with the final
    // kernels of round 4 row items of 16 / 24 / 32 / 40 / 48 / 64 entries give 9.9 / 10.7 / 11.1 / 11.1 / 10.9 / 10.4 k
    // iterations/s at config 1, profiles/r04_small_corpus_item_lengths.txt)
    const int rseg_decide = c->rseg_override ? c->rseg_override : 64;
    c->use_ritems = c->ritems_mode == 1 || (c->ritems_mode < 0 && n < 2 * group_slots && avg > 2.0 * rseg_decide);
    if (c->rseg_override) c->rseg = c->rseg_override;
    else {
        const i64 per_slot = c->nnz / std::max<i64>(group_slots, 1);
        int r = 16;
        while (r * 2 <= per_slot && r < 64) r *= 2;
        c->rseg = r;
    }
    c->n_ritems = 0;
    if (c->use_ritems) {
        CHK(ensure(c, c->ritem_first, sizeof(int) * (size_t)(n + 1)));
        CHK(ensure(c, c->tmp0, sizeof(int) * (size_t)(n + 1)));
        HIPCHK(c, hipMemsetAsync(c->tmp0.p, 0, sizeof(int) * (size_t)(n + 1), c->stream));
        hipLaunchKernelGGL(plsa::k_item_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->indptr, (int)n, c->rseg, c->tmp0.as<int>());
        CHK(exclusive_sum_int(c, c->tmp0.as<int>(), c->ritem_first.as<int>(), n + 1));
        int cnt = 0;
        HIPCHK(c, hipMemcpyAsync(&cnt, c->ritem_first.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->n_ritems = cnt;
        CHK(ensure(c, c->ritem_row, sizeof(int) * (size_t)std::max(cnt, 1)));
        CHK(ensure(c, c->ritem_start, sizeof(int) * (size_t)std::max(cnt, 1)));
        hipLaunchKernelGGL(plsa::k_ritem_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->indptr, c->ritem_first.as<int>(), (int)n, c->rseg, c->ritem_row.as<int>(),
                           c->ritem_start.as<int>());
        CHK(launch_check(c, "k_ritem_fill"));
    }
    c->ritems_valid = true;
    return 0;
}

// items of the document-owned E-step (its own piece length, independent of the document pass' row items)
int ensure_eitems(plsa_ctx *c, int eseg) {
    if (c->eitems_valid && c->eseg == eseg) return 0;
    const i64 n = c->n;
    c->eseg = eseg;
    c->n_eitems = 0;
    if (eseg > 0) {
        CHK(ensure(c, c->tmp0, sizeof(int) * (size_t)(n + 1)));
        CHK(ensure(c, c->tmp1, sizeof(int) * (size_t)(n + 1)));
        HIPCHK(c, hipMemsetAsync(c->tmp0.p, 0, sizeof(int) * (size_t)(n + 1), c->stream));
        hipLaunchKernelGGL(plsa::k_item_counts, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->indptr, (int)n, eseg, c->tmp0.as<int>());
        CHK(exclusive_sum_int(c, c->tmp0.as<int>(), c->tmp1.as<int>(), n + 1));
        int cnt = 0;
        HIPCHK(c, hipMemcpyAsync(&cnt, c->tmp1.as<int>() + n, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->n_eitems = cnt;
        CHK(ensure(c, c->eitem_row, sizeof(int) * (size_t)std::max(cnt, 1)));
        CHK(ensure(c, c->eitem_start, sizeof(int) * (size_t)std::max(cnt, 1)));
        hipLaunchKernelGGL(plsa::k_ritem_fill, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, c->stream,
                           c->indptr, c->tmp1.as<int>(), (int)n, eseg, c->eitem_row.as<int>(), c->eitem_start.as<int>());
        CHK(launch_check(c, "k_ritem_fill"));
        HIPCHK(c, hipStreamSynchronize(c->stream));      // tmp1 is reused by other structure builds
    }
    c->eitems_valid = true;
    return 0;
}

int exclusive_sum_int(plsa_ctx *c, const int *in, int *out, i64 count) {
    size_t bytes = 0;
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in, out, (int)count, c->stream));
    CHK(ensure(c, c->cubtmp, bytes));
    HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, bytes, in, out, (int)count, c->stream));
    return 0;
}

int ensure_csc(plsa_ctx *c) {
    if (c->csc_valid) return 0;
    CHK(ensure_rowidx(c));
    {   // item length: a group walks seg/LPN dependent gather batches per item, so small problems want
        // short items (enough items to fill the chip: config 1 0.244 -> 0.094 ms at 16) and large
        // ones long items (fewer partial rows: 256 measured best at config 3)
        const i64 slots = (i64)c->prop.multiProcessorCount * 32 * (64 / std::max(1, c->lpn));
        // (round 4, final kernels: about 1.25 group slots per item instead of 4 -- config 1 now cuts its columns into
        //  32-entry items, 16 / 32 / 48 / 64 -> 10.5 / 11.1 / 11.0 / 10.9 k iterations/s with 32-entry row items; config 2
        //  64 instead of 32: the same within noise)
        i64 want = c->nnz / std::max<i64>(slots + slots / 4, 1);
        int seg = 16;
        // large corpora: with the XCD stretches balanced, SHORTER items win (an item then spans fewer documents
        // and stays inside the band its XCD's L2 holds): config 3 (k = 64) 256 / 128 / 96 / 64 / 48 entries ->
        // 268 / 269 / 271 / 274 / 272 iterations/s before the band-major order, flat from 48 to 128 with it; config 5
        // (k = 128) 256 / 128 / 64 -> 25.3 / 27.6 / 28.6.
        // Items of one length for every column: a chunk's groups (and a wave's) wait for their longest item --
        // long items for the Zipf-head words only (256 entries, the others 64) cost 1.96 -> 3.0 ms at config 3
        const int cap = 64;
        while (seg * 2 <= want && seg < cap) seg *= 2;
        c->seg = c->seg_override ? c->seg_override : seg;
    }
    const i64 nnz = c->nnz, m = c->m;
    CHK(ensure(c, c->colptr, sizeof(int) * (size_t)(m + 1)));
    CHK(ensure(c, c->csc_row, sizeof(int) * (size_t)nnz));
    CHK(ensure(c, c->csc_val, sizeof(float) * (size_t)nnz));
    CHK(ensure(c, c->csc_pos, sizeof(int) * (size_t)nnz));
    CHK(ensure(c, c->tmp0, sizeof(int) * (size_t)std::max<i64>(nnz, m + 1)));  // counts, then sorted keys
    CHK(ensure(c, c->tmp1, sizeof(int) * (size_t)nnz));                         // iota
    // stable sort of entry positions by column: within a column entries stay in document order
    if (nnz > 0) {
        hipLaunchKernelGGL(plsa::k_iota, dim3(grid_for(c, nnz, 256)), dim3(256), 0, c->stream,
                           c->tmp1.as<int>(), nnz);
        int bits = 1;
        while (((i64)1 << bits) < m) ++bits;
        size_t bytes = 0;
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, c->col, c->tmp0.as<int>(),
                                                     c->tmp1.as<int>(), c->csc_pos.as<int>(), nnz, 0,
                                                     bits, c->stream));
        CHK(ensure(c, c->cubtmp, bytes));
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(c->cubtmp.p, bytes, c->col, c->tmp0.as<int>(),
                                                     c->tmp1.as<int>(), c->csc_pos.as<int>(), nnz, 0,
                                                     bits, c->stream));
        hipLaunchKernelGGL(plsa::k_colptr_from_sorted, dim3((unsigned)((m + 256) / 256)), dim3(256), 0, c->stream,
                           c->tmp0.as<int>(), nnz, (int)m, c->colptr.as<int>());
        hipLaunchKernelGGL(plsa::k_csc_gather, dim3(grid_for(c, nnz, 256)), dim3(256), 0, c->stream,
                           c->csc_pos.as<int>(), c->rowidx.as<int>(), c->val, nnz,
                           c->csc_row.as<int>(), c->csc_val.as<float>());
        CHK(launch_check(c, "k_csc_gather"));
    } else {
        HIPCHK(c, hipMemsetAsync(c->colptr.p, 0, sizeof(int) * (size_t)(m + 1), c->stream));
    }
    // column items
    CHK(ensure(c, c->item_first, sizeof(int) * (size_t)(m + 1)));
    HIPCHK(c, hipMemsetAsync(c->tmp0.p, 0, sizeof(int) * (size_t)(m + 1), c->stream));
    hipLaunchKernelGGL(plsa::k_col_item_counts, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                       c->colptr.as<int>(), (int)m, c->seg, c->tmp0.as<int>());
    CHK(exclusive_sum_int(c, c->tmp0.as<int>(), c->item_first.as<int>(), m + 1));
    int n_items = 0;
    HIPCHK(c, hipMemcpyAsync(&n_items, c->item_first.as<int>() + m, sizeof(int), hipMemcpyDeviceToHost,
                             c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->n_items = n_items;
    const size_t ni = (size_t)std::max<i64>(n_items, 1);
    CHK(ensure(c, c->item_col, sizeof(int) * ni));
    CHK(ensure(c, c->item_start, sizeof(int) * ni));
    CHK(ensure(c, c->item_end, sizeof(int) * ni));
    CHK(ensure(c, c->item_order, sizeof(int) * ni));
    CHK(ensure(c, c->tmp2, sizeof(unsigned long long) * ni * 2));   // sort keys, sorted keys
    CHK(ensure(c, c->tmp1, sizeof(int) * ni));           // item ids
    unsigned long long *d_key = c->tmp2.as<unsigned long long>();
    // band of the visiting order (PLSA_ORDER_BAND documents): 2 MB of P(z|d) rows from k = 64 on -- half an XCD's L2; 8192
    // documents at config 3, 4096 at config 5.  Measured with the final kernels of round 4, config 3: 2048 / 6144 / 8192 /
    // 10240 / 16384 / 32768 documents -> 313 / 318 / 319 / 316 / 303 / 254 iterations/s; config 5: 1024 / 3072 / 4096 / 6144 ->
    // 29.0 / 29.4 / 29.9 / 29.8 (round 3 chose 512 KB with 256-entry items).  Narrow k-vectors stay at 512 KB (config 2:
    // 8192 documents = 1 MB neutral, 16384 = 2 MB 2 % slower)
    const int band_bytes = (c->kp >= 64 ? 2048 : 512) << 10;
    const int band = c->order_band >= 0 ? c->order_band : std::max(64, band_bytes / (c->kp > 0 ? c->kp * 4 : 256));
    // key = band index << len_bits | inverted column length (a column holds at most n entries); first document when band <= 0
    int len_bits = 1;
    while (((i64)1 << len_bits) <= c->n) ++len_bits;
    int key_bits = len_bits;
    if (band > 0) { i64 n_bands = c->n / band + 1; int bb = 1; while (((i64)1 << bb) < n_bands) ++bb; key_bits = len_bits + bb; }
    if (n_items > 0)
        hipLaunchKernelGGL(plsa::k_item_fill, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, c->stream,
                           c->colptr.as<int>(), c->item_first.as<int>(), (int)m, c->seg, c->csc_row.as<int>(),
                           c->item_col.as<int>(), c->item_start.as<int>(), c->item_end.as<int>(), band, len_bits, d_key,
                           c->tmp1.as<int>(), n_items);
    CHK(launch_check(c, "k_item_fill"));
    if (n_items > 0) {   // visiting order: band-major, Zipf-head words first inside a band (stable)
        size_t bytes = 0;
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(nullptr, bytes, d_key, d_key + ni,
                                                     c->tmp1.as<int>(), c->item_order.as<int>(), n_items, 0, key_bits, c->stream));
        CHK(ensure(c, c->cubtmp, bytes));
        HIPCHK(c, hipcub::DeviceRadixSort::SortPairs(c->cubtmp.p, bytes, d_key, d_key + ni,
                                                     c->tmp1.as<int>(), c->item_order.as<int>(), n_items, 0, key_bits, c->stream));
    }
    // visiting-order records (one 16-byte load per item instead of an index chain)
    CHK(ensure(c, c->item_rec, sizeof(int4) * ni));
    if (n_items > 0) {
        hipLaunchKernelGGL(plsa::k_item_records, dim3((unsigned)((n_items + 255) / 256)), dim3(256), 0, c->stream,
                           c->use_item_order ? c->item_order.as<int>() : nullptr, c->item_col.as<int>(),
                           c->item_start.as<int>(), c->item_end.as<int>(), (i64)n_items, c->item_rec.as<int4>());
        CHK(launch_check(c, "k_item_records"));
    }
    c->bal_valid = false;
    // columns whose item count makes a single group's serial reduction a tail (Zipf head words)
    CHK(ensure(c, c->heavy_cols, sizeof(int) * (size_t)(m + 1)));
    HIPCHK(c, hipMemsetAsync(c->heavy_cols.as<int>() + m, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(plsa::k_heavy_list, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, c->stream,
                       c->item_first.as<int>(), (int)m, c->heavy_items, c->heavy_cols.as<int>(),
                       c->heavy_cols.as<int>() + m);
    CHK(launch_check(c, "k_heavy_list"));
    HIPCHK(c, hipMemcpyAsync(&c->n_heavy, c->heavy_cols.as<int>() + m, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->csc_valid = true;
    return 0;
}

int upload_sw(plsa_ctx *c, const float *sw, const float **d_sw) {
    *d_sw = nullptr;
    if (!sw) {
        // weights made resident by plsa_set_sample_weight: no copy, no host wait (the per-iteration calls of the
        // split doc-sharded loop stay enqueue-only)
        if (c->sw_resident) {
            if (c->sw_n != c->n) return fail(c, "resident sample weights were set for %lld documents, the active matrix has %lld",
                                              (long long)c->sw_n, (long long)c->n);
            *d_sw = c->sw_res.as<float>();
        }
        return 0;
    }
    CHK(ensure(c, c->sw, sizeof(float) * (size_t)c->n));
    HIPCHK(c, hipMemcpyAsync(c->sw.p, sw, sizeof(float) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
    // `sw` is borrowed for the call only (it may be a temporary of the caller): the copy must have left the host
    // buffer before any stream-ordered entry point (plsa_em_accumulate) returns
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *d_sw = c->sw.as<float>();
    return 0;
}

// lane decomposition of a k-vector (see plsa_kernels.hpp) + invalidation of the structures that depend on it
void set_shape(plsa_ctx *c, int k) {
    const int kp = (k + 3) / 4 * 4;
    c->k = k; c->kp = kp;
    const int prev_lpn = c->struct_lpn;
    int lpn = 1;
    while (lpn < kp / 4 && lpn < 64) lpn *= 2;
    // k >= 128: 8 floats per lane (two float4 chunks) -- fewer reduction/shuffle instructions per
    // cell, each access still covers whole 128-B lines (measured: config 5 document pass -15 %)
    if (lpn >= 32 && lpn * 4 >= kp && c->chunks_per_lane == 2) lpn /= 2;
    c->lpn = lpn;
    c->ch = (kp / 4 + lpn - 1) / lpn;
    if (c->ch == 3) c->ch = 4;
    // k = 64: the document pass runs as 8 lanes x 2 chunks (a wave covers 8 documents, one DPP step less per group sum,
    // half the log-likelihood reductions per entry): its LL variant 1.94 -> 1.59 ms, the plain one 1.547 -> 1.530 ms at
    // config 3, while the column pass is 3.5 % SLOWER in that shape (32 items per chunk) and keeps 16 x 1
    c->row_lpn = c->lpn; c->row_ch = c->ch;
    if (c->row_shape_8x2 && c->lpn == 16 && c->ch == 1 && kp == 64) { c->row_lpn = 8; c->row_ch = 2; }
    if (lpn != prev_lpn) {        // item lengths / the row-item decision depend on the lane shape
        c->ritems_valid = false;
        c->eitems_valid = false;
        if (!c->seg_override) c->csc_valid = false;
        c->struct_lpn = lpn;
    }
}

int need_factors(plsa_ctx *c) {
    if (c->k <= 0 || !c->U[0].p || !c->Vt[0].p) return fail(c, "factors not set (call plsa_set_factors)");
    if (c->n <= 0) return fail(c, "no corpus uploaded");
    if (c->fac_n != c->n || c->fac_m != c->m)
        return fail(c, "factors were set for a %lld x %lld matrix but the active matrix is %lld x %lld "
                       "(call plsa_set_factors after plsa_upload_csr / plsa_bootstrap)",
                    (long long)c->fac_n, (long long)c->fac_m, (long long)c->n, (long long)c->m);
    return 0;
}

// ---------------------------------------------------------------------------------------------
// kernel wrappers
// ---------------------------------------------------------------------------------------------
int run_ref_e_step(plsa_ctx *c, float thresh);
int run_ref_norm_pwz(plsa_ctx *c, const float *d_sw);
int ensure_ref_heavy(plsa_ctx *c);
bool ref_pairs_now(const plsa_ctx *c);
int run_ref_m_step(plsa_ctx *c, const float *d_sw, bool update_v, float *d_norm_pdz);
int run_ref_loglik(plsa_ctx *c, const float *d_sw, double *out);

int run_e_step(plsa_ctx *c, float thresh) {
    c->ref_tsum_valid = false;          // (a new P(z|w,d): the tile sums of the last reference-arithmetic E-step are history)
    {   // the materialised schedule needs the whole nnz x kp array: say so instead of a bare OOM
        const size_t need = sizeof(float) * (size_t)(c->nnz + 64) * (size_t)c->kp;
        size_t free_b = 0, total_b = 0;
        if (!c->p_borrowed && c->P.cap < need && hipMemGetInfo(&free_b, &total_b) == hipSuccess && free_b + c->P.cap < need)
            return fail(c, c->ref_sums ? "the reference arithmetic (PLSA_REFERENCE_SUMS) stores P(z|w,d) like the reference does: %.1f GB needed, "
                                         "%.1f GB of HBM free -- at this size only the default arithmetic is available"
                                       : "materialising P(z|w,d) needs %.1f GB but only %.1f GB of HBM are free; use the fused "
                           "schedule (PLSA_FUSED), which never stores it, or tile the documents (plsa_em_accumulate_materialised)", need / 1e9, (free_b + c->P.cap) / 1e9);
    }
    if (c->p_borrowed || c->p_lent) {
        const size_t need = sizeof(float) * (size_t)(c->nnz + 64) * (size_t)c->kp;
        if (c->P.cap < need)
            return fail(c, c->p_borrowed ? "the borrowed P(z|w,d) buffer holds %.2f GB, this matrix needs %.2f GB (plsa_p_borrow)"
                                         : "the P(z|w,d) buffer lent out by plsa_p_reserve holds %.2f GB, this matrix needs %.2f GB: it cannot "
                                           "grow while other contexts hold its address (end the loans, then plsa_release_scratch)",
                        c->P.cap / 1e9, need / 1e9);
        c->p_shift = 0;
    } else
    // one tile (64 rows) of slack: the last tile is stored without a predicate
    {   // placement experiment knobs: PLSA_P_SLACK_MB over-allocates, PLSA_P_OFFSET_KB shifts the start
        const char *s1 = getenv("PLSA_P_SLACK_MB"), *s2 = getenv("PLSA_P_OFFSET_KB");
        const size_t slack = s1 ? (size_t)atoll(s1) << 20 : 0;
        c->p_shift = s2 ? (size_t)atoll(s2) * 1024 : 0;
        CHK(ensure_best_placement(c, c->P, sizeof(float) * (size_t)(c->nnz + 64) * (size_t)c->kp + std::max(slack, c->p_shift),
                                  c->placement_candidates, c->placement_gbps, &c->placement_tried));
    }
    // Two traversals.  Document-owned: a group keeps its document's P(z|d) row in registers and gathers only
    // P(w|z) rows (config 3 5.3 ms against 6.3 ms for one group per non-zero), and with the documents cut
    // into pieces of 4 index batches every group of a wave runs the same number of gather/store bursts
    // (config 3 5.27 -> 4.62 ms = 71 % of the HBM peak; config 2 0.357 -> 0.235 ms = 71 %).  Tiny corpora
    // (config 1: 84 us in all) keep the flat kernel: one group per non-zero, perfectly balanced, no setup.
    if (c->ref_sums) return run_ref_e_step(c, thresh);
    bool e_rows = c->e_rows < 0 ? (double)c->nnz * c->kp >= 1e8 : c->e_rows != 0;
    if (e_rows) {
        // piece length: measured optimum 64 entries at k = 64 (48: 4.86, 64: 4.61, 80: 5.03, whole documents:
        // 5.36 ms at config 3; config 5, k = 128: flat within 1 % from 32 to 128), 8 entries at k = 32
        // (0.224 against 0.234-0.238 ms for 16-40 and 0.316 ms for whole documents at config 2)
        const int eseg = c->eseg_override >= 0 ? c->eseg_override : (c->lpn >= 16 ? 64 : (c->lpn == 8 ? 8 : 16));
        CHK(ensure_eitems(c, eseg));
        const bool items = eseg > 0 && c->n_eitems > 0;
        const int grid = grid_for(c, items ? c->n_eitems : c->n, 256 / c->lpn);
        const int *order = nullptr;
        if (!items) CHK(ensure_roworder(c, &order));
        CHK(dispatch_shape(c, [&](auto S) {
            Scope s(c, "k_e_step");
            auto go = [&](auto TN) {      // TN: the denormal-norm rescue is compiled in only for thresholds below TINY_THRESH
                hipLaunchKernelGGL((plsa::k_e_step_rows<decltype(S), decltype(TN)::value>), dim3(grid), dim3(256), 0, c->stream,
                                   c->indptr, c->col, (int)c->n, order, c->U[c->cu].as<float>(), c->Vt[c->cv].as<float>(),
                                   p_base(c), c->kp, thresh, items ? c->eitem_row.as<int>() : nullptr,
                                   items ? c->eitem_start.as<int>() : nullptr, eseg, c->n_eitems);
            };
            if (thresh < plsa::TINY_THRESH) go(std::true_type{}); else go(std::false_type{});
        }));
    } else {
        CHK(ensure_rowidx(c));
        const i64 tiles = (c->nnz + 63) / 64;
        const int grid = grid_for(c, tiles, 4);
        CHK(dispatch_shape(c, [&](auto S) {
            Scope s(c, "k_e_step");
            auto go = [&](auto TN) {
                hipLaunchKernelGGL((plsa::k_e_step<decltype(S), decltype(TN)::value>), dim3(grid), dim3(256), 0, c->stream,
                                   c->rowidx.as<int>(), c->col, c->nnz, c->U[c->cu].as<float>(),
                                   c->Vt[c->cv].as<float>(), p_base(c), c->kp, thresh);
            };
            if (thresh < plsa::TINY_THRESH) go(std::true_type{}); else go(std::false_type{});
        }));
    }
    CHK(launch_check(c, "k_e_step"));
    c->p_valid = true;
    return 0;
}

// document-owned pass: writes U[1-cu]; optional LL partials of the current factors
int run_row_pass(plsa_ctx *c, bool from_p, bool want_ll, const float *d_sw, float thresh,
                 float *d_norm_pdz, int *ll_blocks) {
    CHK(ensure_ritems(c));
    const bool items = c->use_ritems && c->n_ritems > 0;
    int grid = grid_for(c, items ? c->n_ritems : c->n, 256 / c->row_lpn);
    const int *order = nullptr;
    // PLSA_ROW_XCD (experiment): one trip, grid a multiple of 8, XCD x takes the x-th eighth of the visiting list, which is
    // ordered range by range (range = the documents of one eighth), longest document first inside a range
    const int gpb_row = 256 / c->row_lpn;
    const bool xcd_rows = !items && !from_p && c->sort_rows && row_xcd_range(c) > 0;
    if (xcd_rows) grid = (int)(((c->n + gpb_row - 1) / gpb_row + 7) / 8 * 8);
    if (!items) CHK(ensure_roworder(c, &order));
    if (items) CHK(ensure(c, c->rpartial, sizeof(float) * (size_t)c->n_ritems * c->kp));
    const int *ri_row = items ? c->ritem_row.as<int>() : nullptr;
    const int *ri_start = items ? c->ritem_start.as<int>() : nullptr;
    float *rpart = items ? c->rpartial.as<float>() : nullptr;
    const int rseg = c->rseg;
    const i64 n_ritems = c->n_ritems;
    if (want_ll) CHK(ensure(c, c->ll_partials, sizeof(double) * (size_t)grid));
    CHK(dispatch_shape_row(c, [&](auto S) {
        using Sh = decltype(S);
        const int *ip = c->indptr, *cl = c->col;
        const float *vl = c->val, *U = c->U[c->cu].as<float>(), *Vt = c->Vt[c->cv].as<float>();
        const float *P = p_base(c);
        float *Un = c->U[out_u(c)].as<float>();
        double *llp = c->ll_partials.as<double>();
        const int n = (int)c->n, kp = c->kp;
        auto go = [&](auto FP, auto LL, auto TN, const char *name) {
            Scope s(c, name);
            hipLaunchKernelGGL((plsa::k_row_pass<Sh, decltype(FP)::value, decltype(LL)::value, decltype(TN)::value>),
                               dim3(grid), dim3(256), 0, c->ls, ip, cl, vl, n, order, U, Vt, P, Un,
                               d_sw, d_norm_pdz, kp, thresh, llp, ri_row, ri_start, rseg, n_ritems, rpart, xcd_rows ? 1 : 0);
        };
        using T = std::true_type;
        using F = std::false_type;
        const bool tiny = thresh < plsa::TINY_THRESH;   // denormal-norm rescue: compiled in for these thresholds only
        if (from_p) go(T{}, F{}, F{}, "k_row_pass<P>");
        else if (want_ll) { if (tiny) go(F{}, T{}, T{}, "k_row_pass<fused,LL>"); else go(F{}, T{}, F{}, "k_row_pass<fused,LL>"); }
        else { if (tiny) go(F{}, F{}, T{}, "k_row_pass<fused>"); else go(F{}, F{}, F{}, "k_row_pass<fused>"); }
        if (items) {
            Scope s(c, "k_row_reduce");
            hipLaunchKernelGGL((plsa::k_row_reduce<Sh>), dim3(grid_for(c, c->n, 256 / Sh::LPN)), dim3(256), 0, c->ls,
                               c->ritem_first.as<int>(), n, rpart, Un, d_norm_pdz, kp);
        }
    }));
    CHK(launch_check(c, "k_row_pass"));
    if (ll_blocks) *ll_blocks = grid;
    return 0;
}

// chunk boundaries of the column pass from the current fractions
void balance_set_lo(plsa_ctx *c, int n_chunks) {
    c->bal_lo[0] = 0;
    for (int x = 1; x < 8; ++x) {
        int v = (int)(c->bal_frac[x] * n_chunks + 0.5);
        c->bal_lo[x] = std::min(n_chunks, std::max(c->bal_lo[x - 1], v));
    }
    c->bal_lo[8] = n_chunks;
    c->bal_chunks = n_chunks;
}

// Grid of the column pass: ONE chunk per workgroup (the dispatcher then walks each XCD's stretch strictly in list
// order; with a capped grid a workgroup's later chunks lay a whole grid ahead of the window its XCD was working on:
// 32 k / 64 k / 128 k workgroups at config 3 -> 1.86 / 1.83 / 1.79 ms).  With the XCD split every XCD gets grid / 8
// workgroups, so the grid is eight times the longest stretch; the others' surplus workgroups exit at once.
int col_grid(plsa_ctx *c, int n_chunks, bool split) {
    i64 g = n_chunks;
    if (split) {
        int longest = 1;
        for (int x = 0; x < 8; ++x) longest = std::max(longest, c->bal_lo[x + 1] - c->bal_lo[x]);
        g = 8 * (i64)longest;
    }
    return (int)std::max<i64>(1, std::min<i64>(g, (i64)1 << 22));
}

// Measured XCD boundaries of the column pass (see k_col_pass).  `launch(timed)` enqueues one column pass on c->ls.
// Equal stretches first (or the fractions measured for the previous structure on this context: a bootstrap
// resample of the same corpus has the same profile and gets one measurement + one correction), then one untimed and up
// to eight timed launches: every workgroup records its end time, an XCD's time is its last workgroup's, and every stretch is
// resized by (mean time / own time), damped -- until the eight finish within 2 % of each other.  Results never
// depend on the boundaries (partials are per item, norm_pwz rows per chunk), only the speed does.
template <class Launch>
int ensure_balance(plsa_ctx *c, int n_chunks, bool split, Launch &&launch) {
    if (c->bal_valid && c->bal_chunks == n_chunks) return 0;
    CHK(ensure(c, c->xcd_lo, sizeof(int) * 16));
    const bool warm = c->bal_have_frac;
    if (!warm) for (int x = 0; x <= 8; ++x) c->bal_frac[x] = x / 8.0;
    balance_set_lo(c, n_chunks);
    HIPCHK(c, hipMemcpyAsync(c->xcd_lo.p, c->bal_lo, sizeof(int) * 9, hipMemcpyHostToDevice, c->ls));
    c->bal_launches = 0;
    // auto: corpora from ~1e8 cells per iteration (config 2: 4279 -> 4540 iterations/s; the tuning launches of a
    // 20NG-sized corpus would cost a bootstrap member more than they return)
    // PLSA_SMALL_GRID caps the launch below col_grid(): workgroup b would no longer be chunk-stretch b's only visitor
    // and the start stamp would land in a slot read as an end time -- that experiment knob runs on equal stretches
    const bool small_grid_active = c->small_grid > 0 && (double)c->nnz * c->kp < c->overlap_full_limit;
    const bool tune = split && n_chunks >= 64 && !small_grid_active &&
                      (c->balance > 0 || (c->balance < 0 && (double)c->nnz * c->kp >= 1e8));
    if (tune) {
        const size_t cap = (size_t)8 * (size_t)n_chunks + 16;      // any boundaries: at most 8 x n_chunks workgroups
        CHK(ensure(c, c->t_end, sizeof(unsigned long long) * cap));
        std::vector<unsigned long long> te;
        const int max_launches = warm ? 1 : 8;     // a resample of the same corpus: one measurement, one correction
        if (!warm) CHK(launch(false));             // first structure on this context: clocks and caches warm before timing
        double best_spread = 1e300, best_frac[9];
        for (int x = 0; x <= 8; ++x) best_frac[x] = c->bal_frac[x];
        for (int it = 0; it < max_launches; ++it) {
            const int grid = col_grid(c, n_chunks, split);
            te.assign((size_t)grid + 1, 0);
            HIPCHK(c, hipMemsetAsync(c->t_end.p, 0, sizeof(unsigned long long) * ((size_t)grid + 1), c->ls));
            CHK(launch(true));
            HIPCHK(c, hipMemcpyAsync(te.data(), c->t_end.p, sizeof(unsigned long long) * ((size_t)grid + 1),
                                     hipMemcpyDeviceToHost, c->ls));
            HIPCHK(c, hipStreamSynchronize(c->ls));
            c->bal_launches++;
            unsigned long long t0 = ~0ull, last[8] = {0, 0, 0, 0, 0, 0, 0, 0};
            for (int b = 0; b <= grid; ++b) if (te[b]) t0 = std::min(t0, te[b]);
            for (int b = 0; b < grid; ++b) last[b & 7] = std::max(last[b & 7], te[b]);
            double T[8], mean = 0.0, lo_t = 1e300, hi_t = 0.0;
            for (int x = 0; x < 8; ++x) {
                T[x] = std::max(1.0, (double)(last[x] - t0) / 100.0);       // us (100 MHz wall clock)
                c->bal_end_us[x] = T[x];
                mean += T[x] / 8.0; lo_t = std::min(lo_t, T[x]); hi_t = std::max(hi_t, T[x]);
            }
            const double spread = (hi_t - lo_t) / mean;
            if (spread < best_spread) {            // remember the best boundaries seen (a noisy launch must not have the last word)
                best_spread = spread;
                for (int x = 0; x <= 8; ++x) best_frac[x] = c->bal_frac[x];
            }
            if (spread <= 0.02) break;
            if (it == max_launches - 1 && !warm) {   // out of launches: keep the best measured boundaries
                for (int x = 0; x <= 8; ++x) c->bal_frac[x] = best_frac[x];
                balance_set_lo(c, n_chunks);
                HIPCHK(c, hipMemcpyAsync(c->xcd_lo.p, c->bal_lo, sizeof(int) * 9, hipMemcpyHostToDevice, c->ls));
                break;
            }
            double size[8], tot = 0.0;
            for (int x = 0; x < 8; ++x) {
                size[x] = std::max(1e-6, (c->bal_frac[x + 1] - c->bal_frac[x]) * (1.0 + 0.8 * (mean / T[x] - 1.0)));
                tot += size[x];
            }
            double acc = 0.0;
            for (int x = 0; x < 8; ++x) { acc += size[x]; c->bal_frac[x + 1] = acc / tot; }
            c->bal_frac[8] = 1.0;
            balance_set_lo(c, n_chunks);
            HIPCHK(c, hipMemcpyAsync(c->xcd_lo.p, c->bal_lo, sizeof(int) * 9, hipMemcpyHostToDevice, c->ls));
        }
        c->bal_have_frac = true;
    }
    c->bal_valid = true;
    return 0;
}

// vocabulary-owned pass (no atomics): partial k-vectors per column item (+ per-chunk sums of them, from
// which the column tail gets norm_pwz), then per-column sums -> Vacc
// parts: 1 = the column pass itself, 2 = the un-normalised per-column sums of its partials (k_col_reduce),
//        3 = both
int run_col_pass(plsa_ctx *c, bool from_p, const float *d_sw, float thresh, int parts = 3) {
    CHK(ensure_csc(c));
    CHK(ensure(c, c->partial, sizeof(float) * (size_t)std::max<i64>(c->n_items, 1) * c->kp));
    int rc = 0;
    CHK(dispatch_shape_gather(c, table_is_wide(c, c->n), [&](auto S) {     // the pass gathers P(z|d) rows: n of them
        using Sh = decltype(S);
        constexpr int LPN = Sh::LPN, GPB = 256 / LPN;
        const i64 n_visit = c->n_items;
        const int n_chunks = (int)((n_visit + GPB - 1) / GPB);
        const int grid2 = grid_for(c, c->m, GPB);
        // a P(z|d) table that fits every XCD's L2 (20NG shape: 1.5 MB) has no band to keep local: plain grid-stride
        // over the list, balanced by the dispatcher (config 1: 8990 -> 9490 iterations/s)
        const bool u_fits_l2 = (double)c->n * c->kp * 4.0 <= 2.0 * 1024 * 1024;
        const int xcd_split = (c->xcd_split && n_chunks >= 64 && !u_fits_l2) ? 1 : 0;
        if (parts & 1) {
            rc = ensure(c, c->colsum_rows, sizeof(double) * (size_t)std::max(n_chunks, 1) * c->kp);
            if (rc) return;
            const size_t smem = sizeof(double) * (size_t)GPB * c->kp;
            auto launch = [&](bool timed) -> int {
                int grid = col_grid(c, n_chunks, xcd_split != 0);
                // PLSA_SMALL_GRID (experiment knob): cap the pass of a small corpus at that many workgroups per CU
                if (c->small_grid > 0 && (double)c->nnz * c->kp < c->overlap_full_limit)
                    grid = std::min(grid, c->small_grid * c->prop.multiProcessorCount);
                Scope s(c, from_p ? "k_col_pass<P>" : "k_col_pass<fused>");
                const int4 *rec = c->item_rec.as<int4>();
                const int *lo = c->xcd_lo.as<int>(), *cr = c->csc_row.as<int>(), *cp = c->csc_pos.as<int>();
                const float *cvl = c->csc_val.as<float>(), *U = c->U[c->cu].as<float>(), *Vt = c->Vt[c->cv].as<float>();
                float *part = c->partial.as<float>();
                double *sums = c->colsum_rows.as<double>();
                unsigned long long *te = c->t_end.as<unsigned long long>();
                const int kp = c->kp;
                auto go = [&](auto FP, auto TM, auto TN) {
                    hipLaunchKernelGGL((plsa::k_col_pass<Sh, decltype(FP)::value, decltype(TM)::value, decltype(TN)::value>),
                                       dim3(grid), dim3(256), smem, c->ls, rec, n_visit, lo, cr, cvl, cp, U, Vt, p_base(c), d_sw,
                                       part, kp, thresh, xcd_split, sums, te);
                };
                using T = std::true_type;
                using F = std::false_type;
                const bool tiny = !from_p && thresh < plsa::TINY_THRESH;
                if (from_p) { if (timed) go(T{}, T{}, F{}); else go(T{}, F{}, F{}); }
                else if (tiny) { if (timed) go(F{}, T{}, T{}); else go(F{}, F{}, T{}); }
                else { if (timed) go(F{}, T{}, F{}); else go(F{}, F{}, F{}); }
                return launch_check(c, "k_col_pass");
            };
            rc = ensure_balance(c, n_chunks, xcd_split != 0, launch);
            if (rc) return;
            rc = launch(false);
            if (rc) return;
            c->colsum_rows_used = n_chunks;
        }
        if (parts & 2) {
            // heavy columns (one block each) and the rest share one launch
            Scope s(c, "k_col_reduce");
            hipLaunchKernelGGL((plsa::k_col_reduce<Sh>), dim3(grid2 + c->n_heavy), dim3(256),
                               (256 / LPN) * c->kp * sizeof(float), c->ls,
                               c->item_first.as<int>(), (int)c->m, c->heavy_items, c->heavy_cols.as<int>(), c->n_heavy,
                               c->partial.as<float>(), c->Vacc.as<float>(), c->kp);
        }
    }));
    if (rc) return rc;
    CHK(launch_check(c, "k_col_pass"));
    return 0;
}

// Vacc -> normalised topics in Vt[1-cv]  (plsa.py:196-199)
int run_v_normalise(plsa_ctx *c) {
    if (c->sharded && c->comm) {
        // doc-sharded fit: the un-normalised P(w|z) sums of the local rows become the global sums (the
        // `.sum(axis=0)` over tiles of distributed_plsa.py:116-128 / block_parallel_plsa.py:182-185) --
        // in place, on the stream this chain runs on (underneath the document pass when overlapped)
        Scope s(c, "rccl_allreduce_accumulator");
        NCCLCHK(c, ncclAllReduce(c->Vacc.p, c->Vacc.p, (size_t)c->m * c->kp, ncclFloat, ncclSum, c->comm, c->ls));
    }
    const int nb = (int)std::min<i64>(plsa::NORM_BLOCKS, std::max<i64>(1, c->m));
    CHK(ensure(c, c->colsum_partials, sizeof(double) * (size_t)nb * c->kp));
    {
        Scope s(c, "k_colsum_partial");
        hipLaunchKernelGGL(plsa::k_colsum_partial, dim3(nb), dim3(256), 256 * sizeof(double), c->ls,
                           c->Vacc.as<float>(), (int)c->m, c->kp, c->colsum_partials.as<double>());
    }
    CHK(ensure(c, c->norm_pwz, sizeof(float) * (size_t)c->kp));
    {
        Scope s(c, "k_colsum_final");
        hipLaunchKernelGGL(plsa::k_colsum_final, dim3(1), dim3(256), 0, c->ls,
                           c->colsum_partials.as<double>(), nb, c->kp, c->norm_pwz.as<float>());
    }
    {
        Scope s(c, "k_v_normalise");
        const i64 total4 = c->m * c->kp / 4;
        hipLaunchKernelGGL(plsa::k_v_normalise, dim3(grid_for(c, total4, 256)), dim3(256),
                           c->kp * sizeof(float), c->ls, c->Vacc.as<float>(),
                           c->Vt[out_v(c)].as<float>(), (int)c->m, c->kp, c->norm_pwz.as<float>());
    }
    CHK(launch_check(c, "k_v_normalise"));
    return 0;
}

// Everything behind the column pass: norm_pwz from the pass' own per-block sums (two small launches),
// then per-column sums of the item partials and the division in ONE sweep (k_col_reduce_norm) ->
// Vt[1-cv].  The doc-sharded fit needs the un-normalised accumulator for its all-reduce and keeps the
// four-kernel form (k_col_reduce, k_colsum_partial, k_colsum_final, k_v_normalise).
int run_col_tail(plsa_ctx *c) {
    if (c->sharded) {
        CHK(run_col_pass(c, false, nullptr, 0.f, 2));
        return run_v_normalise(c);
    }
    const int rows = c->colsum_rows_used;
    if (rows <= 0) return fail(c, "internal: column tail without a column pass");
    CHK(ensure(c, c->norm_pwz, sizeof(float) * (size_t)c->kp));
    const double *rows_in = c->colsum_rows.as<double>();
    int n_rows = rows;
    if (rows > 2048) {               // many chunks (large corpora): two stages
        const int nb = std::max(64, std::min(1024, rows / 64));
        CHK(ensure(c, c->colsum_rows2, sizeof(double) * (size_t)nb * c->kp));
        Scope s(c, "k_norm_reduce");
        hipLaunchKernelGGL(plsa::k_norm_reduce, dim3(nb), dim3(256), 0, c->ls, rows_in, rows, c->kp,
                           c->colsum_rows2.as<double>());
        rows_in = c->colsum_rows2.as<double>();
        n_rows = nb;
    }
    {
        Scope s(c, "k_colsum_final");
        hipLaunchKernelGGL(plsa::k_colsum_final, dim3(1), dim3(256), 0, c->ls, rows_in, n_rows, c->kp,
                           c->norm_pwz.as<float>());
    }
    CHK(dispatch_shape(c, [&](auto S) {
        using Sh = decltype(S);
        constexpr int GPB = 256 / Sh::LPN;
        const int grid2 = grid_for(c, c->m, GPB);
        Scope s(c, "k_col_reduce_norm");
        hipLaunchKernelGGL((plsa::k_col_reduce_norm<Sh>), dim3(grid2 + c->n_heavy), dim3(256),
                           sizeof(float) * (size_t)(GPB + 1) * c->kp, c->ls, c->item_first.as<int>(), (int)c->m,
                           c->heavy_items, c->heavy_cols.as<int>(), c->n_heavy, c->partial.as<float>(),
                           c->norm_pwz.as<float>(), c->Vt[out_v(c)].as<float>(), c->kp);
    }));
    return launch_check(c, "k_col_reduce_norm");
}

int finish_ll(plsa_ctx *c, int blocks, double *out) {
    CHK(ensure(c, c->ll_out, sizeof(double)));
    {
        Scope s(c, "k_ll_final");
        hipLaunchKernelGGL(plsa::k_ll_final, dim3(1), dim3(256), 0, c->stream,
                           c->ll_partials.as<double>(), blocks, c->ll_out.as<double>());
    }
    CHK(launch_check(c, "k_ll_final"));
    if (c->sharded && c->comm)      // log-likelihood of all shards: one scalar all-reduce per test
        NCCLCHK(c, ncclAllReduce(c->ll_out.p, c->ll_out.p, 1, ncclDouble, ncclSum, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(c->h_ll, c->ll_out.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    if (!out) return hipEventRecord(c->ev_ll, c->stream) == hipSuccess ? 0 : fail(c, "event record failed");   // collected by wait_ll
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = *c->h_ll;
    return 0;
}

// second half of finish_ll(c, blocks, nullptr): the host waits for the likelihood only, not for the work enqueued behind it
int wait_ll(plsa_ctx *c, double *out) {
    HIPCHK(c, hipEventSynchronize(c->ev_ll));
    *out = *c->h_ll;
    return 0;
}

int run_loglik(plsa_ctx *c, const float *d_sw, double *out) {
    if (c->ref_ll) return run_ref_loglik(c, d_sw, out);
    const int grid = grid_for(c, c->n, 256 / c->lpn);
    const int *order = nullptr;
    CHK(ensure_roworder(c, &order));
    CHK(ensure(c, c->ll_partials, sizeof(double) * (size_t)grid));
    CHK(dispatch_shape(c, [&](auto S) {
        Scope s(c, "k_loglik");
        hipLaunchKernelGGL((plsa::k_loglik<decltype(S)>), dim3(grid), dim3(256), 0, c->stream, c->indptr,
                           c->col, c->val, (int)c->n, order, c->U[c->cu].as<float>(),
                           c->Vt[c->cv].as<float>(), d_sw, c->kp, c->ll_partials.as<double>());
    }));
    CHK(launch_check(c, "k_loglik"));
    return finish_ll(c, grid, out);
}

// one M-step from the materialised P: U[1-cu], and (update_v) Vt[1-cv]; swaps the buffers in
int run_m_step_from_p(plsa_ctx *c, const float *d_sw, bool update_v, float *d_norm_pdz) {
    if (!c->p_valid) return fail(c, "plsa_m_step: no P(z|w,d) on the device (run plsa_e_step or plsa_set_p)");
    if (c->ref_sums) return run_ref_m_step(c, d_sw, update_v, d_norm_pdz);
    CHK(run_row_pass(c, true, false, nullptr, 0.f, d_norm_pdz, nullptr));
    if (update_v) {
        CHK(run_col_pass(c, true, d_sw, 0.f, 1));
        CHK(run_col_tail(c));
    }
    c->cu ^= 1;
    if (update_v) c->cv ^= 1;
    return 0;
}

// ---------------------------------------------------------------------------------------------
// reference arithmetic (plsa_ref_kernels.hpp): the reference's statements with the reference's roundings
// ---------------------------------------------------------------------------------------------
// lanes per document / column (G) and topics per lane (NZ) of the reference-arithmetic passes: z = lane + G t
template <class Fn>
int dispatch_ref_group(plsa_ctx *c, Fn &&fn) {
    using std::integral_constant;
    const int kp = c->kp;
    if (kp <= 8) fn(integral_constant<int, 8>{}, integral_constant<int, 1>{});
    else if (kp <= 16) fn(integral_constant<int, 16>{}, integral_constant<int, 1>{});
    else if (kp <= 32) fn(integral_constant<int, 32>{}, integral_constant<int, 1>{});
    else if (kp <= 64) fn(integral_constant<int, 64>{}, integral_constant<int, 1>{});
    else if (kp <= 128) fn(integral_constant<int, 64>{}, integral_constant<int, 2>{});
    else if (kp <= 256) fn(integral_constant<int, 64>{}, integral_constant<int, 4>{});
    else if (kp <= 512) fn(integral_constant<int, 64>{}, integral_constant<int, 8>{});
    else if (kp <= 1024) fn(integral_constant<int, 64>{}, integral_constant<int, 16>{});
    else return fail(c, "unsupported topic count k=%d (max 1024)", c->k);
    return 0;
}

// plsa.py:91-105 with one float32 norm per entry, topics in order (P allocated by run_e_step)
int run_ref_e_step(plsa_ctx *c, float thresh) {
    CHK(ensure_rowidx(c));
    static const bool tiled = [] { const char *e = getenv("PLSA_REF_E_TILED"); return !e || atoi(e) != 0; }();
    if (tiled && c->nnz > 0) {
        Scope s(c, "k_ref_e_step");
        const int kp = c->kp;
        // (PLSA_REF_FUSE_SUMS=0: no tile sums, the chain's k_ref_pair_sums reads P itself)
        const char *fe = getenv("PLSA_REF_FUSE_SUMS");
        const bool fuse = (!fe || atoi(fe) != 0) && ref_pairs_now(c) && !c->ref_e_no_sums;
        int rc_alloc = 0;
        auto go = [&](auto NZ) {
            constexpr int nz = decltype(NZ)::value;
            const i64 tiles = (c->nnz + 64 / nz - 1) / (64 / nz);
            if (fuse && (rc_alloc = ensure(c, c->ref_tsum, sizeof(float) * (size_t)tiles * kp)) == 0) {
                hipLaunchKernelGGL((plsa::ref::k_ref_e_step_tiled<nz, true>), dim3(grid_for(c, tiles, 2)), dim3(128),
                                   sizeof(float) * 2 * (64 / nz) * (size_t)(kp + 1), c->stream, c->rowidx.as<int>(), c->col, c->nnz,
                                   c->U[c->cu].as<float>(), c->Vt[c->cv].as<float>(), p_base(c), kp, thresh, c->val, c->ref_e_sw,
                                   c->ref_tsum.as<float>());
                c->ref_tsum_valid = true; c->ref_tsum_sw = c->ref_e_sw; c->ref_tsum_tj = 64 / nz;
            } else if (!rc_alloc) {
                hipLaunchKernelGGL((plsa::ref::k_ref_e_step_tiled<nz, false>), dim3(grid_for(c, tiles, 2)), dim3(128),
                                   sizeof(float) * 2 * (64 / nz) * (size_t)(kp + 1), c->stream, c->rowidx.as<int>(), c->col, c->nnz,
                                   c->U[c->cu].as<float>(), c->Vt[c->cv].as<float>(), p_base(c), kp, thresh, nullptr, nullptr, nullptr);
            }
        };
        using std::integral_constant;
        if (kp <= 64) go(integral_constant<int, 1>{});
        else if (kp <= 128) go(integral_constant<int, 2>{});
        else if (kp <= 256) go(integral_constant<int, 4>{});
        else if (kp <= 512) go(integral_constant<int, 8>{});
        else go(integral_constant<int, 16>{});
        if (rc_alloc) return rc_alloc;
    } else {
        Scope s(c, "k_ref_e_step");
        hipLaunchKernelGGL(plsa::ref::k_ref_e_step, dim3(grid_for(c, c->nnz, 256)), dim3(256), 0, c->stream,
                           c->rowidx.as<int>(), c->col, c->nnz, c->U[c->cu].as<float>(), c->Vt[c->cv].as<float>(),
                           p_base(c), c->kp, thresh);
    }
    CHK(launch_check(c, "k_ref_e_step"));
    c->p_valid = true;
    return 0;
}

// norm_pwz[z] = the reference's ONE float32 running sum over all non-zeros (plsa.py:193) on c->ls: from per-chunk parity pairs and a
// walk (k_ref_pair_*), or by the serial chain (k_ref_norm_chain); same bits either way.
// the reference arithmetic's long columns (PLSA_REF_HEAVY_MIN entries or more, default 2048; 0: none): list [count, columns...]
int ensure_ref_heavy(plsa_ctx *c) {
    if (c->ref_heavy_valid) return 0;
    const char *e = getenv("PLSA_REF_HEAVY_MIN");
    c->ref_heavy_min = e ? atoi(e) : 2048;
    c->n_ref_heavy = 0;
    if (c->ref_heavy_min > 0 && c->nnz > 0) {
        CHK(ensure(c, c->ref_heavy, sizeof(int) * (size_t)(c->m + 1)));
        HIPCHK(c, hipMemsetAsync(c->ref_heavy.p, 0, sizeof(int), c->stream));
        hipLaunchKernelGGL(plsa::ref::k_ref_heavy_cols, dim3((unsigned)((c->m + 255) / 256)), dim3(256), 0, c->stream,
                           c->colptr.as<int>(), (int)c->m, c->ref_heavy_min, c->ref_heavy.as<int>() + 1, c->ref_heavy.as<int>());
        CHK(launch_check(c, "k_ref_heavy_cols"));
        int n = 0;
        HIPCHK(c, hipMemcpyAsync(&n, c->ref_heavy.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream));
        c->n_ref_heavy = n;
    }
    c->ref_heavy_valid = true;
    return 0;
}

// One chain of float32 additions over the non-zeros in order, per "topic" z < kp, WITHOUT the chain (plsa_ref_kernels.hpp: chunk
// sums -> prefix -> (parity -> increment) pairs -> one checking walk per 64 topics), on c->ls.  kind / P / kp: PAIR_PLAIN or
// PAIR_WEIGHTED over P(z|w,d) (norm_pwz, plsa.py:193), PAIR_NEG_TERMS over the likelihood terms with kp = 1 (plsa.py:322).
int run_ref_pair_chain(plsa_ctx *c, int kind, const float *P, int kp, const float *d_sw, float *out, unsigned long long *stats) {
    const int *ri = c->rowidx.as<int>();
    const bool ll = kind == plsa::ref::PAIR_NEG_TERMS;              // (timing names: the likelihood's launches apart from norm_pwz's)
    // two levels (default): chunks of 256 addends, walked in groups of PAIR_R; PLSA_REF_LEVELS=1: chunks only, longer on large corpora
    bool two = true;
    if (const char *e = getenv("PLSA_REF_LEVELS")) two = atoi(e) != 1;
    int L = !two && c->nnz >= plsa::ref::PAIR_L_LARGE_FROM ? plsa::ref::PAIR_L_LARGE : plsa::ref::PAIR_L_SMALL;
    if (const char *e = getenv("PLSA_REF_CHUNK")) { const int v = atoi(e); if (v >= 64 && v <= 4096 && v % 64 == 0) L = v; }
    const i64 n_chunks = (c->nnz + L - 1) / L;
    const i64 n_groups = (n_chunks + plsa::ref::PAIR_R - 1) / plsa::ref::PAIR_R;
    if (two) {
        CHK(ensure(c, c->ref_pairs2, sizeof(uint4) * (size_t)n_groups * kp));
        CHK(ensure(c, c->ref_exps2, sizeof(unsigned) * (size_t)n_groups * kp));
    }
    const i64 n_super = (n_chunks + plsa::ref::PAIR_SC - 1) / plsa::ref::PAIR_SC, n_pad = n_super * plsa::ref::PAIR_SC;
    CHK(ensure(c, c->ref_csum, sizeof(double) * (size_t)n_pad * kp));
    CHK(ensure(c, c->ref_pairs, sizeof(uint4) * (size_t)n_chunks * kp));
    CHK(ensure(c, c->ref_exps, sizeof(unsigned) * (size_t)n_chunks * kp));
    const int grid = grid_for(c, n_super, 4);
    double *csum = c->ref_csum.as<double>();
    uint4 *prs = c->ref_pairs.as<uint4>();
    unsigned *exps = c->ref_exps.as<unsigned>();
    auto by_nz = [&](auto &&go) {
        using std::integral_constant;
        if (kp <= 64) go(integral_constant<int, 1>{});
        else if (kp <= 128) go(integral_constant<int, 2>{});
        else if (kp <= 256) go(integral_constant<int, 4>{});
        else if (kp <= 512) go(integral_constant<int, 8>{});
        else go(integral_constant<int, 16>{});
    };
    by_nz([&](auto NZ) {
        constexpr int nz = decltype(NZ)::value;
        auto go = [&](auto KIND) {
            constexpr int kd = decltype(KIND)::value;
            if (!ll && c->ref_tsum_valid && c->ref_tsum_sw == d_sw && P == p_base(c) && c->p_valid && L % c->ref_tsum_tj == 0) {
                // the E-step that wrote this P left the sums of its tiles: no second pass over P
                Scope s(c, "k_ref_pair_sums");
                const i64 n_tiles = (c->nnz + c->ref_tsum_tj - 1) / c->ref_tsum_tj;
                hipLaunchKernelGGL(plsa::ref::k_ref_pair_sums_from_tiles, dim3(grid_for(c, n_chunks * kp, 256)), dim3(256), 0, c->ls,
                                   c->ref_tsum.as<float>(), kp, L / c->ref_tsum_tj, n_tiles, n_chunks, n_pad, csum);
            } else {
                Scope s(c, ll ? "k_ref_ll_pair_sums" : "k_ref_pair_sums");
                hipLaunchKernelGGL((plsa::ref::k_ref_pair_sums<nz, kd>), dim3(grid), dim3(256), 0, c->ls, ri, c->val, c->nnz, P,
                                   d_sw, kp, L, n_chunks, n_pad, csum);
            }
            {
                Scope s(c, ll ? "k_ref_ll_pair_prefix" : "k_ref_pair_prefix");
                hipLaunchKernelGGL(plsa::ref::k_ref_pair_prefix, dim3(kp), dim3(256), 0, c->ls, csum, n_chunks, n_pad);
            }
            {
                Scope s(c, ll ? "k_ref_ll_pair_build" : "k_ref_pair_build");
                hipLaunchKernelGGL((plsa::ref::k_ref_pair_build<nz, kd>), dim3(grid), dim3(256), 0, c->ls, ri, c->val, c->nnz, P,
                                   d_sw, kp, L, n_chunks, n_pad, csum, prs, exps);
            }
            if (two) {
                {
                    Scope s(c, ll ? "k_ref_ll_pair_compose" : "k_ref_pair_compose");
                    hipLaunchKernelGGL(plsa::ref::k_ref_pair_compose, dim3(grid_for(c, n_groups * kp, 256)), dim3(256), 0, c->ls, prs, exps,
                                       kp, n_chunks, n_groups, c->ref_pairs2.as<uint4>(), c->ref_exps2.as<unsigned>());
                }
                Scope s(c, ll ? "k_ref_ll_pair_walk" : "k_ref_pair_walk");
                hipLaunchKernelGGL((plsa::ref::k_ref_pair_walk<kd, true>), dim3((kp + 63) / 64), dim3(plsa::ref::WALK_THREADS), 0, c->ls, ri,
                                   c->val, c->nnz, P, d_sw, kp, L, n_groups, c->ref_pairs2.as<uint4>(), c->ref_exps2.as<unsigned>(), out,
                                   stats, n_chunks, prs, exps);
            } else {
                Scope s(c, ll ? "k_ref_ll_pair_walk" : "k_ref_pair_walk");
                hipLaunchKernelGGL((plsa::ref::k_ref_pair_walk<kd, false>), dim3((kp + 63) / 64), dim3(plsa::ref::WALK_THREADS), 0, c->ls, ri,
                                   c->val, c->nnz, P, d_sw, kp, L, n_chunks, prs, exps, out, stats, n_chunks, prs, exps);
            }
        };
        using std::integral_constant;
        if (kind == plsa::ref::PAIR_NEG_TERMS) {
            if constexpr (nz == 1) go(integral_constant<int, plsa::ref::PAIR_NEG_TERMS>{});      // (kp = 1)
        } else if (kind == plsa::ref::PAIR_WEIGHTED) go(integral_constant<int, plsa::ref::PAIR_WEIGHTED>{});
        else go(integral_constant<int, plsa::ref::PAIR_PLAIN>{});
    });
    return launch_check(c, "k_ref_pair_walk");
}

// does this context evaluate its long chains from parity pairs right now? (PLSA_REF_CHAIN; auto: from 4096 non-zeros, until a
// finished walk reported more than a quarter of its chunks on the slow way)
bool ref_pairs_now(const plsa_ctx *c) {
    return c->ref_chain_mode == 1 || (c->ref_chain_mode == 0 && !c->ref_pairs_off && c->nnz >= 4096);
}

int run_ref_norm_pwz(plsa_ctx *c, const float *d_sw) {
    const int kp = c->kp;
    const int *ri = c->rowidx.as<int>();
    float *out = c->norm_pwz.as<float>();
    if (c->nnz <= 0) { HIPCHK(c, hipMemsetAsync(out, 0, sizeof(float) * (size_t)kp, c->ls)); return 0; }
    // the last walk's count of slow chunks, if it has arrived (never waited for)
    if (c->ref_stats_pending && hipEventQuery(c->ev_ref_stats) == hipSuccess) {
        c->ref_stats_pending = false;
        const unsigned long long slow = c->h_ref_stats[0], chunks = c->h_ref_stats[1];
        c->ref_slow_total += slow; c->ref_chunks_total += chunks;
        if (c->ref_chain_mode == 0 && chunks > 0 && slow * 4 > chunks) c->ref_pairs_off = true;
    }
    const bool pairs = ref_pairs_now(c);
    auto by_nz = [&](auto &&go) {
        using std::integral_constant;
        if (kp <= 64) go(integral_constant<int, 1>{});
        else if (kp <= 128) go(integral_constant<int, 2>{});
        else if (kp <= 256) go(integral_constant<int, 4>{});
        else if (kp <= 512) go(integral_constant<int, 8>{});
        else go(integral_constant<int, 16>{});
    };
    if (!pairs) {
        Scope s(c, "k_ref_norm_chain");
        by_nz([&](auto NZ) {
            if (d_sw)
                hipLaunchKernelGGL((plsa::ref::k_ref_norm_chain<decltype(NZ)::value, true>), dim3(1), dim3(plsa::ref::CHAIN_THREADS), 0,
                                   c->ls, ri, c->val, c->nnz, p_base(c), d_sw, kp, out);
            else
                hipLaunchKernelGGL((plsa::ref::k_ref_norm_chain<decltype(NZ)::value, false>), dim3(1), dim3(plsa::ref::CHAIN_THREADS), 0,
                                   c->ls, ri, c->val, c->nnz, p_base(c), d_sw, kp, out);
        });
        return launch_check(c, "k_ref_norm_chain");
    }
    CHK(ensure(c, c->ref_stats, 32));
    if (!c->h_ref_stats) {
        HIPCHK(c, hipHostMalloc(reinterpret_cast<void **>(&c->h_ref_stats), 16, hipHostMallocDefault));
        HIPCHK(c, hipEventCreateWithFlags(&c->ev_ref_stats, hipEventDisableTiming));
    }
    unsigned long long *stats = c->ref_stats.as<unsigned long long>();
    HIPCHK(c, hipMemsetAsync(stats, 0, 16, c->ls));
    CHK(run_ref_pair_chain(c, d_sw ? plsa::ref::PAIR_WEIGHTED : plsa::ref::PAIR_PLAIN, p_base(c), kp, d_sw, out, stats));
    CHK(launch_check(c, "k_ref_pair_walk"));
    if (!c->ref_stats_pending) {       // (one read-back in flight at a time; a walk whose count is skipped is simply not counted)
        HIPCHK(c, hipMemcpyAsync(c->h_ref_stats, stats, 16, hipMemcpyDeviceToHost, c->ls));
        HIPCHK(c, hipEventRecord(c->ev_ref_stats, c->ls));
        c->ref_stats_pending = true;
    }
    return 0;
}

// plsa.py:172-204 / 277-310 / 795-816 from the materialised P: U[out], (update_v) Vt[out]; swaps the buffers in.
// The norm_pwz chain (one workgroup, nnz dependent additions per topic) runs on the second stream beside the document
// and column passes.
int run_ref_m_step(plsa_ctx *c, const float *d_sw, bool update_v, float *d_norm_pdz) {
    if (c->sharded) return fail(c, "the reference arithmetic has no doc-sharded form (norm_pwz is ONE chain over all non-zeros)");
    const int *order = nullptr;
    CHK(ensure_roworder(c, &order));
    int heavy_min = INT32_MAX;
    if (update_v) {
        CHK(ensure_csc(c));
        CHK(ensure_rowidx(c));
        CHK(ensure_ref_heavy(c));
        if (c->n_ref_heavy > 0) heavy_min = c->ref_heavy_min;
        CHK(ensure(c, c->norm_pwz, sizeof(float) * (size_t)c->kp));
        HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
        HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
        c->ls = c->stream2;
        CHK(run_ref_norm_pwz(c, d_sw));
        c->ls = c->stream;
        CHK(launch_check(c, "k_ref_norm_chain"));
        HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
    }
    // the tiled document pass pays from ~300 k documents on (config 3 whole: 22 -> 10 ms); below, the wave tile that holds the few
    // longest documents is the pass, and the group kernel walks a long document faster (PLSA_REF_ROW_TILED=1 / 0 pins either)
    const char *row_tiled_env = getenv("PLSA_REF_ROW_TILED");
    const bool row_tiled = row_tiled_env ? atoi(row_tiled_env) != 0 : c->n >= 300000;
    CHK(dispatch_ref_group(c, [&](auto G, auto NZ) {
        constexpr int g = decltype(G)::value, nz = decltype(NZ)::value;
        if (row_tiled) {
            Scope s(c, "k_ref_row_pass");
            constexpr int tj = 64 / nz;
            hipLaunchKernelGGL((plsa::ref::k_ref_row_pass_tiled<nz>), dim3(grid_for(c, (c->n + tj - 1) / tj, 2)), dim3(128),
                               sizeof(float) * 2 * tj * (size_t)(c->kp + 1), c->stream, c->indptr, c->val, (int)c->n, order,
                               p_base(c), c->U[out_u(c)].as<float>(), d_norm_pdz, c->kp);
        } else {
            Scope s(c, "k_ref_row_pass");
            hipLaunchKernelGGL((plsa::ref::k_ref_row_pass<g, nz>), dim3(grid_for(c, c->n, 256 / g)), dim3(256),
                               sizeof(float) * (size_t)(256 / g) * c->kp, c->stream, c->indptr, c->val, (int)c->n, order,
                               p_base(c), c->U[out_u(c)].as<float>(), d_norm_pdz, c->kp);
        }
        if (update_v) {
            Scope s(c, "k_ref_col_pass");
            hipLaunchKernelGGL((plsa::ref::k_ref_col_pass<g, nz>), dim3(grid_for(c, c->m, 256 / g)), dim3(256), 0, c->stream,
                               c->colptr.as<int>(), c->csc_row.as<int>(), c->csc_val.as<float>(), c->csc_pos.as<int>(),
                               (int)c->m, p_base(c), d_sw, c->Vacc.as<float>(), c->kp, heavy_min);
        }
    }));
    if (update_v && c->n_ref_heavy > 0) {
        // the long columns: one workgroup each, the norm_pwz chain's kernel over the column's entries (6.4 ns per entry where a
        // group's own walk costs ~160: the Zipf head was the pass -- 24.6 ms at the config-3 150 k sample, 157 ms at the whole)
        Scope s(c, "k_ref_col_heavy");
        const int kp = c->kp;
        auto go = [&](auto NZ) {
            constexpr int nz = decltype(NZ)::value;
            if (d_sw)
                hipLaunchKernelGGL((plsa::ref::k_ref_norm_chain<nz, true, true>), dim3(c->n_ref_heavy), dim3(plsa::ref::CHAIN_THREADS), 0,
                                   c->stream, c->csc_row.as<int>(), c->csc_val.as<float>(), (i64)0, p_base(c), d_sw, kp, c->Vacc.as<float>(),
                                   c->csc_pos.as<int>(), c->ref_heavy.as<int>() + 1, c->colptr.as<int>());
            else
                hipLaunchKernelGGL((plsa::ref::k_ref_norm_chain<nz, false, true>), dim3(c->n_ref_heavy), dim3(plsa::ref::CHAIN_THREADS), 0,
                                   c->stream, c->csc_row.as<int>(), c->csc_val.as<float>(), (i64)0, p_base(c), d_sw, kp, c->Vacc.as<float>(),
                                   c->csc_pos.as<int>(), c->ref_heavy.as<int>() + 1, c->colptr.as<int>());
        };
        using std::integral_constant;
        if (kp <= 64) go(integral_constant<int, 1>{});
        else if (kp <= 128) go(integral_constant<int, 2>{});
        else if (kp <= 256) go(integral_constant<int, 4>{});
        else if (kp <= 512) go(integral_constant<int, 8>{});
        else go(integral_constant<int, 16>{});
    }
    CHK(launch_check(c, "k_ref_row_pass / k_ref_col_pass"));
    if (update_v) {
        HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
        Scope s(c, "k_v_normalise");
        const i64 total4 = c->m * c->kp / 4;
        hipLaunchKernelGGL(plsa::k_v_normalise, dim3(grid_for(c, total4, 256)), dim3(256), c->kp * sizeof(float), c->stream,
                           c->Vacc.as<float>(), c->Vt[out_v(c)].as<float>(), (int)c->m, c->kp, c->norm_pwz.as<float>());
        CHK(launch_check(c, "k_v_normalise"));
    }
    c->cu ^= 1;
    if (update_v) c->cv ^= 1;
    return 0;
}

// plsa.py:372-386 as ONE float32 running sum over the non-zeros (PLSA_REFERENCE_LL)
int run_ref_loglik(plsa_ctx *c, const float *d_sw, double *out) {
    if (c->sharded) return fail(c, "the reference arithmetic has no doc-sharded form");
    CHK(ensure_rowidx(c));
    CHK(ensure(c, c->ref_terms, sizeof(float) * (size_t)std::max<i64>(c->nnz, 1)));
    CHK(ensure(c, c->ll_out, sizeof(double)));
    {
        Scope s(c, "k_ref_ll_terms");
        hipLaunchKernelGGL(plsa::ref::k_ref_ll_terms, dim3(grid_for(c, c->nnz, 256)), dim3(256), 0, c->stream,
                           c->rowidx.as<int>(), c->col, c->val, c->nnz, c->U[c->cu].as<float>(), c->Vt[c->cv].as<float>(),
                           d_sw, c->kp, c->ref_terms.as<float>());
    }
    if (ref_pairs_now(c)) {
        // the chain of the NEGATED terms from parity pairs (one "topic"); its slow-chunk counts go to their own slots, unread
        CHK(ensure(c, c->ref_stats, 32));
        CHK(ensure(c, c->ref_ll_neg, sizeof(float)));
        unsigned long long *stats = c->ref_stats.as<unsigned long long>();
        CHK(run_ref_pair_chain(c, plsa::ref::PAIR_NEG_TERMS, c->ref_terms.as<float>(), 1, nullptr, c->ref_ll_neg.as<float>(), stats + 2));
        hipLaunchKernelGGL(plsa::ref::k_ref_ll_from_walk, dim3(1), dim3(1), 0, c->stream, c->ref_ll_neg.as<float>(), c->ll_out.as<double>());
    } else {
        Scope s(c, "k_ref_ll_chain");
        hipLaunchKernelGGL(plsa::ref::k_ref_ll_chain, dim3(1), dim3(64), 0, c->stream, c->ref_terms.as<float>(), c->nnz,
                           c->ll_out.as<double>());
    }
    CHK(launch_check(c, "k_ref_ll_chain"));
    HIPCHK(c, hipMemcpyAsync(c->h_ll, c->ll_out.p, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    *out = *c->h_ll;
    return 0;
}

// PLSA_REFERENCE_SUMS / PLSA_REFERENCE_LL of a driver call: in force for that call on top of plsa_set_arithmetic's setting
struct ArithmeticScope {
    plsa_ctx *c;
    bool sums, ll;
    ArithmeticScope(plsa_ctx *c_, int flags) : c(c_), sums(c_->ref_sums), ll(c_->ref_ll) {
        if (flags & PLSA_REFERENCE_SUMS) c->ref_sums = true;
        if (flags & PLSA_REFERENCE_LL) c->ref_ll = true;
    }
    ~ArithmeticScope() { c->ref_sums = sums; c->ref_ll = ll; }
};

// the reference's stop test, plsa.py:634-638: float32 arithmetic, float64 comparison with tolerance
// (block_parallel_plsa.py:329-331 has no `change == 0` arm: zero_arm = false)
bool stop_test(float cur, float &prev, double tol, bool zero_arm = true) {
    const float change = fabsf(cur - prev);
    if ((zero_arm && change == 0.0f) || (double)(change / fabsf(cur)) < tol) return true;
    prev = cur;
    return false;
}

}  // namespace

// =================================================================================================
// C ABI
// =================================================================================================
extern "C" {

int plsa_device_count(int *count) {
    hipError_t e = hipGetDeviceCount(count);
    if (e != hipSuccess) { *count = 0; return fail(nullptr, "hipGetDeviceCount: %s", hipGetErrorString(e)); }
    return 0;
}

const char *plsa_last_error(const plsa_ctx *ctx) { return ctx ? ctx->err.c_str() : g_err.c_str(); }

int plsa_create(int device, plsa_ctx **out) {
    *out = nullptr;
    int cnt = 0;
    hipError_t e = hipGetDeviceCount(&cnt);
    if (e != hipSuccess || cnt <= 0)
        return fail(nullptr, "no HIP device available (%s)", hipGetErrorString(e));
    if (device < 0 || device >= cnt) return fail(nullptr, "device %d out of range [0,%d)", device, cnt);
    plsa_ctx *c = new plsa_ctx();
    c->device = device;
    if (hipSetDevice(device) != hipSuccess || hipGetDeviceProperties(&c->prop, device) != hipSuccess) {
        delete c;
        return fail(nullptr, "hipSetDevice(%d) failed", device);
    }
    if (strncmp(c->prop.gcnArchName, "gfx950", 6) != 0) {
        std::string arch = c->prop.gcnArchName;
        delete c;
        return fail(nullptr, "libplsa_hip is built for gfx950 (MI355X) only; device %d is %s", device,
                    arch.c_str());
    }
    // the second stream carries the short column tail underneath the document pass: highest priority, so that its
    // few workgroups are placed as soon as slots free up instead of queueing behind the pass' 32 k workgroups
    int prio_lo = 0, prio_hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&prio_lo, &prio_hi);
    const char *prio_env = getenv("PLSA_TAIL_PRIORITY");
    const bool tail_prio = !prio_env || atoi(prio_env) != 0;
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
        (tail_prio ? hipStreamCreateWithPriority(&c->stream2, hipStreamNonBlocking, prio_hi)
                   : hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_row, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_tail, hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_ll, hipEventDisableTiming) != hipSuccess ||
        hipHostMalloc((void **)&c->h_ll, sizeof(double) * 2, hipHostMallocDefault) != hipSuccess) {
        delete c;
        return fail(nullptr, "stream / pinned buffer creation failed");
    }
    c->ls = c->stream;
    if (const char *s = getenv("PLSA_OVERLAP")) c->overlap = atoi(s) != 0;
    if (const char *s = getenv("PLSA_OVERLAP_FULL_LIMIT")) c->overlap_full_limit = atof(s);
    if (const char *s = getenv("PLSA_ROW_ITEMS")) c->ritems_mode = atoi(s);
    if (const char *s = getenv("PLSA_ROW_SEG")) c->rseg_override = std::max(1, atoi(s));
    int mult = 128;  // blocks per CU a grid may hold: large (but bounded) grids measured best
    if (const char *s = getenv("PLSA_GRID_MULT")) mult = std::max(1, atoi(s));
    c->grid_cap = c->prop.multiProcessorCount * mult;
    if (const char *s = getenv("PLSA_CONTIG")) g_contig = atoi(s) != 0;
    if (const char *s = getenv("PLSA_PLACEMENT_CANDIDATES")) c->placement_candidates = std::max(1, atoi(s));
    if (const char *s = getenv("PLSA_COL_SEG")) c->seg_override = std::max(1, atoi(s));
    if (const char *s = getenv("PLSA_HEAVY_ITEMS")) c->heavy_items = std::max(1, atoi(s));
    if (const char *s = getenv("PLSA_SORT_ROWS")) c->sort_rows = atoi(s) != 0;
    if (const char *s = getenv("PLSA_ROW_XCD")) c->row_xcd = atoi(s) != 0;
    if (const char *s = getenv("PLSA_ITEM_ORDER")) c->use_item_order = atoi(s) != 0;
    if (const char *s = getenv("PLSA_XCD_SPLIT")) c->xcd_split = atoi(s) != 0;
    if (const char *s = getenv("PLSA_BALANCE")) c->balance = atoi(s);
    if (const char *s = getenv("PLSA_ORDER_BAND")) c->order_band = atoi(s);
    if (const char *s = getenv("PLSA_GRAPH")) c->graph = atoi(s) != 0;
    if (const char *s = getenv("PLSA_PIPELINE")) c->pipeline = atoi(s) != 0;
    if (const char *s = getenv("PLSA_CHUNKS_PER_LANE")) c->chunks_per_lane = atoi(s);
    if (const char *s = getenv("PLSA_E_ROWS")) c->e_rows = atoi(s);
    if (const char *s = getenv("PLSA_E_SEG")) c->eseg_override = atoi(s);
    if (const char *s = getenv("PLSA_MT_STREAMS")) c->mt_streams = std::max(1, std::min(4096, atoi(s)));
    if (const char *s = getenv("PLSA_MT_MIN_BLOCKS")) c->mt_min_blocks = std::max(1, atoi(s));
    if (const char *s = getenv("PLSA_SMALL_GRID")) c->small_grid = std::max(0, atoi(s));
    if (const char *s = getenv("PLSA_ROW_SHAPE")) c->row_shape_8x2 = atoi(s) != 0;
    if (const char *s = getenv("PLSA_FORCE_WIDE")) c->force_wide = atoi(s) != 0;
    if (const char *s = getenv("PLSA_MT_CHAIN")) c->mt_chain = atoi(s) != 0;
    if (const char *s = getenv("PLSA_SPECULATE")) c->speculate = atoi(s);
    if (const char *s = getenv("PLSA_REF_CHAIN"))      // norm_pwz of the reference arithmetic: auto (default) | pairs | serial
        c->ref_chain_mode = !strcmp(s, "pairs") ? 1 : (!strcmp(s, "serial") ? 2 : 0);
    *out = c;
    return 0;
}

void plsa_destroy(plsa_ctx *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipStreamSynchronize(c->stream);
    if (c->comm) { (void)ncclCommDestroy(c->comm); c->comm = nullptr; }
    release(c->comm_send); release(c->comm_recv); release(c->comm_small); release(c->comm_stack);
    if (c->comm_host) { (void)hipHostFree(c->comm_host); c->comm_host = nullptr; c->comm_host_cap = 0; }
    release(c->item_end); release(c->colsum_rows); release(c->colsum_rows2);
    release(c->item_rec); release(c->xcd_lo); release(c->t_end);
    release(c->mt_words); release(c->mt_state); release(c->mt_fin); release(c->mt_poly); release(c->mt_seq);
    DevBuf *all[] = {&c->b_indptr, &c->b_col, &c->b_val, &c->a_indptr, &c->a_col, &c->a_val, &c->rowidx,
                     &c->colptr, &c->csc_row, &c->csc_val, &c->csc_pos, &c->item_first, &c->item_col,
                     &c->item_start, &c->item_order, &c->partial, &c->heavy_cols, &c->row_order, &c->ritem_first, &c->ritem_row, &c->ritem_start, &c->rpartial, &c->eitem_row, &c->eitem_start, &c->U[0], &c->U[1], &c->U[2], &c->Vt[0], &c->Vt[1], &c->Vt[2], &c->Vacc,
                     &c->P, &c->sw, &c->sw_res, &c->ll_partials, &c->ll_out, &c->colsum_partials, &c->norm_pwz,
                     &c->norm_pdz, &c->tmp0, &c->tmp1, &c->tmp2, &c->cubtmp, &c->ref_terms, &c->ref_csum, &c->ref_pairs, &c->ref_exps, &c->ref_stats, &c->ref_ll_neg, &c->ref_heavy, &c->ref_pairs2, &c->ref_exps2, &c->ref_tsum};
    if (c->p_borrowed) { c->P.p = nullptr; c->P.cap = 0; }      // lent memory is the lender's to free
    for (DevBuf *b : all) release(*b);
    for (auto &t : c->timed) { (void)hipEventDestroy(t.a); (void)hipEventDestroy(t.b); }
    for (auto e : c->pool) (void)hipEventDestroy(e);
    if (c->h_ll) (void)hipHostFree(c->h_ll);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    if (c->ev_join) (void)hipEventDestroy(c->ev_join);
    if (c->ev_row) (void)hipEventDestroy(c->ev_row);
    if (c->ev_tail) (void)hipEventDestroy(c->ev_tail);
    if (c->ev_ll) (void)hipEventDestroy(c->ev_ll);
    if (c->stream2) (void)hipStreamDestroy(c->stream2);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

int plsa_synchronize(plsa_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_hw_queues(void) {
    const char *s = getenv("GPU_MAX_HW_QUEUES");
    return s && atoi(s) > 0 ? atoi(s) : 4;           // 4: the HIP runtime's default
}

int plsa_device_info(plsa_ctx *c, char *name64, char *arch64, int *cus, int64_t *hbm_bytes) {
    if (name64) { strncpy(name64, c->prop.name, 63); name64[63] = 0; }
    if (arch64) { strncpy(arch64, c->prop.gcnArchName, 63); arch64[63] = 0; }
    if (cus) *cus = c->prop.multiProcessorCount;
    if (hbm_bytes) *hbm_bytes = (int64_t)c->prop.totalGlobalMem;
    return 0;
}

int plsa_upload_csr(plsa_ctx *c, const int32_t *indptr, const int32_t *indices, const float *data,
                    int64_t n, int64_t m, int64_t nnz) {
    HIPCHK(c, hipSetDevice(c->device));
    if (n <= 0 || m <= 0 || nnz < 0) return fail(c, "plsa_upload_csr: bad shape n=%lld m=%lld nnz=%lld",
                                                 (long long)n, (long long)m, (long long)nnz);
    if (n >= INT32_MAX || m >= INT32_MAX || nnz >= INT32_MAX)
        return fail(c, "plsa_upload_csr: n, m and nnz must each be < 2^31");
    if (indptr[0] != 0 || indptr[n] != nnz) return fail(c, "plsa_upload_csr: indptr[0] != 0 or indptr[n] != nnz");
    CHK(ensure(c, c->b_indptr, sizeof(int) * (size_t)(n + 1)));
    CHK(ensure(c, c->b_col, sizeof(int) * (size_t)nnz));
    CHK(ensure(c, c->b_val, sizeof(float) * (size_t)nnz));
    HIPCHK(c, hipMemcpyAsync(c->b_indptr.p, indptr, sizeof(int) * (size_t)(n + 1), hipMemcpyHostToDevice, c->stream));
    if (nnz) {      // (large arrays: chunked through page-locked slots by helper threads, see staged_copy)
        CHK(copy_to_device(c, c->b_col.p, indices, sizeof(int) * (size_t)nnz));
        CHK(copy_to_device(c, c->b_val.p, data, sizeof(float) * (size_t)nnz));
    }
    // the header's contract, checked on the device (the arrays are there now; one streaming pass): a violation is a
    // status code for the caller, not a GPU fault three calls later
    CHK(ensure(c, c->tmp2, 16));
    int bad = 0;
    HIPCHK(c, hipMemsetAsync(c->tmp2.p, 0, sizeof(int), c->stream));
    hipLaunchKernelGGL(plsa::k_validate_csr, dim3(grid_for(c, std::max<i64>(n, nnz), 256)), dim3(256), 0, c->stream,
                       c->b_indptr.as<int>(), c->b_col.as<int>(), (i64)n, (i64)m, (i64)nnz, c->tmp2.as<int>());
    CHK(launch_check(c, "k_validate_csr"));
    HIPCHK(c, hipMemcpyAsync(&bad, c->tmp2.p, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad) {
        c->bn = c->bm = c->bnnz = 0;                   // nothing usable is resident
        c->n = c->m = c->nnz = 0;
        c->active_is_base = true;
        return fail(c, "plsa_upload_csr: %s%s%s", (bad & 1) ? "indptr is not non-decreasing within [0, nnz]" : "",
                    bad == 3 ? "; " : "", (bad & 2) ? "column index outside [0, m)" : "");
    }
    c->bn = n; c->bm = m; c->bnnz = nnz;
    c->syn_n = 0;
    c->active_is_base = true;
    set_active_pointers(c);
    return 0;
}

int plsa_bootstrap(plsa_ctx *c, const int64_t *idx, int64_t n_out) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->bn <= 0) return fail(c, "plsa_bootstrap: no corpus uploaded");
    if (!idx) {
        c->active_is_base = true;
        set_active_pointers(c);
        return 0;
    }
    if (n_out <= 0 || n_out >= INT32_MAX) return fail(c, "plsa_bootstrap: bad n_out");
    // idx -> device, row lengths (int64), exclusive scan -> output row pointers
    CHK(ensure(c, c->tmp0, sizeof(i64) * (size_t)n_out));
    CHK(ensure(c, c->tmp1, sizeof(i64) * (size_t)(n_out + 1)));
    CHK(ensure(c, c->tmp2, sizeof(i64) * (size_t)(n_out + 1) + 16));
    i64 *d_idx = c->tmp0.as<i64>();
    int *d_len = c->tmp1.as<int>();
    i64 *d_ptr = c->tmp2.as<i64>();
    int *d_bad = reinterpret_cast<int *>(d_ptr + n_out + 1);
    HIPCHK(c, hipMemcpyAsync(d_idx, idx, sizeof(i64) * (size_t)n_out, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipMemsetAsync(d_bad, 0, sizeof(int), c->stream));
    HIPCHK(c, hipMemsetAsync(d_len, 0, sizeof(int) * (size_t)(n_out + 1), c->stream));
    hipLaunchKernelGGL(plsa::k_boot_lengths, dim3((unsigned)((n_out + 255) / 256)), dim3(256), 0, c->stream,
                       c->b_indptr.as<int>(), d_idx, (i64)n_out, c->bn, d_len, d_bad);
    CHK(launch_check(c, "k_boot_lengths"));
    {
        // int lengths summed into int64 pointers (a resample may exceed the base nnz)
        auto in64 = hipcub::TransformInputIterator<i64, hipcub::CastOp<i64>, int *>(d_len, hipcub::CastOp<i64>());
        size_t bytes = 0;
        HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in64, d_ptr, (int)(n_out + 1), c->stream));
        CHK(ensure(c, c->cubtmp, bytes));
        HIPCHK(c, hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, bytes, in64, d_ptr, (int)(n_out + 1), c->stream));
    }
    i64 total = 0;
    int bad = 0;
    HIPCHK(c, hipMemcpyAsync(&total, d_ptr + n_out, sizeof(i64), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipMemcpyAsync(&bad, d_bad, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (bad) return fail(c, "plsa_bootstrap: row index out of range [0,%lld)", (long long)c->bn);
    if (total >= INT32_MAX) return fail(c, "plsa_bootstrap: resampled nnz %lld >= 2^31", (long long)total);
    CHK(ensure(c, c->a_indptr, sizeof(int) * (size_t)(n_out + 1)));
    CHK(ensure(c, c->a_col, sizeof(int) * (size_t)total));
    CHK(ensure(c, c->a_val, sizeof(float) * (size_t)total));
    {
        Scope s(c, "k_boot_gather");
        hipLaunchKernelGGL(plsa::k_boot_gather, dim3(grid_for(c, n_out + 1, 4)), dim3(256), 0, c->stream,
                           c->b_indptr.as<int>(), c->b_col.as<int>(), c->b_val.as<float>(), d_idx,
                           (i64)n_out, d_ptr, c->a_indptr.as<int>(), c->a_col.as<int>(),
                           c->a_val.as<float>());
    }
    CHK(launch_check(c, "k_boot_gather"));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->active_is_base = false;
    c->n = n_out; c->m = c->bm; c->nnz = total;
    set_active_pointers(c);
    return 0;
}

int plsa_active_shape(plsa_ctx *c, int64_t *n, int64_t *m, int64_t *nnz) {
    if (n) *n = c->n;
    if (m) *m = c->m;
    if (nnz) *nnz = c->nnz;
    return 0;
}

int plsa_download_active_csr(plsa_ctx *c, int32_t *indptr, int32_t *indices, float *data) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n <= 0) return fail(c, "no corpus uploaded");
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (indptr) HIPCHK(c, hipMemcpy(indptr, c->indptr, sizeof(int) * (size_t)(c->n + 1), hipMemcpyDeviceToHost));
    if (indices && c->nnz) HIPCHK(c, hipMemcpy(indices, c->col, sizeof(int) * (size_t)c->nnz, hipMemcpyDeviceToHost));
    if (data && c->nnz) HIPCHK(c, hipMemcpy(data, c->val, sizeof(float) * (size_t)c->nnz, hipMemcpyDeviceToHost));
    return 0;
}

int plsa_set_factors(plsa_ctx *c, const float *U, const float *V, int64_t n, int64_t m, int32_t k) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n <= 0) return fail(c, "plsa_set_factors: upload a corpus first");
    if (n != c->n || m != c->m)
        return fail(c, "plsa_set_factors: factor shapes (n=%lld, m=%lld) do not match the active matrix (%lld x %lld)",
                    (long long)n, (long long)m, (long long)c->n, (long long)c->m);
    if (k <= 0 || k > 1024) return fail(c, "plsa_set_factors: k=%d outside [1,1024]", k);
    if (!U) return fail(c, "plsa_set_factors: U is NULL");
    if (!V && (k != c->k || !c->Vt[0].p)) return fail(c, "plsa_set_factors: V is NULL but no topics with k=%d are resident", k);
    set_shape(c, k);
    const int kp = c->kp;
    c->fac_n = n; c->fac_m = m;
    for (int i = 0; i < 2; ++i) CHK(ensure(c, c->U[i], sizeof(float) * (size_t)n * kp));
    for (int i = 0; i < 2; ++i) CHK(ensure(c, c->Vt[i], sizeof(float) * (size_t)m * kp));
    CHK(ensure(c, c->Vacc, sizeof(float) * (size_t)m * kp));
    c->p_valid = false;
    c->cu = 0;
    if (kp != k) HIPCHK(c, hipMemsetAsync(c->U[0].p, 0, sizeof(float) * (size_t)n * kp, c->stream));
    if (kp == k)
        CHK(copy_to_device(c, c->U[0].p, U, sizeof(float) * (size_t)n * k));
    else
        HIPCHK(c, hipMemcpy2DAsync(c->U[0].p, sizeof(float) * kp, U, sizeof(float) * k, sizeof(float) * k,
                                   (size_t)n, hipMemcpyHostToDevice, c->stream));
    if (V) {
        c->cv = 0;
        CHK(ensure(c, c->tmp0, sizeof(float) * (size_t)k * m));
        HIPCHK(c, hipMemcpyAsync(c->tmp0.p, V, sizeof(float) * (size_t)k * m, hipMemcpyHostToDevice, c->stream));
        dim3 grid((unsigned)((m + 31) / 32), (unsigned)((kp + 31) / 32));
        hipLaunchKernelGGL(plsa::k_v_to_vt, grid, dim3(256), 0, c->stream, c->tmp0.as<float>(),
                           c->Vt[0].as<float>(), k, (int)m, kp);
        CHK(launch_check(c, "k_v_to_vt"));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// Throughput-mode initialisation on the device (counter-based RNG; not the reference's stream).
int plsa_init_factors_device(plsa_ctx *c, int32_t k, uint64_t seed) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n <= 0) return fail(c, "plsa_init_factors_device: upload a corpus first");
    if (k <= 0 || k > 1024) return fail(c, "plsa_init_factors_device: k=%d outside [1,1024]", k);
    const i64 n = c->n, m = c->m;
    set_shape(c, k);
    const int kp = c->kp;
    c->fac_n = n; c->fac_m = m;
    for (int i = 0; i < 2; ++i) CHK(ensure(c, c->U[i], sizeof(float) * (size_t)n * kp));
    for (int i = 0; i < 2; ++i) CHK(ensure(c, c->Vt[i], sizeof(float) * (size_t)m * kp));
    CHK(ensure(c, c->Vacc, sizeof(float) * (size_t)m * kp));
    c->p_valid = false; c->cu = 0; c->cv = 0;
    // P(w|z): uniform draws per (topic, word), topic rows normalised -> generate [k, m] then transpose
    CHK(ensure(c, c->tmp0, sizeof(float) * (size_t)std::max<i64>(k, 4) * m));
    hipLaunchKernelGGL(plsa::k_init_rows, dim3(grid_for(c, k, 4)), dim3(256), 0, c->stream, c->tmp0.as<float>(),
                       (i64)k, (int)m, (int)m, (unsigned long long)plsa::mix64(seed ^ 0x5157ull));
    dim3 grid((unsigned)((m + 31) / 32), (unsigned)((kp + 31) / 32));
    hipLaunchKernelGGL(plsa::k_v_to_vt, grid, dim3(256), 0, c->stream, c->tmp0.as<float>(), c->Vt[0].as<float>(), k, (int)m, kp);
    hipLaunchKernelGGL(plsa::k_init_rows, dim3(grid_for(c, n, 4)), dim3(256), 0, c->stream, c->U[0].as<float>(),
                       n, k, kp, (unsigned long long)plsa::mix64(seed ^ 0xD0C5ull));
    CHK(launch_check(c, "k_init_rows"));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// plsa_init(X, k, init="random", rng) + the float32 casts of plsa_fit (plsa.py:455-456, 510-511,
// 709-710) with the reference's own MT19937 stream, evaluated on the device.  state_io: the 624 key
// words + position of numpy.random.RandomState.get_state(); on return it holds the advanced state.
// V_host == nullptr: plsa_init (topics drawn first, then the document rows).  V_host != nullptr: the
// refit initialisation (plsa.py:979-981) -- only rand(n, k) is drawn, the topics are the given ones.
static int mt_init(plsa_ctx *c, int32_t k, uint32_t *state_io /*[625]*/, const float *V_host) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->n <= 0) return fail(c, "plsa_init_factors_mt19937: upload a corpus first");
    if (k <= 0 || k > 1024) return fail(c, "plsa_init_factors_mt19937: k=%d outside [1,1024]", k);
    if (state_io[624] > 624) return fail(c, "plsa_init_factors_mt19937: bad generator position");
    const i64 n = c->n, m = c->m;
    set_shape(c, k);
    const int kp = c->kp;
    c->fac_n = n; c->fac_m = m;
    for (int i = 0; i < 2; ++i) CHK(ensure(c, c->U[i], sizeof(float) * (size_t)n * kp));
    for (int i = 0; i < 2; ++i) CHK(ensure(c, c->Vt[i], sizeof(float) * (size_t)m * kp));
    CHK(ensure(c, c->Vacc, sizeof(float) * (size_t)m * kp));
    c->p_valid = false; c->cu = 0; c->cv = 0;
    const i64 v_doubles = V_host ? 0 : (i64)k * m;
    const i64 n_doubles = v_doubles + n * (i64)k;
    const i64 n_words = 2 * n_doubles;
    // Split the stream into pieces that start 2^b blocks apart (csrc/mt_jump.hpp) once it is long
    // enough to pay for the jumps; the words produced are those of the one sequential stream.
    const int pos0 = (int)state_io[624];
    const i64 first = std::min<i64>(624 - pos0, n_words);
    const i64 total_blocks = (n_words - first + 623) / 624;
    i64 per_stream = std::max<i64>(total_blocks, 1);
    int n_streams = 1, log2_per = 0;
    if (c->mt_streams > 1 && total_blocks >= c->mt_min_blocks) {
        while (((i64)1 << log2_per) * c->mt_streams < total_blocks) ++log2_per;
        per_stream = (i64)1 << log2_per;
        n_streams = (int)((total_blocks + per_stream - 1) / per_stream);
    }
    int streams_p2 = 1, levels = 0;
    while (streams_p2 < n_streams) { streams_p2 *= 2; ++levels; }
    std::vector<uint32_t> polys((size_t)levels * 624);
    for (int l = 0; l < levels; ++l)        // level l jumps by (streams_p2 >> (l+1)) * per_stream blocks
        if (!mtjump::block_jump_polynomial(log2_per + (levels - 1 - l), polys.data() + (size_t)l * 624))
            return fail(c, "plsa_init_factors_mt19937: MT19937 characteristic polynomial not recovered");
    DevBuf &words = c->mt_words, &st = c->mt_state, &fin = c->mt_fin, &gp = c->mt_poly;
    CHK(ensure(c, words, sizeof(unsigned) * (size_t)n_words));
    CHK(ensure(c, st, sizeof(unsigned) * 624 * (size_t)streams_p2));
    CHK(ensure(c, fin, sizeof(unsigned) * 640 + sizeof(double) * 1024));
    if (levels) CHK(ensure(c, gp, sizeof(unsigned) * polys.size()));
    if (!V_host && !c->mt_chain)     // chunk sums (u64), parity pairs (2 x u64), tile sums (f64) and binade guesses (int) of the topic marginals
        CHK(ensure(c, c->mt_seq, (size_t)k * (size_t)((m + plsa::MT_SEQ_L - 1) / plsa::MT_SEQ_L) * (3 * sizeof(plsa::u64) + sizeof(int) + sizeof(double))));
    hipError_t e = hipMemsetAsync(st.p, 0, sizeof(unsigned) * 624 * (size_t)streams_p2, c->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(st.p, state_io, sizeof(unsigned) * 624, hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess && levels)
        e = hipMemcpyAsync(gp.p, polys.data(), sizeof(unsigned) * polys.size(), hipMemcpyHostToDevice, c->stream);
    if (e == hipSuccess) {
        if (levels) {
            Scope s(c, "k_mt_jump");
            for (int l = 0; l < levels; ++l) {
                const int step = streams_p2 >> (l + 1);
                hipLaunchKernelGGL(plsa::k_mt_jump, dim3((unsigned)(streams_p2 / (2 * step)), plsa::MT_JUMP_SLICES), dim3(256), 0,
                                   c->stream, gp.as<unsigned>() + (size_t)l * 624, st.as<unsigned>(), step, n_streams);
            }
        }
        { Scope s(c, "k_mt19937_fill");
          hipLaunchKernelGGL(plsa::k_mt19937_fill, dim3((unsigned)n_streams), dim3(PLSA_MT_THREADS), 0, c->stream,
                             st.as<unsigned>(), words.as<unsigned>(), pos0, per_stream, n_words, n_streams, fin.as<unsigned>()); }
        if (V_host) {
            float *Vtmp = reinterpret_cast<float *>(c->Vt[1].p);     // [m*kp] floats >= k*m: free scratch here
            e = hipMemcpyAsync(Vtmp, V_host, sizeof(float) * (size_t)k * m, hipMemcpyHostToDevice, c->stream);
            dim3 grid((unsigned)((m + 31) / 32), (unsigned)((kp + 31) / 32));
            if (e == hipSuccess)
                hipLaunchKernelGGL(plsa::k_v_to_vt, grid, dim3(256), 0, c->stream, Vtmp, c->Vt[0].as<float>(), k, (int)m, kp);
        } else {
            // V[k, m] in the reference layout (words are consumed in that order), then the layout transpose
            float *Vtmp = reinterpret_cast<float *>(c->Vt[1].p);     // [m*kp] floats >= k*m: free scratch here
            double *marg = reinterpret_cast<double *>(fin.as<unsigned>() + 632) ;   // k <= 1024 doubles behind the state
            if (c->mt_chain) {       // PLSA_MT_CHAIN=1: the plain chain of m dependent adds per topic (A/B, tests)
                hipLaunchKernelGGL(plsa::k_mt_marginal_v, dim3((unsigned)k), dim3(64), 0, c->stream, words.as<unsigned>(), k, (int)m, marg);
            } else {                 // the same roundings from per-chunk parity pairs (plsa_kernels.hpp: k_mt_chunk_pairs)
                const int nch = (int)((m + plsa::MT_SEQ_L - 1) / plsa::MT_SEQ_L);
                const size_t per = (size_t)k * nch;
                plsa::u64 *csum = c->mt_seq.as<plsa::u64>(), *pairs = csum + per;
                double *tsum = reinterpret_cast<double *>(pairs + 2 * per);          // [k][tiles] <= per doubles
                int *guess = reinterpret_cast<int *>(tsum + per);
                const dim3 tiles((unsigned)((nch + plsa::MT_SEQ_L - 1) / plsa::MT_SEQ_L), (unsigned)k);
                hipLaunchKernelGGL(plsa::k_mt_chunk_sums, tiles, dim3(64), 0, c->stream, words.as<unsigned>(), (int)m, nch, csum, tsum);
                hipLaunchKernelGGL(plsa::k_mt_chunk_pairs, tiles, dim3(64), 0, c->stream, words.as<unsigned>(), (int)m, nch,
                                   csum, tsum, pairs, guess);
                hipLaunchKernelGGL(plsa::k_mt_marginal_walk, dim3((unsigned)k), dim3(64), 0, c->stream, words.as<unsigned>(),
                                   (int)m, nch, pairs, guess, marg);
            }
            hipLaunchKernelGGL(plsa::k_mt_scale_v, dim3(grid_for(c, (i64)k * m, 256)), dim3(256), 0, c->stream,
                               words.as<unsigned>(), marg, k, (int)m, Vtmp);
            dim3 grid((unsigned)((m + 31) / 32), (unsigned)((kp + 31) / 32));
            hipLaunchKernelGGL(plsa::k_v_to_vt, grid, dim3(256), 0, c->stream, Vtmp, c->Vt[0].as<float>(), k, (int)m, kp);
        }
        hipLaunchKernelGGL(plsa::k_mt_init_u, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, c->stream,
                           words.as<unsigned>(), v_doubles, n, k, kp, c->U[0].as<float>());
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e == hipSuccess) e = hipMemcpyAsync(state_io, fin.p, sizeof(unsigned) * 625, hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    if (e != hipSuccess) return fail(c, "plsa_init_factors_mt19937: %s", hipGetErrorString(e));
    return 0;
}

int plsa_init_factors_mt19937(plsa_ctx *c, int32_t k, uint32_t *state_io /*[625]*/) {
    if (!state_io) return fail(c, "plsa_init_factors_mt19937: state_io is NULL");
    return mt_init(c, k, state_io, nullptr);
}

int plsa_mt_marginals(plsa_ctx *c, double *out, int32_t k) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!out || k <= 0 || k > 1024 || !c->mt_fin.p || k != c->k)
        return fail(c, "plsa_mt_marginals: no MT19937 initialisation with k=%d on this context", k);
    HIPCHK(c, hipMemcpy(out, c->mt_fin.as<unsigned>() + 632, sizeof(double) * (size_t)k, hipMemcpyDeviceToHost));
    return 0;
}

int plsa_refit_init_mt19937(plsa_ctx *c, const float *V, int64_t m, int32_t k, uint32_t *state_io /*[625]*/) {
    if (!state_io || !V) return fail(c, "plsa_refit_init_mt19937: NULL argument");
    if (m != c->m) return fail(c, "plsa_refit_init_mt19937: topics have %lld words, the active matrix %lld", (long long)m, (long long)c->m);
    return mt_init(c, k, state_io, V);
}

int plsa_get_factors(plsa_ctx *c, float *U, float *V) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    if (V) {        // transposed on the device first: it is in flight while P(z|d) travels
        CHK(ensure(c, c->tmp0, sizeof(float) * (size_t)c->k * c->m));
        dim3 grid((unsigned)((c->m + 31) / 32), (unsigned)((c->kp + 31) / 32));
        hipLaunchKernelGGL(plsa::k_vt_to_v, grid, dim3(256), 0, c->stream, c->Vt[c->cv].as<float>(),
                           c->tmp0.as<float>(), c->k, (int)c->m, c->kp);
        CHK(launch_check(c, "k_vt_to_v"));
    }
    if (U) {
        if (c->kp == c->k)      // rows are contiguous: one linear copy (the strided 2-D form ran at 10 GB/s from 1 M x 64)
            CHK(copy_to_host(c, U, c->U[c->cu].p, sizeof(float) * (size_t)c->n * c->k));
        else
            HIPCHK(c, hipMemcpy2DAsync(U, sizeof(float) * c->k, c->U[c->cu].p, sizeof(float) * c->kp,
                                       sizeof(float) * c->k, (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    }
    if (V) CHK(copy_to_host(c, V, c->tmp0.p, sizeof(float) * (size_t)c->k * c->m));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_copy_components_to_device(plsa_ctx *c, void *dst) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    dim3 grid((unsigned)((c->m + 31) / 32), (unsigned)((c->kp + 31) / 32));
    hipLaunchKernelGGL(plsa::k_vt_to_v, grid, dim3(256), 0, c->stream, c->Vt[c->cv].as<float>(),
                       reinterpret_cast<float *>(dst), c->k, (int)c->m, c->kp);
    CHK(launch_check(c, "k_vt_to_v"));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_set_arithmetic(plsa_ctx *c, int32_t mode) {
    if (mode & ~(PLSA_REFERENCE_SUMS | PLSA_REFERENCE_LL))
        return fail(c, "plsa_set_arithmetic: mode %d is not a combination of PLSA_REFERENCE_SUMS and PLSA_REFERENCE_LL", mode);
    c->ref_sums = (mode & PLSA_REFERENCE_SUMS) != 0;
    c->ref_ll = (mode & PLSA_REFERENCE_LL) != 0;
    return 0;
}

int plsa_e_step(plsa_ctx *c, float thresh, float *P_out) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    CHK(run_e_step(c, thresh));
    if (P_out && c->nnz)
        HIPCHK(c, hipMemcpy2DAsync(P_out, sizeof(float) * c->k, p_base(c), sizeof(float) * c->kp,
                                   sizeof(float) * c->k, (size_t)c->nnz, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_set_p(plsa_ctx *c, const float *P) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    if (c->p_borrowed || c->p_lent) {
        if (c->P.cap < sizeof(float) * (size_t)(c->nnz + 64) * c->kp)
            return fail(c, "plsa_set_p: the %s P(z|w,d) buffer is too small (it cannot be re-allocated while shared)", c->p_borrowed ? "borrowed" : "lent");
    } else
    CHK(ensure(c, c->P, sizeof(float) * (size_t)(c->nnz + 64) * c->kp + c->p_shift));
    if (c->kp != c->k) HIPCHK(c, hipMemsetAsync(p_base(c), 0, sizeof(float) * (size_t)c->nnz * c->kp, c->stream));
    if (c->nnz)
        HIPCHK(c, hipMemcpy2DAsync(p_base(c), sizeof(float) * c->kp, P, sizeof(float) * c->k,
                                   sizeof(float) * c->k, (size_t)c->nnz, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    c->p_valid = true;
    c->ref_tsum_valid = false;
    return 0;
}

int plsa_m_step(plsa_ctx *c, const float *sw, int32_t update_v, float *norm_pwz, float *norm_pdz) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    const float *d_sw = nullptr;
    CHK(upload_sw(c, sw, &d_sw));
    CHK(ensure(c, c->norm_pdz, sizeof(float) * (size_t)c->n));
    CHK(run_m_step_from_p(c, d_sw, update_v != 0, c->norm_pdz.as<float>()));
    if (norm_pwz && update_v)
        HIPCHK(c, hipMemcpyAsync(norm_pwz, c->norm_pwz.p, sizeof(float) * (size_t)c->k, hipMemcpyDeviceToHost, c->stream));
    if (norm_pdz)
        HIPCHK(c, hipMemcpyAsync(norm_pdz, c->norm_pdz.p, sizeof(float) * (size_t)c->n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_log_likelihood(plsa_ctx *c, const float *sw, double *ll) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    const float *d_sw = nullptr;
    CHK(upload_sw(c, sw, &d_sw));
    return run_loglik(c, d_sw, ll);
}

// plsa_fit_inner, enstop/plsa.py:583-640.
//
// Materialised mode (flags without PLSA_FUSED) follows the reference's kernel sequence literally:
// E-step -> M-step -> (every n_iter_per_test iterations) log-likelihood.
//
// Fused mode never writes P(z|w,d).  The log-likelihood the reference evaluates after iteration i
// (i % n_iter_per_test == 0) is the likelihood of the factors that iteration i+1 reads, so it is
// accumulated for free inside iteration i+1's document pass; the stop decision therefore arrives one
// pass late and, when it says "stop", iteration i+1's output (sitting in the alternate buffers) is
// discarded by not swapping -- the returned factors and iteration count are exactly the reference's.
int plsa_fit(plsa_ctx *c, const float *sw, int32_t n_iter, int32_t n_iter_per_test, double tolerance,
             float thresh, int32_t flags, int32_t *iters_done, float *ll_trace, int32_t *n_ll) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    if (n_iter < 0 || n_iter_per_test <= 0) return fail(c, "plsa_fit: bad n_iter / n_iter_per_test");
    ArithmeticScope arithmetic_scope(c, flags);
    if ((c->ref_sums || c->ref_ll) && (flags & PLSA_SHARDED))
        return fail(c, "plsa_fit: PLSA_REFERENCE_SUMS / PLSA_REFERENCE_LL have no doc-sharded form (the reference's sums are single chains over all non-zeros)");
    // the reference arithmetic IS the reference's kernel sequence (E-step into P(z|w,d), M-step from it, likelihood): there is
    // one such arithmetic, so PLSA_FUSED has nothing to select there and is ignored
    const bool fused = (flags & PLSA_FUSED) && !c->ref_sums && !c->ref_ll, trace = flags & PLSA_TRACE_LL;
    const bool zero_arm = !(flags & PLSA_STOP_NO_ZERO_ARM);
    struct ShardedScope {           // PLSA_SHARDED: this context's rows are one shard of the corpus
        plsa_ctx *c;
        ShardedScope(plsa_ctx *c_, bool on) : c(c_) { c->sharded = on; }
        ~ShardedScope() { c->sharded = false; }
    } sharded_scope(c, (flags & PLSA_SHARDED) != 0);
    const float *d_sw = nullptr;
    CHK(upload_sw(c, sw, &d_sw));
    // plsa.py:606-628: with use_sample_weights == False the M-step ignores the weights, the
    // log-likelihood (plsa.py:591, 631) still applies them
    const float *d_sw_m = (flags & PLSA_SW_LL_ONLY) ? nullptr : d_sw;
    int nll = 0, iters = 0;
    double ll = 0.0;
    float prev = 0.f;
    // plsa.py:591: the likelihood of the initial factors.  The fused schedule gets it for free from the
    // first iteration's document pass (which reads exactly those factors) instead of a separate launch.
    bool first_ll_in_pass = fused && n_iter > 0;
    if (!first_ll_in_pass) {
        CHK(run_loglik(c, d_sw, &ll));
        prev = (float)ll;
        if (ll_trace) ll_trace[nll] = prev;
        nll++;
    }

    if (!fused) {
        struct ESw { plsa_ctx *c; ~ESw() { c->ref_e_sw = nullptr; } } e_sw_guard{c};
        c->ref_e_sw = d_sw_m;               // (reference arithmetic: the E-step leaves its tile sums with the weights the M-step will use)
        for (int i = 0; i < n_iter; ++i) {
            CHK(run_e_step(c, thresh));                              // plsa.py:597
            CHK(run_m_step_from_p(c, d_sw_m, true, nullptr));       // plsa.py:606-628
            iters++;
            if (i % n_iter_per_test == 0) {                          // plsa.py:630
                if (i == n_iter - 1 && !trace) break;                // outcome cannot matter any more
                CHK(run_loglik(c, d_sw, &ll));
                const float cur = (float)ll;
                if (ll_trace) ll_trace[nll] = cur;
                nll++;
                if (stop_test(cur, prev, tolerance, zero_arm)) break;
            }
        }
    } else {
        bool pending = false;  // a test is due on the factors currently in (cu, cv)
        bool stopped = false;
        const bool graph_requested = ((flags & PLSA_GRAPH) || c->graph) && !c->sharded && !c->timing;
        // small corpora: column chain and document pass as two pipelines that exchange events (see below)
        const bool pipelined = c->overlap && !c->sharded && !graph_requested && c->pipeline &&
                               (double)c->nnz * c->kp < c->overlap_full_limit;
        if (pipelined) {      // everything enqueued so far (factors, corpus) precedes both pipelines
            HIPCHK(c, hipEventRecord(c->ev_row, c->stream));
            HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_row, 0));
            HIPCHK(c, hipEventRecord(c->ev_tail, c->stream2));
        }
        struct PipelineJoin {    // whatever way the loop is left, the column chain has finished before the call returns
            plsa_ctx *c; bool on;
            ~PipelineJoin() { if (on) { (void)hipStreamSynchronize(c->stream2); } }
        } pipeline_join{c, pipelined};
        // Speculation across a likelihood test.  The test after iteration i - 1 rides on iteration i's document pass; the
        // host used to wait for it before enqueuing iteration i + 1 -- an idle chip for one host round trip plus the launch
        // latency of the next passes, every n_iter_per_test iterations (config 2: ~40 us per test = 2.3 % of the run).
        // With a THIRD set of factor buffers iteration i + 1 is enqueued first (it reads iteration i's output and writes the
        // third set, so the factors a "stop" must return are still untouched) and the host then waits for the likelihood
        // alone.  "Stop" discards two iterations instead of one; the returned factors, count and trace are the same.
        const bool speculate = (c->speculate > 0 || (c->speculate < 0 && (double)c->nnz * c->kp < c->overlap_full_limit)) &&
                               !c->sharded && !graph_requested && n_iter_per_test >= 2 && n_iter >= 3;
        struct Rot3Scope {       // leaves cu, cv in {0, 1} (everything outside this loop addresses the alternate as 1 - cu)
            plsa_ctx *c;
            ~Rot3Scope() {
                if (!c->rot3) return;
                c->rot3 = false;
                if (c->cu == 2) { std::swap(c->U[2], c->U[0]); c->cu = 0; }
                if (c->cv == 2) { std::swap(c->Vt[2], c->Vt[0]); c->cv = 0; }
            }
        } rot3_scope{c};
        if (speculate) {
            CHK(ensure(c, c->U[2], sizeof(float) * (size_t)c->n * c->kp));
            CHK(ensure(c, c->Vt[2], sizeof(float) * (size_t)c->m * c->kp));
            c->rot3 = true;
        }
        bool first_wait = false;     // the initial likelihood is in flight (speculating loop)
        int first_slot = 0;
        auto collect_first = [&]() -> int {
            if (!first_wait) return 0;
            first_wait = false;
            CHK(wait_ll(c, &ll));
            prev = (float)ll;
            if (ll_trace) ll_trace[first_slot] = prev;
            return 0;
        };
        auto advance = [&]() {
            if (c->rot3) { c->cu = (c->cu + 1) % 3; c->cv = (c->cv + 1) % 3; } else { c->cu ^= 1; c->cv ^= 1; }
        };
        // one fused EM iteration from the factors in (cu, cv) into the alternate buffers (no swap here)
        auto enqueue_iteration = [&](bool want_ll, int *blocks) -> int {
            // PLSA_SHARDED: every collective of the communicator goes on c->stream in program order (accumulator
            // all-reduce, then the likelihood all-reduce) -- no second stream, identical order on every rank
            const bool overlap = c->overlap && !c->sharded;
            if (overlap && (double)c->nnz * c->kp < c->overlap_full_limit) {
                // small problems leave CUs idle inside each kernel (measured: config 1 0.50 -> 0.37 ms,
                // config 2 0.43 -> 0.37 ms per iteration; neutral at config 3, -6 % at config 5):
                // the document pass (VALU-heavy, gathers the small topic table) and the column chain
                // (fabric-bound gathers of P(z|d) rows) read the same current factors and write
                // disjoint outputs: run them on two streams so their stalls overlap
                CHK(ensure_csc(c));
                CHK(ensure_ritems(c));
                const int *unused = nullptr;
                if (!c->use_ritems) CHK(ensure_roworder(c, &unused));
                if (pipelined) {
                    // The column chain of successive iterations is one dependency chain (column pass -> tail -> next
                    // column pass): it stays back to back on the second stream, and the two streams only exchange
                    // "document pass i done" / "column chain i done" events.  A fork + join through the first stream
                    // put two cross-stream hops (17 us of 121 at config 1) between tail i and column pass i+1.
                    HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_tail, 0));      // P(w|z) of the previous chain
                    HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_row, 0));      // P(z|d) of the previous document pass
                    c->ls = c->stream2;
                    int rc = run_col_pass(c, false, d_sw_m, thresh, 1);
                    if (!rc) rc = run_col_tail(c);
                    c->ls = c->stream;
                    if (rc) return rc;
                    HIPCHK(c, hipEventRecord(c->ev_tail, c->stream2));
                    CHK(run_row_pass(c, false, want_ll, d_sw, thresh, nullptr, blocks));
                    HIPCHK(c, hipEventRecord(c->ev_row, c->stream));
                    return 0;
                }
                HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
                c->ls = c->stream2;
                int rc = run_col_pass(c, false, d_sw_m, thresh, 1);
                if (!rc) rc = run_col_tail(c);
                c->ls = c->stream;
                if (rc) return rc;
                HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
                CHK(run_row_pass(c, false, want_ll, d_sw, thresh, nullptr, blocks));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
            } else if (overlap) {
                // large problems: both passes saturate the memory system on their own (running them
                // side by side is neutral at config 3, -6 % at config 5), but the short chain of
                // column sums / normalisation after the column pass leaves the chip nearly idle --
                // it runs on the second stream underneath the document pass
                CHK(run_col_pass(c, false, d_sw_m, thresh, 1));
                HIPCHK(c, hipEventRecord(c->ev_fork, c->stream));
                HIPCHK(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
                c->ls = c->stream2;
                int rc = run_col_tail(c);
                c->ls = c->stream;
                if (rc) return rc;
                HIPCHK(c, hipEventRecord(c->ev_join, c->stream2));
                CHK(run_row_pass(c, false, want_ll, d_sw, thresh, nullptr, blocks));
                HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
            } else {
                CHK(run_row_pass(c, false, want_ll, d_sw, thresh, nullptr, blocks));
                CHK(run_col_pass(c, false, d_sw_m, thresh, 1));
                CHK(run_col_tail(c));
            }
            return 0;
        };
        // PLSA_GRAPH (flag or environment): the iterations between two likelihood tests replayed from a hipGraph
        // of TWO iterations (the double buffers alternate, a pair returns to the starting buffers; the single
        // eager iteration of a likelihood test flips the parity, hence one graph per starting pair), captured
        // from the launch sequence above.  Kernel arguments are baked into the graph, so it lives for this
        // call only.  Off by default: measured neutral (the host is not the bound, and a dependent kernel
        // boundary costs the same 1.5 us eager or replayed)
        const bool use_graph = graph_requested;
        hipGraphExec_t gexecs[4] = {nullptr, nullptr, nullptr, nullptr};   // one per starting buffer pair (cu, cv)
        struct GraphGuard {
            hipGraphExec_t (&g)[4];
            ~GraphGuard() { for (auto e : g) if (e) (void)hipGraphExecDestroy(e); }
        } graph_guard{gexecs};
        for (int i = 0; i < n_iter; ++i) {
            int blocks = 0;
            const bool want_ll = pending || first_ll_in_pass;
            if (use_graph && !want_ll && i + 1 < n_iter && (i % n_iter_per_test) != 0) {
                // iterations i and i + 1, neither carries a likelihood test
                hipGraphExec_t &gexec = gexecs[c->cu * 2 + c->cv];
                if (!gexec) {
                    hipGraph_t graph = nullptr;
                    HIPCHK(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
                    int rc = enqueue_iteration(false, &blocks);
                    c->cu ^= 1; c->cv ^= 1;
                    if (!rc) rc = enqueue_iteration(false, &blocks);
                    c->cu ^= 1; c->cv ^= 1;
                    const hipError_t e_end = hipStreamEndCapture(c->stream, &graph);
                    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
                    HIPCHK(c, e_end);
                    const hipError_t e_inst = hipGraphInstantiate(&gexec, graph, nullptr, nullptr, 0);
                    (void)hipGraphDestroy(graph);
                    HIPCHK(c, e_inst);
                }
                HIPCHK(c, hipGraphLaunch(gexec, c->stream));
                iters += 2;
                ++i;
                pending = (i % n_iter_per_test == 0);
                continue;
            }
            CHK(enqueue_iteration(want_ll, &blocks));
            if (first_ll_in_pass) {
                if (speculate) {       // nothing is decided on the initial likelihood: it is collected when the first test needs it
                    CHK(finish_ll(c, blocks, nullptr));
                    first_wait = true;
                    first_slot = nll;
                } else {
                    CHK(finish_ll(c, blocks, &ll));
                    prev = (float)ll;
                    if (ll_trace) ll_trace[nll] = prev;
                }
                nll++;
                first_ll_in_pass = false;
            } else if (pending && speculate && i + 1 < n_iter) {
                CHK(collect_first());
                CHK(finish_ll(c, blocks, nullptr));            // on its way to the host; not waited for yet
                const int su = c->cu, sv = c->cv;              // the factors a stop returns
                advance();                                     // iteration i + 1 reads iteration i's output ...
                int blocks2 = 0;
                const int rc = enqueue_iteration(false, &blocks2);   // ... and writes the third set (n_iter_per_test >= 2: no test rides on it)
                if (rc) { c->cu = su; c->cv = sv; return rc; }
                CHK(wait_ll(c, &ll));
                const float cur = (float)ll;
                if (ll_trace) ll_trace[nll] = cur;
                nll++;
                if (stop_test(cur, prev, tolerance, zero_arm)) {     // discard both passes
                    c->cu = su; c->cv = sv;
                    stopped = true;
                    break;
                }
                advance();
                iters += 2;
                ++i;                                           // (iteration i + 1 is done; it is no multiple-of-test successor)
                pending = (i % n_iter_per_test == 0);
                continue;
            } else if (pending) {
                CHK(collect_first());
                CHK(finish_ll(c, blocks, &ll));
                const float cur = (float)ll;
                if (ll_trace) ll_trace[nll] = cur;
                nll++;
                if (stop_test(cur, prev, tolerance, zero_arm)) { stopped = true; break; }  // discard this pass
            }
            advance();
            iters++;
            pending = (i % n_iter_per_test == 0);
        }
        CHK(collect_first());
        if (pipelined) HIPCHK(c, hipStreamWaitEvent(c->stream, c->ev_tail, 0));   // the last column chain precedes whatever follows
        if (!stopped && pending && trace) {  // test of the last iteration: result-neutral
            CHK(run_loglik(c, d_sw, &ll));
            if (ll_trace) ll_trace[nll] = (float)ll;
            nll++;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (iters_done) *iters_done = iters;
    if (n_ll) *n_ll = nll;
    return 0;
}

// plsa_refit_inner, enstop/plsa.py:884-920: topics frozen, only P(z|d) moves.
int plsa_refit(plsa_ctx *c, const float *sw, int32_t n_iter, int32_t n_iter_per_test, double tolerance,
               float thresh, int32_t flags, int32_t *iters_done, float *ll_trace, int32_t *n_ll) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    if (n_iter < 0 || n_iter_per_test <= 0) return fail(c, "plsa_refit: bad n_iter / n_iter_per_test");
    ArithmeticScope arithmetic_scope(c, flags);
    const bool fused = (flags & PLSA_FUSED) && !c->ref_sums && !c->ref_ll, trace = flags & PLSA_TRACE_LL;
    const float *d_sw = nullptr;
    CHK(upload_sw(c, sw, &d_sw));
    int nll = 0, iters = 0;
    double ll = 0.0;
    float prev = 0.f;
    bool first_ll_in_pass = fused && n_iter > 0;       // as in plsa_fit: the initial likelihood rides on pass 0
    if (!first_ll_in_pass) {
        CHK(run_loglik(c, d_sw, &ll));
        prev = (float)ll;
        if (ll_trace) ll_trace[nll] = prev;
        nll++;
    }
    // plsa.py:913-918: the test only acts on a positive log-likelihood
    auto refit_stop = [&](float cur) {
        if (cur > 0.0f) {
            const float change = fabsf(cur - prev);
            if ((double)(change / fabsf(cur)) < tolerance) return true;
            prev = cur;
        }
        return false;
    };
    if (!fused) {
        struct NoSums { plsa_ctx *c; ~NoSums() { c->ref_e_no_sums = false; } } no_sums_guard{c};
        c->ref_e_no_sums = true;            // (reference arithmetic: no norm_pwz chain follows these E-steps, their tile sums would be wasted)
        for (int i = 0; i < n_iter; ++i) {
            CHK(run_e_step(c, thresh));
            CHK(run_m_step_from_p(c, nullptr, false, nullptr));
            iters++;
            if (i % n_iter_per_test == 0) {
                if (i == n_iter - 1 && !trace) break;
                CHK(run_loglik(c, d_sw, &ll));
                const float cur = (float)ll;
                if (ll_trace) ll_trace[nll] = cur;
                nll++;
                if (refit_stop(cur)) break;
            }
        }
    } else {
        bool pending = false, stopped = false;
        for (int i = 0; i < n_iter; ++i) {
            int blocks = 0;
            // the refit M-step ignores sample weights for P(z|d) (plsa.py:806-809); they only enter
            // the log-likelihood, which this pass accumulates when a test is pending
            CHK(run_row_pass(c, false, pending || first_ll_in_pass, d_sw, thresh, nullptr, &blocks));
            if (first_ll_in_pass) {
                CHK(finish_ll(c, blocks, &ll));
                prev = (float)ll;
                if (ll_trace) ll_trace[nll] = prev;
                nll++;
                first_ll_in_pass = false;
            } else if (pending) {
                CHK(finish_ll(c, blocks, &ll));
                const float cur = (float)ll;
                if (ll_trace) ll_trace[nll] = cur;
                nll++;
                if (refit_stop(cur)) { stopped = true; break; }
            }
            c->cu ^= 1;
            iters++;
            pending = (i % n_iter_per_test == 0);
        }
        if (!stopped && pending && trace) {
            CHK(run_loglik(c, d_sw, &ll));
            if (ll_trace) ll_trace[nll] = (float)ll;
            nll++;
        }
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (iters_done) *iters_done = iters;
    if (n_ll) *n_ll = nll;
    return 0;
}

// ---- doc-sharded single fit: local accumulate / (caller all-reduces) / finish -----------------------
int plsa_em_accumulate(plsa_ctx *c, const float *sw, float thresh, double *ll_partial) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    if (c->ref_sums || c->ref_ll) return fail(c, "plsa_em_accumulate: the reference arithmetic (plsa_set_arithmetic) has no doc-sharded form");
    const float *d_sw = nullptr;
    CHK(upload_sw(c, sw, &d_sw));
    int blocks = 0;
    CHK(run_row_pass(c, false, ll_partial != nullptr, d_sw, thresh, nullptr, &blocks));
    CHK(run_col_pass(c, false, d_sw, thresh));
    if (ll_partial) CHK(finish_ll(c, blocks, ll_partial));    // (its scalar read-back is the only host wait)
    return 0;
}

// The reference's kernel SEQUENCE over the local rows (E-step into P(z|w,d), M-step from it) with the accumulate / finish
// split of the doc-sharded fit: what a doc-block TILE of block_parallel_plsa.py:373-403 does -- responsibilities of the block,
// partial factors of the block (:182-185) -- with the block's P(z|w,d) alive only inside this call.
int plsa_em_accumulate_materialised(plsa_ctx *c, const float *sw, float thresh, double *ll_partial) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    if (c->ref_sums || c->ref_ll) return fail(c, "plsa_em_accumulate_materialised: the reference arithmetic (plsa_set_arithmetic) has no doc-block form");
    const float *d_sw = nullptr;
    CHK(upload_sw(c, sw, &d_sw));
    if (ll_partial) CHK(run_loglik(c, d_sw, ll_partial));       // log-likelihood of the CURRENT factors, local rows
    CHK(run_e_step(c, thresh));
    CHK(run_row_pass(c, true, false, nullptr, 0.f, nullptr, nullptr));
    CHK(run_col_pass(c, true, d_sw, 0.f));                      // partial sums + per-column sums into the accumulator
    // P(z|w,d) may be another context's by the next call (plsa_p_borrow): its last reader has finished when this returns
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    c->p_valid = false;
    return 0;
}

// P(z|w,d) capacity of this context: at least `bytes` (own allocation); *device_ptr = its address.
int plsa_p_reserve(plsa_ctx *c, int64_t bytes, void **device_ptr) {
    HIPCHK(c, hipSetDevice(c->device));
    if (bytes <= 0 || !device_ptr) return fail(c, "plsa_p_reserve: bad arguments");
    if (c->p_borrowed) return fail(c, "plsa_p_reserve: this context borrows its P(z|w,d) buffer");
    if (c->p_lent && c->P.cap < (size_t)bytes)
        return fail(c, "plsa_p_reserve: the buffer handed out earlier (%.2f GB) cannot grow to %.2f GB while other contexts may hold its "
                       "address: end the loans (plsa_p_borrow(NULL)) and call plsa_release_scratch first", c->P.cap / 1e9, bytes / 1e9);
    HIPCHK(c, hipStreamSynchronize(c->stream));
    CHK(ensure(c, c->P, (size_t)bytes));
    c->p_lent = true;
    c->p_shift = 0;
    c->p_valid = false;
    *device_ptr = c->P.p;
    return 0;
}

// Use `device_ptr` (memory of the same device that outlives every later call on this context; e.g. another context's
// plsa_p_reserve) for P(z|w,d) instead of an allocation of one's own; NULL returns to own allocations.  Contexts that
// share a buffer must not run materialising calls concurrently.
int plsa_p_borrow(plsa_ctx *c, void *device_ptr, int64_t bytes) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (!c->p_borrowed) release(c->P);
    c->P.p = device_ptr;
    c->P.cap = device_ptr ? (size_t)std::max<int64_t>(bytes, 0) : 0;
    c->p_borrowed = device_ptr != nullptr;
    c->p_valid = false;
    c->p_shift = 0;
    return 0;
}

int plsa_set_sample_weight(plsa_ctx *c, const float *sw) {
    HIPCHK(c, hipSetDevice(c->device));
    c->sw_resident = false;
    if (!sw) return 0;
    if (c->n <= 0) return fail(c, "plsa_set_sample_weight: no matrix uploaded");
    CHK(ensure(c, c->sw_res, sizeof(float) * (size_t)c->n));
    HIPCHK(c, hipMemcpyAsync(c->sw_res.p, sw, sizeof(float) * (size_t)c->n, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));          // `sw` may be freed on return
    c->sw_resident = true;
    c->sw_n = c->n;
    return 0;
}

int plsa_em_finish(plsa_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    CHK(run_v_normalise(c));
    c->cu ^= 1; c->cv ^= 1;
    return 0;                       // stream-ordered: no host synchronisation (readers synchronise)
}

int plsa_accumulator_device(plsa_ctx *c, void **ptr, int64_t *n_floats) {
    CHK(need_factors(c));
    if (ptr) *ptr = c->Vacc.p;
    if (n_floats) *n_floats = c->m * c->kp;
    return 0;
}

int plsa_accumulator_get(plsa_ctx *c, float *host) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    HIPCHK(c, hipMemcpyAsync(host, c->Vacc.p, sizeof(float) * (size_t)c->m * c->kp, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_accumulator_set(plsa_ctx *c, const float *host) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    HIPCHK(c, hipMemcpyAsync(c->Vacc.p, host, sizeof(float) * (size_t)c->m * c->kp, hipMemcpyHostToDevice, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- multi-GPU exchange over RCCL (xGMI): one communicator per context ------------------------------
int plsa_comm_unique_id(void *id128) {
    if (!id128) return fail(nullptr, "plsa_comm_unique_id: NULL");
    static_assert(sizeof(ncclUniqueId) == PLSA_COMM_ID_BYTES, "RCCL unique id size");
    ncclUniqueId id;
    ncclResult_t r = ncclGetUniqueId(&id);
    if (r != ncclSuccess) return fail(nullptr, "ncclGetUniqueId: %s", ncclGetErrorString(r));
    memcpy(id128, &id, sizeof id);
    return 0;
}

int plsa_comm_init(plsa_ctx *c, const void *id128, int32_t rank, int32_t world) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!id128 || world < 1 || rank < 0 || rank >= world) return fail(c, "plsa_comm_init: bad arguments");
    if (c->comm) return fail(c, "plsa_comm_init: this context already has a communicator");
    ncclUniqueId id;
    memcpy(&id, id128, sizeof id);
    NCCLCHK(c, ncclCommInitRank(&c->comm, world, id, rank));
    c->comm_rank = rank; c->comm_world = world;
    CHK(ensure(c, c->comm_small, 4096));
    return 0;
}

int plsa_comm_last_error(plsa_ctx *c, char *buf, int64_t cap) {
    if (!buf || cap <= 0) return 1;
    const char *msg = ncclGetLastError(c ? c->comm : nullptr);     // RCCL keeps one text per process; comm may be NULL
    snprintf(buf, (size_t)cap, "%s", msg ? msg : "");
    return 0;
}

int plsa_comm_destroy(plsa_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) {
        HIPCHK(c, hipStreamSynchronize(c->stream));
        HIPCHK(c, hipStreamSynchronize(c->stream2));
        NCCLCHK(c, ncclCommDestroy(c->comm));
        c->comm = nullptr;
    }
    c->comm_rank = 0; c->comm_world = 1;
    return 0;
}

int plsa_comm_info(plsa_ctx *c, int32_t *rank, int32_t *world) {
    if (rank) *rank = c->comm_rank;
    if (world) *world = c->comm_world;
    return 0;
}

int plsa_comm_barrier(plsa_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->comm) {
        HIPCHK(c, hipMemsetAsync(c->comm_small.p, 0, sizeof(int), c->stream));
        NCCLCHK(c, ncclAllReduce(c->comm_small.p, c->comm_small.p, 1, ncclInt32, ncclSum, c->comm, c->stream));
    }
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// ---- member stack: the np.vstack of enstop_.py:231 without leaving the GPU ----------------------------
// The topic matrices of the ensemble members a process fits are kept in one device block [slots][k][m]
// (plsa_copy_components_to_device writes a slot -- from this context or from another context of the same
// device: members of a small corpus are fitted by several contexts at once).
int plsa_stack_reserve(plsa_ctx *c, int64_t slots, int64_t m, int32_t k, void **base_device) {
    HIPCHK(c, hipSetDevice(c->device));
    if (slots < 1 || m < 1 || k < 1 || !base_device) return fail(c, "plsa_stack_reserve: bad arguments");
    CHK(ensure(c, c->comm_stack, sizeof(float) * (size_t)slots * (size_t)k * (size_t)m));
    *base_device = c->comm_stack.p;
    return 0;
}

// One exchange for the whole ensemble: slot s of every rank is all-gathered into [s][rank] (all slots in ONE
// grouped RCCL launch), so the gathered block is the stack in RUN order -- run r was fitted by rank r % world
// in slot r / world (enstop_amd/enstop_.py) -- and goes to a page-locked host buffer in one copy.  Without a
// communicator (one process) it is the device stack itself.  *host: [slots * world][k][m], valid until the next
// call on this context.
// dst != nullptr: the gathered stack goes straight into the caller's host array (one pass; a fresh NumPy array costs its
// page faults exactly once, a re-used one nothing: 195 MB in 3.7 ms against 17 ms through the page-locked buffer plus a
// NumPy copy).  dst == nullptr: into the context's page-locked buffer, *host_view receives its address.
static int allgather_stack_impl(plsa_ctx *c, int64_t slots, int64_t m, int32_t k, float *dst, float **host_view) {
    HIPCHK(c, hipSetDevice(c->device));
    if (slots < 1 || m < 1 || k < 1 || (!dst && !host_view)) return fail(c, "plsa_comm_allgather_stack: bad arguments");
    const size_t km = (size_t)k * (size_t)m, world = (size_t)c->comm_world;
    if (c->comm_stack.cap < sizeof(float) * (size_t)slots * km)
        return fail(c, "plsa_comm_allgather_stack: no stack of %lld slots reserved (plsa_stack_reserve)", (long long)slots);
    const size_t bytes = sizeof(float) * (size_t)slots * km * world;
    if (!dst && c->comm_host_cap < bytes) {
        if (c->comm_host) { HIPCHK(c, hipHostFree(c->comm_host)); c->comm_host = nullptr; c->comm_host_cap = 0; }
        HIPCHK(c, hipHostMalloc((void **)&c->comm_host, bytes, hipHostMallocDefault));
        c->comm_host_cap = bytes;
    }
    const float *src = c->comm_stack.as<float>();
    if (c->comm) {
        CHK(ensure(c, c->comm_recv, bytes));
        Scope s(c, "rccl_allgather_stack");
        NCCLCHK(c, ncclGroupStart());
        for (int64_t sl = 0; sl < slots; ++sl)
            NCCLCHK(c, ncclAllGather(c->comm_stack.as<float>() + (size_t)sl * km,
                                     c->comm_recv.as<float>() + (size_t)sl * world * km, km, ncclFloat, c->comm, c->stream));
        NCCLCHK(c, ncclGroupEnd());
        src = c->comm_recv.as<float>();
    }
    float *to = dst ? dst : c->comm_host;
    HIPCHK(c, hipMemcpyAsync(to, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    if (host_view) *host_view = to;
    return 0;
}

int plsa_comm_allgather_stack(plsa_ctx *c, int64_t slots, int64_t m, int32_t k, float **host) {
    return allgather_stack_impl(c, slots, m, k, nullptr, host);
}

int plsa_comm_allgather_stack_to(plsa_ctx *c, int64_t slots, int64_t m, int32_t k, float *dst) {
    if (!dst) return fail(c, "plsa_comm_allgather_stack_to: dst is NULL");
    return allgather_stack_impl(c, slots, m, k, dst, nullptr);
}

int plsa_comm_allgather_host(plsa_ctx *c, const void *send, int64_t bytes, void *recv) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!send || !recv || bytes <= 0) return fail(c, "plsa_comm_allgather_host: bad arguments");
    if (!c->comm) { memmove(recv, send, (size_t)bytes); return 0; }
    CHK(ensure(c, c->comm_send, (size_t)bytes));
    CHK(ensure(c, c->comm_recv, (size_t)bytes * (size_t)c->comm_world));
    HIPCHK(c, hipMemcpyAsync(c->comm_send.p, send, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, ncclAllGather(c->comm_send.p, c->comm_recv.p, (size_t)bytes, ncclUint8, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(recv, c->comm_recv.p, (size_t)bytes * (size_t)c->comm_world, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_comm_allreduce_f64(plsa_ctx *c, double *inout, int64_t count, int32_t op) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!inout || count <= 0 || (op != 0 && op != 1)) return fail(c, "plsa_comm_allreduce_f64: bad arguments");
    if (!c->comm) return 0;
    CHK(ensure(c, c->comm_send, sizeof(double) * (size_t)count));
    HIPCHK(c, hipMemcpyAsync(c->comm_send.p, inout, sizeof(double) * (size_t)count, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, ncclAllReduce(c->comm_send.p, c->comm_send.p, (size_t)count, ncclDouble, op == 0 ? ncclSum : ncclMax,
                             c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(inout, c->comm_send.p, sizeof(double) * (size_t)count, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_comm_broadcast_host(plsa_ctx *c, void *buf, int64_t bytes, int32_t root) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!buf || bytes <= 0 || root < 0 || root >= c->comm_world) return fail(c, "plsa_comm_broadcast_host: bad arguments");
    if (!c->comm) return 0;
    CHK(ensure(c, c->comm_send, (size_t)bytes));
    if (c->comm_rank == root)
        HIPCHK(c, hipMemcpyAsync(c->comm_send.p, buf, (size_t)bytes, hipMemcpyHostToDevice, c->stream));
    NCCLCHK(c, ncclBroadcast(c->comm_send.p, c->comm_send.p, (size_t)bytes, ncclUint8, root, c->comm, c->stream));
    HIPCHK(c, hipMemcpyAsync(buf, c->comm_send.p, (size_t)bytes, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

// in-place sum of the un-normalised P(w|z) accumulator over the communicator, enqueued on the context's
// stream (no host synchronisation): the exchange step between plsa_em_accumulate and plsa_em_finish
int plsa_allreduce_accumulator(plsa_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    CHK(need_factors(c));
    if (!c->comm) return 0;
    NCCLCHK(c, ncclAllReduce(c->Vacc.p, c->Vacc.p, (size_t)c->m * c->kp, ncclFloat, ncclSum, c->comm, c->stream));
    return 0;
}

int plsa_reference_chain_info(plsa_ctx *c, int64_t *slow_chunks, int64_t *chunks, int32_t *serial_now) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream2));
    if (c->ref_stats_pending) {
        c->ref_stats_pending = false;
        c->ref_slow_total += c->h_ref_stats[0]; c->ref_chunks_total += c->h_ref_stats[1];
        if (c->ref_chain_mode == 0 && c->h_ref_stats[1] > 0 && c->h_ref_stats[0] * 4 > c->h_ref_stats[1]) c->ref_pairs_off = true;
    }
    if (slow_chunks) *slow_chunks = (int64_t)c->ref_slow_total;
    if (chunks) *chunks = (int64_t)c->ref_chunks_total;
    if (serial_now) *serial_now = (c->ref_chain_mode == 2 || (c->ref_chain_mode == 0 && c->ref_pairs_off)) ? 1 : 0;
    return 0;
}

int plsa_placement_info(plsa_ctx *c, int32_t *candidates, double *best_gbps, double *worst_gbps) {
    if (candidates) *candidates = c->placement_tried;
    if (best_gbps) *best_gbps = c->placement_gbps[0];
    if (worst_gbps) *worst_gbps = c->placement_gbps[1];
    return 0;
}

int plsa_schedule_info(plsa_ctx *c, int32_t *xcd_lo, double *xcd_end_us, int32_t *timed_launches, int32_t *item_len,
                       int64_t *n_items) {
    if (xcd_lo) for (int x = 0; x <= 8; ++x) xcd_lo[x] = c->bal_valid ? c->bal_lo[x] : 0;
    if (xcd_end_us) for (int x = 0; x < 8; ++x) xcd_end_us[x] = c->bal_launches > 0 ? c->bal_end_us[x] : 0.0;
    if (timed_launches) *timed_launches = c->bal_launches;
    if (item_len) *item_len = c->csc_valid ? c->seg : 0;
    if (n_items) *n_items = c->csc_valid ? c->n_items : 0;
    return 0;
}

int plsa_release_scratch(plsa_ctx *c) {
    HIPCHK(c, hipSetDevice(c->device));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    // a BORROWED P(z|w,d) stays borrowed (nothing of this context's own would be freed, and silently dropping the loan made
    // the next materialising call allocate a private full-size array); a LENT one is freed: the caller ends the loans first
    if (!c->p_borrowed) release(c->P);
    c->p_lent = false;
    release(c->ref_terms); release(c->ref_csum); release(c->ref_pairs); release(c->ref_exps); release(c->ref_ll_neg); release(c->ref_heavy); release(c->ref_pairs2); release(c->ref_exps2); release(c->ref_tsum);
    release(c->partial); release(c->tmp0); release(c->tmp1); release(c->tmp2); release(c->cubtmp);
    release(c->mt_words); release(c->mt_state); release(c->mt_fin); release(c->mt_poly); release(c->mt_seq);
    // member stack + gather buffers of the ensemble exchange (16 runs x 64 topics x 100 k words = 0.4 GB): re-created
    // by the next plsa_stack_reserve / plsa_comm_allgather_stack
    // The page-locked landing buffer of plsa_comm_allgather_stack (c->comm_host) is NOT freed here: the caller may still
    // hold the pointer that call returned (a NumPy view in enstop_amd: gather_stack(view=True)); it lives until the next
    // gather that needs a larger one, or plsa_destroy.
    release(c->comm_stack); release(c->comm_recv); release(c->comm_send);
    c->p_valid = false;
    c->p_shift = 0;
    return 0;
}

int plsa_timing_enable(plsa_ctx *c, int32_t on) {
    if (!on) CHK(timing_flush(c));
    c->timing = on != 0;
    return 0;
}

int plsa_timing_reset(plsa_ctx *c) {
    CHK(timing_flush(c));
    std::fill(c->acc_ms.begin(), c->acc_ms.end(), 0.0);
    std::fill(c->acc_n.begin(), c->acc_n.end(), 0);
    return 0;
}

int plsa_timing_get(plsa_ctx *c, const char *prefix, double *total_ms, int64_t *launches) {
    CHK(timing_flush(c));
    double t = 0.0;
    i64 n = 0;
    const size_t pl = strlen(prefix);
    for (size_t i = 0; i < c->names.size(); ++i)
        if (c->names[i].compare(0, pl, prefix) == 0) { t += c->acc_ms[i]; n += c->acc_n[i]; }
    if (total_ms) *total_ms = t;
    if (launches) *launches = n;
    return 0;
}

int plsa_timing_report(plsa_ctx *c, char *buf, int64_t cap) {
    CHK(timing_flush(c));
    std::string s;
    char line[256];
    for (size_t i = 0; i < c->names.size(); ++i) {
        if (!c->acc_n[i]) continue;
        snprintf(line, sizeof line, "%s %lld %.6f\n", c->names[i].c_str(), (long long)c->acc_n[i], c->acc_ms[i]);
        s += line;
    }
    if (cap > 0) { strncpy(buf, s.c_str(), (size_t)cap - 1); buf[cap - 1] = 0; }
    return 0;
}

int plsa_measure_stream_bandwidth(plsa_ctx *c, int64_t bytes, int32_t kind, int32_t reps, double *gbps) {
    HIPCHK(c, hipSetDevice(c->device));
    if (bytes < (1 << 20) || reps < 1 || kind < 0 || kind > 6) return fail(c, "plsa_measure_stream_bandwidth: bad arguments");
    DevBuf a, b;
    const i64 n4 = bytes / 16;
    int rc = ensure(c, a, (size_t)n4 * 16);
    if (!rc && kind == 2) rc = ensure(c, b, (size_t)n4 * 16);
    if (rc) { release(a); release(b); return rc; }
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int grid = grid_for(c, n4, 256);
    for (int r = -1; r < reps; ++r) {          // r == -1: untimed warm-up (page faults, clocks)
        if (r == 0) (void)hipEventRecord(e0, c->stream);
        if (kind == 0) hipLaunchKernelGGL((plsa::k_probe_fill<true>), dim3(grid), dim3(256), 0, c->stream, a.as<float>(), n4);
        else if (kind == 1) hipLaunchKernelGGL((plsa::k_probe_fill<false>), dim3(grid), dim3(256), 0, c->stream, a.as<float>(), n4);
        else if (kind >= 4) hipLaunchKernelGGL(plsa::k_probe_fill_tiled, dim3(grid_for(c, n4, 256 * (kind == 4 ? 16 : kind == 5 ? 4 : 64))), dim3(256), 0, c->stream, a.as<float>(), n4, kind == 4 ? 16 : kind == 5 ? 4 : 64);
        else if (kind == 3) hipLaunchKernelGGL(plsa::k_probe_read, dim3(grid), dim3(256), 0, c->stream, a.as<float>(), a.as<float>(), n4);
        else hipLaunchKernelGGL(plsa::k_probe_copy, dim3(grid), dim3(256), 0, c->stream, a.as<float>(), b.as<float>(), n4);
    }
    (void)hipEventRecord(e1, c->stream);
    hipError_t e = hipStreamSynchronize(c->stream);
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    release(a); release(b);
    if (e != hipSuccess) return fail(c, "stream probe failed: %s", hipGetErrorString(e));
    const double moved = (double)n4 * 16.0 * (kind == 2 ? 2.0 : 1.0) * reps;
    *gbps = moved / 1e9 / (ms / 1e3);
    return 0;
}

// enstop/utils.py:22-41 with axis=1 (float64, sequential marginal, guarded division)
void plsa_host_normalize_rows(double *a, int64_t rows, int64_t cols) {
    for (int64_t i = 0; i < rows; ++i) {
        double *r = a + i * cols;
        double marginal = 0.0;
        for (int64_t j = 0; j < cols; ++j) marginal += r[j];
        if (marginal > 0.0)
            for (int64_t j = 0; j < cols; ++j) r[j] /= marginal;
    }
}

// enstop/enstop_.py:258-266 (pairwise umap.distances.hellinger over the stacked topics) on the device
int plsa_all_pairs_hellinger(plsa_ctx *c, const float *topics, int64_t t, int64_t m, double *D) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!topics || !D || t <= 0 || m <= 0 || t > 65536) return fail(c, "plsa_all_pairs_hellinger: bad arguments");
    const int nt = (int)((t + plsa::HELL_TILE - 1) / plsa::HELL_TILE);
    std::vector<int> ti, tj;
    for (int i = 0; i < nt; ++i) for (int j = i; j < nt; ++j) { ti.push_back(i); tj.push_back(j); }
    // enough vocabulary slices to fill the chip, each a multiple of the staging step
    int slices = (int)std::max<i64>(1, std::min<i64>(64, (4 * (i64)c->prop.multiProcessorCount + (i64)ti.size() - 1) / (i64)ti.size()));
    while (slices > 1 && (double)slices * (double)t * (double)t * 8.0 > 4e9) --slices;   // bound the partial buffer
    i64 slice = ((m + slices - 1) / slices + plsa::HELL_KSTEP - 1) / plsa::HELL_KSTEP * plsa::HELL_KSTEP;
    slices = (int)((m + slice - 1) / slice);
    DevBuf R, l1, part, dD, dt;
    int rc = ensure(c, R, sizeof(float) * (size_t)t * m);
    if (!rc) rc = ensure(c, l1, sizeof(double) * (size_t)t);
    if (!rc) rc = ensure(c, part, sizeof(double) * (size_t)slices * t * t);
    if (!rc) rc = ensure(c, dD, sizeof(double) * (size_t)t * t);
    if (!rc) rc = ensure(c, dt, sizeof(int) * 2 * ti.size());
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(R.p, topics, sizeof(float) * (size_t)t * m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dt.p, ti.data(), sizeof(int) * ti.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dt.as<int>() + ti.size(), tj.data(), sizeof(int) * tj.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(plsa::k_hell_prepare, dim3((unsigned)t), dim3(256), 0, c->stream, R.as<float>(), (int)t, (i64)m, l1.as<double>());
            { Scope s(c, "k_hell_gram");
              hipLaunchKernelGGL(plsa::k_hell_gram, dim3((unsigned)ti.size(), (unsigned)slices), dim3(256), 0, c->stream,
                                 R.as<float>(), (int)t, (i64)m, slice, dt.as<int>(), dt.as<int>() + ti.size(), part.as<double>()); }
            hipLaunchKernelGGL(plsa::k_hell_finish, dim3((unsigned)((t * t + 255) / 256)), dim3(256), 0, c->stream,
                               part.as<double>(), slices, (int)t, l1.as<double>(), dD.as<double>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(D, dD.p, sizeof(double) * (size_t)t * t, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    release(R); release(l1); release(part); release(dD); release(dt);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, "plsa_all_pairs_hellinger: %s", hipGetErrorString(e));
    return 0;
}

// enstop/enstop_.py:244-253 (all_pairs_kl_divergence over the stacked topics) on the device
int plsa_all_pairs_kl(plsa_ctx *c, const float *topics, int64_t t, int64_t m, double *D) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!topics || !D || t <= 0 || m <= 0 || t > 65536) return fail(c, "plsa_all_pairs_kl: bad arguments");
    const int nt = (int)((t + plsa::HELL_TILE - 1) / plsa::HELL_TILE);
    const i64 tiles = (i64)nt * nt;
    int slices = (int)std::max<i64>(1, std::min<i64>(64, (4 * (i64)c->prop.multiProcessorCount + tiles - 1) / tiles));
    while (slices > 1 && (double)slices * (double)t * (double)t * 8.0 > 4e9) --slices;
    i64 slice = ((m + slices - 1) / slices + plsa::HELL_KSTEP - 1) / plsa::HELL_KSTEP * plsa::HELL_KSTEP;
    slices = (int)((m + slice - 1) / slice);
    DevBuf T, part, dD;
    int rc = ensure(c, T, sizeof(float) * (size_t)t * m);
    if (!rc) rc = ensure(c, part, sizeof(double) * (size_t)slices * t * t);
    if (!rc) rc = ensure(c, dD, sizeof(double) * (size_t)t * t);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(T.p, topics, sizeof(float) * (size_t)t * m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            { Scope s(c, "k_kl_gram");
              hipLaunchKernelGGL(plsa::k_kl_gram, dim3((unsigned)tiles, (unsigned)slices), dim3(256), 0, c->stream,
                                 T.as<float>(), (int)t, (i64)m, slice, part.as<double>()); }
            hipLaunchKernelGGL(plsa::k_sum_slices, dim3((unsigned)((t * t + 255) / 256)), dim3(256), 0, c->stream,
                               part.as<double>(), slices, (i64)t * t, dD.as<double>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(D, dD.p, sizeof(double) * (size_t)t * t, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    release(T); release(part); release(dD);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, "plsa_all_pairs_kl: %s", hipGetErrorString(e));
    return 0;
}

// enstop/enstop_.py:299-308, 340-345 (weights == NULL) and 385-393 (membership-strength weights)
int plsa_cluster_representatives(plsa_ctx *c, const float *topics, int64_t t, int64_t m, const int32_t *labels,
                                 const double *weights, int32_t n_clusters, float *out) {
    HIPCHK(c, hipSetDevice(c->device));
    if (!topics || !labels || !out || t <= 0 || m <= 0 || n_clusters < 0)
        return fail(c, "plsa_cluster_representatives: bad arguments");
    if (n_clusters == 0) return 0;
    // members of each cluster in row order (labels < 0 are noise; labels >= n_clusters are an error)
    std::vector<int> first((size_t)n_clusters + 1, 0), members;
    for (int64_t i = 0; i < t; ++i) {
        if (labels[i] >= n_clusters) return fail(c, "plsa_cluster_representatives: label %d >= n_clusters %d", labels[i], n_clusters);
        if (labels[i] >= 0) first[(size_t)labels[i] + 1]++;
    }
    for (int cl = 0; cl < n_clusters; ++cl) first[(size_t)cl + 1] += first[(size_t)cl];
    members.resize((size_t)first[(size_t)n_clusters] + 1);
    {
        std::vector<int> fill(first.begin(), first.end() - 1);
        for (int64_t i = 0; i < t; ++i) if (labels[i] >= 0) members[(size_t)fill[(size_t)labels[i]]++] = (int)i;
    }
    // enstop_.py:385-393 np.average raises on an all-zero weight vector; callers handle that case
    const int nb = (int)((m + 255) / 256);
    DevBuf T, dfirst, dmem, dw, rep, bs, dout;
    int rc = ensure(c, T, sizeof(float) * (size_t)t * m);
    if (!rc) rc = ensure(c, dfirst, sizeof(int) * first.size());
    if (!rc) rc = ensure(c, dmem, sizeof(int) * members.size());
    if (!rc && weights) rc = ensure(c, dw, sizeof(double) * (size_t)t);
    if (!rc) rc = ensure(c, rep, sizeof(double) * (size_t)n_clusters * m);
    if (!rc) rc = ensure(c, bs, sizeof(double) * (size_t)n_clusters * nb);
    if (!rc) rc = ensure(c, dout, sizeof(float) * (size_t)n_clusters * m);
    hipError_t e = hipSuccess;
    if (!rc) {
        e = hipMemcpyAsync(T.p, topics, sizeof(float) * (size_t)t * m, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dfirst.p, first.data(), sizeof(int) * first.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) e = hipMemcpyAsync(dmem.p, members.data(), sizeof(int) * members.size(), hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess && weights) e = hipMemcpyAsync(dw.p, weights, sizeof(double) * (size_t)t, hipMemcpyHostToDevice, c->stream);
        if (e == hipSuccess) {
            hipLaunchKernelGGL(plsa::k_rep_accumulate, dim3((unsigned)nb, (unsigned)n_clusters), dim3(256), 0, c->stream,
                               T.as<float>(), (i64)m, dfirst.as<int>(), dmem.as<int>(), weights ? dw.as<double>() : nullptr,
                               rep.as<double>(), bs.as<double>());
            hipLaunchKernelGGL(plsa::k_rep_normalise, dim3((unsigned)nb, (unsigned)n_clusters), dim3(256), 0, c->stream,
                               rep.as<double>(), (i64)m, bs.as<double>(), nb, dout.as<float>());
            e = hipGetLastError();
        }
        if (e == hipSuccess) e = hipMemcpyAsync(out, dout.p, sizeof(float) * (size_t)n_clusters * m, hipMemcpyDeviceToHost, c->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    }
    release(T); release(dfirst); release(dmem); release(dw); release(rep); release(bs); release(dout);
    if (rc) return rc;
    if (e != hipSuccess) return fail(c, "plsa_cluster_representatives: %s", hipGetErrorString(e));
    return 0;
}

int plsa_host_mt19937_jump(uint32_t *key, int32_t log2_blocks) {
    if (!key || log2_blocks < 0 || log2_blocks > 40) return 1;
    uint32_t g[624];
    if (!mtjump::block_jump_polynomial(log2_blocks, g)) return 1;
    mtjump::apply_jump(key, g);
    return 0;
}

// Synthetic corpus in HBM (see plsa_synth.hpp).  The mean token count per document is calibrated
// by a short secant iteration so that the number of DISTINCT (doc, word) pairs lands within 0.5 %
// of nnz_target; the final matrix depends only on the arguments.
static int generate_synthetic_impl(plsa_ctx *c, int64_t n, int64_t m, int64_t nnz_target, double zipf_s,
                                   uint64_t seed, int k0, double alpha, double background, int64_t *nnz_out) {
    HIPCHK(c, hipSetDevice(c->device));
    if (n <= 0 || m <= 1 || nnz_target < n || n >= INT32_MAX || m >= INT32_MAX)
        return fail(c, "plsa_generate_synthetic: bad arguments (need nnz_target >= n, m > 1)");
    if (k0 < 0 || k0 > 256 || (k0 > 0 && (!(alpha > 0.0) || background < 0.0 || background > 1.0)))
        return fail(c, "plsa_generate_synthetic_topics: need 1 <= k0 <= 256, alpha > 0, 0 <= background <= 1");
    // Zipf CDF over ranks (host, float64) and an affine permutation rank -> word id
    std::vector<double> cdf((size_t)m);
    double tot = 0.0;
    for (int64_t r = 0; r < m; ++r) { tot += std::pow((double)(r + 1), -zipf_s); cdf[(size_t)r] = tot; }
    for (int64_t r = 0; r < m; ++r) cdf[(size_t)r] /= tot;
    cdf[(size_t)m - 1] = 1.0;
    uint64_t a = (uint64_t)((double)m * 0.6180339887498949) | 1ull;
    auto gcd = [](uint64_t x, uint64_t y) { while (y) { uint64_t t = x % y; x = y; y = t; } return x; };
    while (gcd(a, (uint64_t)m) != 1) a += 2;
    const uint64_t b = plsa::mix64(seed ^ 0xABCDEFull) % (uint64_t)m;
    // topical corpus: one affine ranking per latent topic (odd multipliers spread by the hash, made coprime with m),
    // the shared ranking above in slot k0
    std::vector<uint64_t> perm;
    if (k0 > 0) {
        perm.resize(2 * (size_t)(k0 + 1));
        for (int t = 0; t < k0; ++t) {
            uint64_t at = (plsa::mix64(seed ^ plsa::mix64(0x70C1Cull + (uint64_t)t)) % (uint64_t)m) | 1ull;
            while (gcd(at, (uint64_t)m) != 1) at += 2;
            perm[2 * (size_t)t] = at % (uint64_t)m ? at % (uint64_t)m : 1;
            perm[2 * (size_t)t + 1] = plsa::mix64(seed ^ plsa::mix64(0xB0FF5E7ull + (uint64_t)t)) % (uint64_t)m;
        }
        perm[2 * (size_t)k0] = a; perm[2 * (size_t)k0 + 1] = b;
    }
    DevBuf d_cdf, d_tok, d_ptr, d_keys, d_keys2, d_flag, d_pos, d_perm;
    auto cleanup = [&]() { release(d_cdf); release(d_tok); release(d_ptr); release(d_keys);
                           release(d_keys2); release(d_flag); release(d_pos); release(d_perm); };
#define SYN(expr) do { int r_ = (expr); if (r_) { cleanup(); return r_; } } while (0)
#define SYNHIP(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { cleanup(); \
        return fail(c, "%s failed: %s", #expr, hipGetErrorString(e_)); } } while (0)
    SYN(ensure(c, d_cdf, sizeof(double) * (size_t)m));
    SYNHIP(hipMemcpyAsync(d_cdf.p, cdf.data(), sizeof(double) * (size_t)m, hipMemcpyHostToDevice, c->stream));
    SYN(ensure(c, d_tok, sizeof(int) * (size_t)(n + 1)));
    SYN(ensure(c, d_ptr, sizeof(i64) * (size_t)(n + 1)));
    if (k0 > 0) {
        SYN(ensure(c, d_perm, sizeof(uint64_t) * perm.size()));
        SYNHIP(hipMemcpyAsync(d_perm.p, perm.data(), sizeof(uint64_t) * perm.size(), hipMemcpyHostToDevice, c->stream));
    }
    const double sigma = 0.6;
    double mean_tokens = 1.4 * (double)nnz_target / (double)n;
    i64 T = 0;
    int nnz = 0;
    int dbits = 1;
    while (((i64)1 << dbits) < n) ++dbits;
    for (int round = 0; round < 8; ++round) {
        const double mu = std::log(mean_tokens) - 0.5 * sigma * sigma;
        hipLaunchKernelGGL(plsa::k_synth_doc_tokens, dim3((unsigned)((n + 256) / 256)), dim3(256), 0, c->stream,
                           (int)n, mu, sigma, (int)std::min<i64>(m * 4, 1 << 20), seed, d_tok.as<int>());
        {
            auto in64 = hipcub::TransformInputIterator<i64, hipcub::CastOp<i64>, int *>(d_tok.as<int>(), hipcub::CastOp<i64>());
            size_t bytes = 0;
            SYNHIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, in64, d_ptr.as<i64>(), (int)(n + 1), c->stream));
            SYN(ensure(c, c->cubtmp, bytes));
            SYNHIP(hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, bytes, in64, d_ptr.as<i64>(), (int)(n + 1), c->stream));
        }
        SYNHIP(hipMemcpyAsync(&T, d_ptr.as<i64>() + n, sizeof(i64), hipMemcpyDeviceToHost, c->stream));
        SYNHIP(hipStreamSynchronize(c->stream));
        if (T >= INT32_MAX) { cleanup(); return fail(c, "plsa_generate_synthetic: %lld tokens >= 2^31", (long long)T); }
        SYN(ensure(c, d_keys, sizeof(unsigned long long) * (size_t)T));
        SYN(ensure(c, d_keys2, sizeof(unsigned long long) * (size_t)T));
        SYN(ensure(c, d_flag, sizeof(int) * (size_t)T));
        SYN(ensure(c, d_pos, sizeof(int) * (size_t)T));
        if (k0 > 0)
            hipLaunchKernelGGL(plsa::k_synth_draw_topics, dim3(grid_for(c, n, 4)), dim3(256), 0, c->stream, (int)n, (int)m,
                               d_ptr.as<i64>(), d_cdf.as<double>(), d_perm.as<uint64_t>(), k0, alpha, background, seed,
                               d_keys.as<unsigned long long>());
        else
            hipLaunchKernelGGL(plsa::k_synth_draw, dim3(grid_for(c, n, 4)), dim3(256), 0, c->stream, (int)n, (int)m,
                               d_ptr.as<i64>(), d_cdf.as<double>(), a, b, seed, d_keys.as<unsigned long long>());
        {
            size_t bytes = 0;
            SYNHIP(hipcub::DeviceRadixSort::SortKeys(nullptr, bytes, d_keys.as<unsigned long long>(),
                                                     d_keys2.as<unsigned long long>(), T, 0, 32 + dbits, c->stream));
            SYN(ensure(c, c->cubtmp, bytes));
            SYNHIP(hipcub::DeviceRadixSort::SortKeys(c->cubtmp.p, bytes, d_keys.as<unsigned long long>(),
                                                     d_keys2.as<unsigned long long>(), T, 0, 32 + dbits, c->stream));
        }
        hipLaunchKernelGGL(plsa::k_synth_heads, dim3(grid_for(c, T, 256)), dim3(256), 0, c->stream,
                           d_keys2.as<unsigned long long>(), T, d_flag.as<int>());
        {
            size_t bytes = 0;
            SYNHIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_flag.as<int>(), d_pos.as<int>(), (int)T, c->stream));
            SYN(ensure(c, c->cubtmp, bytes));
            SYNHIP(hipcub::DeviceScan::ExclusiveSum(c->cubtmp.p, bytes, d_flag.as<int>(), d_pos.as<int>(), (int)T, c->stream));
        }
        int last_pos = 0, last_flag = 0;
        SYNHIP(hipMemcpyAsync(&last_pos, d_pos.as<int>() + (T - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        SYNHIP(hipMemcpyAsync(&last_flag, d_flag.as<int>() + (T - 1), sizeof(int), hipMemcpyDeviceToHost, c->stream));
        SYNHIP(hipStreamSynchronize(c->stream));
        nnz = last_pos + last_flag;
        const double rel = ((double)nnz - (double)nnz_target) / (double)nnz_target;
        if (std::fabs(rel) < 0.005 || round == 7) break;
        // distinct pairs grow sub-linearly in tokens: damped multiplicative correction
        mean_tokens *= std::pow((double)nnz_target / (double)nnz, 1.25);
    }
    SYN(ensure(c, c->b_indptr, sizeof(int) * (size_t)(n + 1)));
    SYN(ensure(c, c->b_col, sizeof(int) * (size_t)nnz));
    SYN(ensure(c, c->b_val, sizeof(float) * (size_t)nnz));
    hipLaunchKernelGGL(plsa::k_synth_emit, dim3(grid_for(c, T, 256)), dim3(256), 0, c->stream,
                       d_keys2.as<unsigned long long>(), T, d_flag.as<int>(), d_pos.as<int>(), (int)n, nnz,
                       c->b_indptr.as<int>(), c->b_col.as<int>(), c->b_val.as<float>());
    SYN(launch_check(c, "k_synth_emit"));
    SYNHIP(hipStreamSynchronize(c->stream));
#undef SYN
#undef SYNHIP
    cleanup();
    c->bn = n; c->bm = m; c->bnnz = nnz;
    c->active_is_base = true;
    set_active_pointers(c);
    c->syn_n = k0 > 0 ? n : 0; c->syn_k0 = k0; c->syn_alpha = alpha; c->syn_seed = seed;
    if (nnz_out) *nnz_out = nnz;
    return 0;
}

int plsa_synthetic_dominant_topics(plsa_ctx *c, int32_t *out /*[n] host*/) {
    HIPCHK(c, hipSetDevice(c->device));
    if (c->syn_n <= 0 || c->syn_n != c->bn) return fail(c, "plsa_synthetic_dominant_topics: the base corpus is not a topical synthetic corpus");
    if (!out) return fail(c, "plsa_synthetic_dominant_topics: out is NULL");
    CHK(ensure(c, c->tmp1, sizeof(int) * (size_t)c->syn_n));
    hipLaunchKernelGGL(plsa::k_synth_dominant_topic, dim3((unsigned)((c->syn_n + 255) / 256)), dim3(256), 0, c->stream,
                       (int)c->syn_n, c->syn_k0, c->syn_alpha, c->syn_seed, c->tmp1.as<int>());
    CHK(launch_check(c, "k_synth_dominant_topic"));
    HIPCHK(c, hipMemcpyAsync(out, c->tmp1.p, sizeof(int) * (size_t)c->syn_n, hipMemcpyDeviceToHost, c->stream));
    HIPCHK(c, hipStreamSynchronize(c->stream));
    return 0;
}

int plsa_generate_synthetic(plsa_ctx *c, int64_t n, int64_t m, int64_t nnz_target, double zipf_s,
                            uint64_t seed, int64_t *nnz_out) {
    return generate_synthetic_impl(c, n, m, nnz_target, zipf_s, seed, 0, 0.0, 0.0, nnz_out);
}

int plsa_generate_synthetic_topics(plsa_ctx *c, int64_t n, int64_t m, int64_t nnz_target, double zipf_s,
                                   uint64_t seed, int32_t k0, double alpha, double background, int64_t *nnz_out) {
    if (k0 < 1) return fail(c, "plsa_generate_synthetic_topics: need 1 <= k0 <= 256");
    return generate_synthetic_impl(c, n, m, nnz_target, zipf_s, seed, k0, alpha, background, nnz_out);
}

}  // extern "C"
