// plsa_kernels.hpp -- gfx950 (CDNA4, wave64) device kernels of the pLSA EM engine.
//
// Work decomposition shared by every kernel: a "group" of LPN adjacent lanes (LPN = 1..64, power
// of two) handles one non-zero (or one document row / one vocabulary column item); each lane owns
// CH float4 chunks of the k-vector, chunk c = lane_in_group + LPN*j.  k is padded to
// kp = 4*ceil(k/4) in every device layout (U [n,kp], Vt [m,kp] word-major, P [nnz,kp]); pad entries
// are zero and can never pass the `> thresh` test, so they do not perturb norms.  For k = 64:
// LPN = 16, CH = 1 -> a wave covers 4 non-zeros per step and every gather / store is one 16-byte
// access per lane, 256 contiguous bytes per non-zero.  FULL (kp == 4*LPN*CH, i.e. k = 4,8,..,256
// and 512, 1024) makes kp a compile-time constant and removes every lane predicate.
//
// Latency hiding: each wave keeps UNR non-zeros' factor rows in flight (all gathers of a batch are
// issued before the first use) and prefetches the next batch of (index, count) pairs.  Occupancy as
// hipcc reports it (enstop_amd/kernel_resources.json, written at build time; k = 64): document pass
// 60 VGPR = 8 waves per SIMD, column pass 90 VGPR = 5 waves, materialising E-step (document-owned) 124
// VGPR = 4 waves.  Round 4 measured that neither more waves (5 -> 8) nor more rows in flight (4 -> 16)
// moves the column pass (profiles/r04_column_pass_bound.md): it sits at its L2-miss service bound.
//
// Reference statements implemented here (paths relative to the reference root):
//   E-step            enstop/plsa.py:91-105      M-step scatter    enstop/plsa.py:182-194, 287-300
//   M-step normalise  enstop/plsa.py:196-202     log-likelihood    enstop/plsa.py:375-384
//   refit M-step      enstop/plsa.py:801-814
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#ifndef PLSA_UNR
#define PLSA_UNR 4      // non-zeros whose gathers are in flight together, per group
#endif
#ifndef PLSA_NT_STREAMS
#define PLSA_NT_STREAMS 0   // non-temporal loads for read-once streams: measured neutral on the fused
                            // passes and -14 % on the E-step (U rows are re-used from L1 by ~100 nnz)
#endif
#ifndef PLSA_UNR_E
#define PLSA_UNR_E 16   // E-step: float4 gathers per lane and burst (divided by the chunks per lane)
#endif
#ifndef PLSA_WAVES
#define PLSA_WAVES 1    // min waves per SIMD requested from the register allocator for the hot kernels
#endif
#ifndef PLSA_UNR_COL
#define PLSA_UNR_COL 8  // same for the column pass (its U gathers miss L2 more often: measured best)
#endif
#ifndef PLSA_WAVES_COL
#define PLSA_WAVES_COL PLSA_WAVES   // min waves per SIMD for the column pass alone (A/B knob, tools/kernel_resources.py)
#endif

namespace plsa {

typedef long long i64;

// ------------------------------------------------------------------------------------------------
// cross-lane sums inside a group of LPN lanes: DPP butterflies (no LDS traffic) up to 16 lanes,
// ds_bpermute-based xor shuffles across the 16-lane DPP rows.
// ------------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xF, 0xF, true));
}

template <int LPN>
__device__ __forceinline__ float group_sum(float v) {
    if (LPN >= 2) v += dpp_mov<0xB1>(v);    // quad_perm [1,0,3,2]  : lane ^ 1
    if (LPN >= 4) v += dpp_mov<0x4E>(v);    // quad_perm [2,3,0,1]  : lane ^ 2
    if (LPN >= 8) v += dpp_mov<0x141>(v);   // row_half_mirror      : other quad of the 8
    if (LPN >= 16) v += dpp_mov<0x140>(v);  // row_mirror           : other half of the 16
    if (LPN >= 32) v += __shfl_xor(v, 16, 64);
    if (LPN >= 64) v += __shfl_xor(v, 32, 64);
    return v;
}

__device__ __forceinline__ float hsum(const float4 &a) { return (a.x + a.y) + (a.z + a.w); }
__device__ __forceinline__ float4 zero4() { return make_float4(0.f, 0.f, 0.f, 0.f); }

__device__ __forceinline__ float4 ld4(const float *p) { return *reinterpret_cast<const float4 *>(p); }
__device__ __forceinline__ void st4(float *p, const float4 &v) { *reinterpret_cast<float4 *>(p) = v; }
typedef float f4v __attribute__((ext_vector_type(4)));
// P stores are non-temporal.  sc1 / sc0 sc1 (write-through, line dropped from L2) and sc1 nt were
// measured 4-9 % slower for the E-step on MI355X; plain stores 5 % slower (they evict the factor rows).
__device__ __forceinline__ void st4_nt(float *p, const float4 &v) {
    f4v t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f4v *>(p));  // global_store_dwordx4 ... nt
}
__device__ __forceinline__ float4 ld4_nt(const float *p) {
    const f4v t = __builtin_nontemporal_load(reinterpret_cast<const f4v *>(p));   // global_load_dwordx4 ... nt
    return make_float4(t.x, t.y, t.z, t.w);
}

__device__ __forceinline__ int ldi(const int *p) { return PLSA_NT_STREAMS ? __builtin_nontemporal_load(p) : *p; }
__device__ __forceinline__ float ldf(const float *p) { return PLSA_NT_STREAMS ? __builtin_nontemporal_load(p) : *p; }

// Shape of the lane decomposition.  When FULL, kp is the compile-time constant 4*LPN*CH.  WIDE: the gathered factor
// table may reach 4 GB, its rows are addressed with 64-bit arithmetic (gather_row below; the host picks it per launch).
template <int LPN_, int CH_, bool FULL_, bool WIDE_ = false>
struct Shape {
    static constexpr int LPN = LPN_, CH = CH_;
    static constexpr bool FULL = FULL_, WIDE = WIDE_;
    // document-owned kernels: rows in flight per group.  Narrow groups (k <= 32: 8 lanes, an index load covers 8 entries)
    // take the whole index batch in ONE round of gathers -- the pass is bound by the serial chain of its documents there
    // (config 2: 4730 -> 5470 iterations/s, config 1: 9490 -> 10060); 16-lane groups stay at PLSA_UNR (8 rows cost
    // registers and 1 % at config 3)
    static constexpr int UNR = (LPN_ <= 8 && CH_ == 1) ? LPN_ : ((LPN_ < PLSA_UNR) ? LPN_ : PLSA_UNR);
    static constexpr int UNR_COL = (LPN_ < PLSA_UNR_COL) ? LPN_ : PLSA_UNR_COL;
    // E-step: gathers per burst, ideally the whole index batch (LPN entries) so that a wave alternates
    // between one long run of loads and one long run of stores (measured: 4 -> 5.81 ms, 16 -> 5.43 ms at k = 64)
    static constexpr int UNR_EF = (LPN_ < 4) ? LPN_ : 4;   // flat (per-non-zero) E-step kernel
    static constexpr int UNR_E_CAP = (PLSA_UNR_E / CH_ > 0) ? PLSA_UNR_E / CH_ : 1;
    static constexpr int UNR_E = (LPN_ < UNR_E_CAP) ? LPN_ : UNR_E_CAP;
    __device__ static __forceinline__ int kp(int kp_rt) { return FULL_ ? 4 * LPN_ * CH_ : kp_rt; }
    // offset of chunk j for lane li, and whether it lies inside the row
    __device__ static __forceinline__ int c4(int li, int j) { return 4 * (li + LPN_ * j); }
    __device__ static __forceinline__ bool ok(int li, int j, int kp) { return FULL_ || c4(li, j) < kp; }
};

// gather the lane's chunks of one factor row WITHOUT a branch: out-of-row chunks read offset 0
// (a valid address) -- ZERO_INVALID then forces them to zero (needed for one operand only:
// the product with a zeroed chunk is 0 and fails the threshold test).
// STREAM marks rows that are read once per launch (the owner's own row, the index streams): they
// are loaded non-temporally so that the 4 MB per-XCD L2 keeps the rows that ARE re-used (the
// gathered factor table).
template <class S, bool ZERO_INVALID, bool STREAM = false>
__device__ __forceinline__ void load_row(const float *row, int li, int kp, float4 (&out)[S::CH]) {
#pragma unroll
    for (int j = 0; j < S::CH; ++j) {
        const bool ok = S::ok(li, j, kp);
        const float *p = row + (ok ? S::c4(li, j) : 0);
        const float4 v = STREAM ? ld4_nt(p) : ld4(p);
        out[j] = (ZERO_INVALID && !ok) ? zero4() : v;
    }
}

// Row `row` of a factor table that is gathered by index (P(w|z) rows by word in the document pass, P(z|d) rows by
// document in the column pass).  Tables below 4 GB -- every BASELINE configuration -- are addressed by a 32-bit BYTE offset
// from the wave-uniform table base (global_load_dwordx4 v, v_off, s[base:base+1]): one VALU instruction per gather
// (v_lshl_or_b32) where the 64-bit form needs three (sign extension, 64-bit shift, 64-bit add): 16 of ~250 VALU
// instructions per batch of 8 entries at k = 32 (config 2: +4 %).  S::WIDE keeps the 64-bit form for larger tables.
template <class S>
__device__ __forceinline__ void gather_row(const float *table, int row, int li, int kp, float4 (&out)[S::CH]) {
    if constexpr (S::WIDE) {
        load_row<S, false>(table + (i64)row * kp, li, kp, out);
    } else {
        const unsigned base = (unsigned)row * (unsigned)(kp * 4);
#pragma unroll
        for (int j = 0; j < S::CH; ++j) {
            const unsigned off = S::ok(li, j, kp) ? base + 4u * (unsigned)S::c4(li, j) : base;
            out[j] = *reinterpret_cast<const float4 *>(reinterpret_cast<const char *>(table) + off);
        }
    }
}

// responsibilities of one non-zero for this lane's chunks.
//   keep[j] = (Vt*U > thresh) ? Vt*U : 0        (plsa.py:97-102)
//   returns the lane-partial of the thresholded norm; unth gets the un-thresholded partial
template <int CH, bool WANT_UNTH>
__device__ __forceinline__ float products(const float4 (&u)[CH], const float4 (&vt)[CH],
                                          float thresh, float4 (&keep)[CH], float &unth) {
    float part = 0.f;
    unth = 0.f;
#pragma unroll
    for (int j = 0; j < CH; ++j) {
        float4 v;
        v.x = vt[j].x * u[j].x; v.y = vt[j].y * u[j].y; v.z = vt[j].z * u[j].z; v.w = vt[j].w * u[j].w;
        if (WANT_UNTH) unth = j ? unth + hsum(v) : hsum(v);
        keep[j].x = v.x > thresh ? v.x : 0.f;
        keep[j].y = v.y > thresh ? v.y : 0.f;
        keep[j].z = v.z > thresh ? v.z : 0.f;
        keep[j].w = v.w > thresh ? v.w : 0.f;
        part = j ? part + hsum(keep[j]) : hsum(keep[j]);   // no "0 + x" (it survives as an instruction: -0 + 0)
    }
    return part;
}

// 1/norm when norm > 0 else 0 (plsa.py:103-105: rows with a zero norm stay all-zero); v_rcp_f32
__device__ __forceinline__ float inv_norm(float norm) {
    return norm > 0.f ? __builtin_amdgcn_rcpf(norm) : 0.f;
}

template <int CH>
__device__ __forceinline__ void scale(float4 (&a)[CH], float s) {
#pragma unroll
    for (int j = 0; j < CH; ++j) { a[j].x *= s; a[j].y *= s; a[j].z *= s; a[j].w *= s; }
}

// Thresholds below the smallest normal float (e_step_thresh = 0 is legal) let a responsibility norm be
// denormal; 1/norm then overflows although every quotient v/norm the reference forms (plsa.py:104) is
// <= 1.  TINY (a TEMPLATE parameter of the hot kernels, chosen by the host from thresh < TINY_THRESH) compiles in a rescue that multiplies such a norm
// and its products by 2^100 first -- exact, powers of two -- so the reciprocal stays finite.  With the
// default threshold (1e-32) every non-zero norm exceeds 1e-32 and the rescue is not in the code at all: rounds 1-3 passed
// `tiny` as a run-time flag, which left one exec-masked region of 5 multiplies + 4 scalar instructions per non-zero in every
// launch (9 of ~45 instructions per entry) and kept the scheduler from interleaving the entries of a batch.
constexpr float TINY_THRESH = 5e-38f;
template <int CH>
__device__ __forceinline__ void rescue_tiny(bool tiny, float &norm, float4 (&keep)[CH]) {
    if (tiny && norm < 0x1p-100f) {
        norm *= 0x1p100f;
        scale(keep, 0x1p100f);
    }
}

// ------------------------------------------------------------------------------------------------
// k_e_step: the materialising E-step, plsa.py:91-105.  nnz-parallel: a wave takes a tile of 64
// consecutive non-zeros, loads their (doc, word) ids with one coalesced access each, then walks the
// tile 64/LPN non-zeros per step, UNR steps per batch: all 2*UNR row gathers of a batch are issued
// back to back, then the batch is reduced and streamed out.  P is written with non-temporal
// 16-byte stores (written once, never re-read here: it must not evict the factor rows from L2);
// the buffer carries one tile of slack so the last tile needs no store predicate.
// Algorithmic bytes: 4(n+1) + 4 nnz + 4k nnz + 4k(n+m)   (SURVEY.md section 8d).
// ------------------------------------------------------------------------------------------------
template <class S, bool TINY = false>
__global__ __launch_bounds__(256, PLSA_WAVES) void k_e_step(const int *__restrict__ rowidx,
                                                const int *__restrict__ colidx, i64 nnz,
                                                const float *__restrict__ U,
                                                const float *__restrict__ Vt, float *__restrict__ P,
                                                int kp_rt, float thresh) {
    constexpr int LPN = S::LPN, CH = S::CH, UNR = S::UNR_EF;
    constexpr int GPW = 64 / LPN;  // groups (= non-zeros per step) per wave
    const int kp = S::kp(kp_rt);
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int g = lane / LPN, li = lane % LPN;
    const i64 tiles = (nnz + 63) >> 6;
    constexpr bool tiny = TINY;
    for (i64 t = (i64)blockIdx.x * 4 + wave; t < tiles; t += (i64)gridDim.x * 4) {
        const i64 base = t << 6;
        const i64 mine = base + lane;
        const int d_l = mine < nnz ? __builtin_nontemporal_load(rowidx + mine) : 0;
        const int w_l = mine < nnz ? __builtin_nontemporal_load(colidx + mine) : 0;
#pragma unroll
        for (int s0 = 0; s0 < LPN; s0 += UNR) {
            float4 u[UNR][CH], vt[UNR][CH];
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                const int src = (s0 + q) * GPW + g;
                const int d = __shfl(d_l, src, 64);
                const int w = __shfl(w_l, src, 64);
                load_row<S, true, PLSA_NT_STREAMS>(U + (i64)d * kp, li, kp, u[q]);
                load_row<S, false>(Vt + (i64)w * kp, li, kp, vt[q]);
            }
#pragma unroll
            for (int q = 0; q < UNR; ++q) {
                float4 keep[CH];
                float unth;
                float norm = group_sum<LPN>(products<CH, false>(u[q], vt[q], thresh, keep, unth));
                rescue_tiny(tiny, norm, keep);
                const float inv = inv_norm(norm);
                float *prow = P + (base + (s0 + q) * GPW + g) * kp;
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    if (S::ok(li, j, kp)) {
                        float4 p;
                        p.x = keep[j].x * inv; p.y = keep[j].y * inv;
                        p.z = keep[j].z * inv; p.w = keep[j].w * inv;
                        st4_nt(prow + S::c4(li, j), p);
                    }
                }
            }
        }
    }
}

// Document-owned E-step: a group keeps its document's P(z|d) row in registers and walks the
// document's non-zeros (same traversal as k_row_pass: rows in descending-length order, or row items
// for corpora with few long documents), so the only gathers are the P(w|z) rows and the (doc) index
// stream is not read at all.  Measured against the flat kernel above on config 3: the P(z|d) row
// loads are L1/L2 hits there but still cost a tenth of the kernel (tools/experiments/README.md).
template <class S, bool TINY = false>
__global__ __launch_bounds__(256, PLSA_WAVES) void k_e_step_rows(const int *__restrict__ indptr,
                                                     const int *__restrict__ colidx, int n,
                                                     const int *__restrict__ row_order,
                                                     const float *__restrict__ U,
                                                     const float *__restrict__ Vt, float *__restrict__ P,
                                                     int kp_rt, float thresh,
                                                     const int *__restrict__ ritem_row,
                                                     const int *__restrict__ ritem_start, int rseg,
                                                     i64 n_ritems) {
    constexpr int LPN = S::LPN, CH = S::CH, UNR = S::UNR_E;
    constexpr int GPB = 256 / LPN;
    const int kp = S::kp(kp_rt);
    const int li = threadIdx.x % LPN;
    const int gid = threadIdx.x / LPN;
    const bool items = ritem_row != nullptr;
    const i64 n_work = items ? n_ritems : (i64)n;
    constexpr bool tiny = TINY;
    for (i64 r = (i64)blockIdx.x * GPB + gid; r < n_work; r += (i64)gridDim.x * GPB) {
        const int d = items ? ritem_row[r] : (row_order ? row_order[r] : (int)r);
        const int j0 = items ? ritem_start[r] : indptr[d];
        const int j1 = items ? min(j0 + rseg, indptr[d + 1]) : indptr[d + 1];
        float4 u[CH];
        load_row<S, true, PLSA_NT_STREAMS>(U + (i64)d * kp, li, kp, u);
        int w_n = (j0 + li < j1) ? ldi(colidx + j0 + li) : 0;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int w_l = w_n;
            const int jn = jb + LPN + li;
            w_n = jn < j1 ? ldi(colidx + jn) : 0;
            const int cnt = min(LPN, j1 - jb);
            for (int s0 = 0; s0 < cnt; s0 += UNR) {
                float4 vt[UNR][CH];
#pragma unroll
                for (int q = 0; q < UNR; ++q)
                    load_row<S, false>(Vt + (i64)__shfl(w_l, s0 + q, LPN) * kp, li, kp, vt[q]);
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    float4 keep[CH];
                    float unth;
                    float norm = group_sum<LPN>(products<CH, false>(u, vt[q], thresh, keep, unth));
                    rescue_tiny(tiny, norm, keep);
                    const float inv = inv_norm(norm);
                    if (s0 + q < cnt) {
                        float *prow = P + (i64)(jb + s0 + q) * kp;
#pragma unroll
                        for (int j = 0; j < CH; ++j) {
                            if (S::ok(li, j, kp)) {
                                float4 p;
                                p.x = keep[j].x * inv; p.y = keep[j].y * inv;
                                p.z = keep[j].z * inv; p.w = keep[j].w * inv;
                                st4_nt(prow + S::c4(li, j), p);
                            }
                        }
                    }
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_row_pass: document-owned half of the M-step (plsa.py:182-194 restricted to P(z|d), then the
// P(z|d) part of 196-202), optionally fused with the E-step (FROM_P = false: responsibilities are
// recomputed in registers and never touch HBM) and with the log-likelihood of the CURRENT factors
// (WANT_LL, plsa.py:375-384).
// A group owns one row: no atomics on U, the row norm is a group sum, the normalised row is written
// once.  Rows are visited through `row_order` (descending length) so the groups of a wave finish
// together; entries beyond the row end are padded with (word 0, count 0) and add exact zeros.
// ------------------------------------------------------------------------------------------------
template <class S, bool FROM_P, bool WANT_LL, bool TINY = false>
__global__ __launch_bounds__(256, PLSA_WAVES) void k_row_pass(const int *__restrict__ indptr,
                                                  const int *__restrict__ colidx,
                                                  const float *__restrict__ vals, int n,
                                                  const int *__restrict__ row_order,
                                                  const float *__restrict__ U,
                                                  const float *__restrict__ Vt,
                                                  const float *__restrict__ P,
                                                  float *__restrict__ U_new,
                                                  const float *__restrict__ sw,
                                                  float *__restrict__ norm_pdz_out, int kp_rt,
                                                  float thresh, double *__restrict__ ll_partials,
                                                  const int *__restrict__ ritem_row,
                                                  const int *__restrict__ ritem_start, int rseg,
                                                  i64 n_ritems, float *__restrict__ rpartial, int xcd_rows) {
    constexpr int LPN = S::LPN, CH = S::CH, UNR = S::UNR;
    constexpr int GPB = 256 / LPN;  // groups per block
    const int kp = S::kp(kp_rt);
    const int li = threadIdx.x % LPN;
    const int gid = threadIdx.x / LPN;
    double ll = 0.0;
    // xcd_rows (PLSA_ROW_XCD, experiment): workgroup b runs on XCD b % 8; XCD x takes the x-th EIGHTH of the visiting list
    // (the grid covers the list in one trip and is a multiple of 8), so that an XCD's L2 sees the P(w|z) rows of one
    // contiguous range of documents only -- worth something when neighbouring documents share vocabulary
    const i64 first_block = xcd_rows ? (i64)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3) : (i64)blockIdx.x;
    constexpr bool tiny = !FROM_P && TINY;
    // item mode (ritem_row != nullptr): rows are cut into items of <= rseg entries, a group owns
    // one item and writes an un-normalised partial row that k_row_reduce adds up in item order.
    // Used when there are too few / too uneven rows to fill the chip (few long documents).
    const bool items = ritem_row != nullptr;
    const i64 n_work = items ? n_ritems : (i64)n;
    for (i64 r = first_block * GPB + gid; r < n_work; r += (i64)gridDim.x * GPB) {
        const int d = items ? ritem_row[r] : (row_order ? row_order[r] : (int)r);
        const int j0 = items ? ritem_start[r] : indptr[d];
        const int j1 = items ? min(j0 + rseg, indptr[d + 1]) : indptr[d + 1];
        float4 u[CH], acc[CH];
        load_row<S, true, PLSA_NT_STREAMS>(U + (i64)d * kp, li, kp, u);
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = zero4();
        const float swd = (WANT_LL && sw) ? sw[d] : 1.0f;
        // software-pipelined index stream: the next LPN (word, count) pairs are in flight while the
        // current ones are consumed
        int w_n = (j0 + li < j1) ? ldi(colidx + j0 + li) : 0;
        float x_n = (j0 + li < j1) ? ldf(vals + j0 + li) : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int w_l = w_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            w_n = jn < j1 ? ldi(colidx + jn) : 0;
            x_n = jn < j1 ? ldf(vals + jn) : 0.f;
            const int cnt = min(LPN, j1 - jb);
            for (int s0 = 0; s0 < cnt; s0 += UNR) {
                float4 a[UNR][CH];   // Vt rows (fused) or P rows (FROM_P)
                float x[UNR];
                // x / norm is formed ONCE, in the lane that loaded the entry (lane s0 + q holds the count of entry q; s0 == 0 here),
                // and broadcast -- instead of broadcasting the count and forming the quotient in every lane: 8 x (reciprocal,
                // compare, select, multiply) per batch become 8 selects + one such sequence; same operations on the same values
                constexpr bool OWNER = !FROM_P && UNR == LPN;
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    const int w = __shfl(w_l, s0 + q, LPN);
                    if (!OWNER) x[q] = __shfl(x_l, s0 + q, LPN);
                    if (FROM_P) load_row<S, false>(P + (i64)min(jb + s0 + q, j1 - 1) * kp, li, kp, a[q]);
                    else gather_row<S>(Vt, w, li, kp, a[q]);
                }
                float my_dot = 1.f, my_x = 0.f;   // WANT_LL: lane q of the group takes non-zero q of the batch
                if (OWNER) {
                    float nmine = 0.f;
#pragma unroll
                    for (int q = 0; q < UNR; ++q) {
                        float unth;
                        const float part = products<CH, WANT_LL>(u, a[q], thresh, a[q], unth);
                        float norm = group_sum<LPN>(part);
                        rescue_tiny(tiny, norm, a[q]);
                        if (WANT_LL) {
                            const float dot = group_sum<LPN>(unth);
                            if (li == q && s0 + q < cnt) { my_dot = dot; my_x = x_l; }
                        }
                        nmine = (li == s0 + q) ? norm : nmine;
                    }
                    const float xs_mine = x_l * inv_norm(nmine);   // lane s0 + q: entry q of the batch
#pragma unroll
                    for (int q = 0; q < UNR; ++q) {
                        const float xs = __shfl(xs_mine, s0 + q, LPN);
#pragma unroll
                        for (int j = 0; j < CH; ++j) {   // s = x * P(z|w,d); U[d,z] += s   plsa.py:188-191
                            acc[j].x += xs * a[q][j].x; acc[j].y += xs * a[q][j].y;
                            acc[j].z += xs * a[q][j].z; acc[j].w += xs * a[q][j].w;
                        }
                    }
                } else {
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    float4 pz[CH];
                    if (FROM_P) {
#pragma unroll
                        for (int j = 0; j < CH; ++j) pz[j] = S::ok(li, j, kp) ? a[q][j] : zero4();
                    } else {
                        float unth;
                        const float part = products<CH, WANT_LL>(u, a[q], thresh, pz, unth);
                        float norm = group_sum<LPN>(part);
                        rescue_tiny(tiny, norm, pz);
                        if (WANT_LL) {
                            const float dot = group_sum<LPN>(unth);
                            if (li == q && s0 + q < cnt) { my_dot = dot; my_x = x[q]; }
                        }
                        x[q] *= inv_norm(norm);   // s = x * (v / norm) evaluated as v * (x / norm)
                    }
#pragma unroll
                    for (int j = 0; j < CH; ++j) {   // s = x * P(z|w,d); U[d,z] += s   plsa.py:188-191
                        acc[j].x += x[q] * pz[j].x; acc[j].y += x[q] * pz[j].y;
                        acc[j].z += x[q] * pz[j].z; acc[j].w += x[q] * pz[j].w;
                    }
                }
                }
                // x * log(sum_z P(w|z) P(z|d)) * sample_weight, plsa.py:380-383: one logf sequence per batch
                // (lanes 0 .. UNR-1 each hold one non-zero; padded slots keep x = 0, dot = 1 -> exactly 0)
                if (WANT_LL && !FROM_P && li < UNR) ll += (double)(my_x * logf(my_dot) * swd);
            }
        }
        if (items) {
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (S::ok(li, j, kp)) st4(rpartial + r * kp + S::c4(li, j), acc[j]);
            continue;
        }
        // norm_pdz[d] and the division, plsa.py:194, 200-202
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) part += hsum(acc[j]);
        const float rown = group_sum<LPN>(part);
        if (norm_pdz_out && li == 0) norm_pdz_out[d] = rown;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (S::ok(li, j, kp)) {
                float4 o = acc[j];
                if (rown > 0.f) { o.x /= rown; o.y /= rown; o.z /= rown; o.w /= rown; }
                st4(U_new + (i64)d * kp + S::c4(li, j), o);
            }
        }
    }
    if (WANT_LL) {
        // block reduction of the per-group log-likelihood partials -> one double per block
        __shared__ double red[256];
        red[threadIdx.x] = ll;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) ll_partials[blockIdx.x] = red[0];
    }
}

// k_row_reduce: adds the item partials of each document (fixed order), then norm_pdz and the
// division (plsa.py:194, 200-202) -- the tail of k_row_pass for the item mode.
template <class S>
__global__ __launch_bounds__(256) void k_row_reduce(const int *__restrict__ ritem_first, int n,
                                                    const float *__restrict__ rpartial,
                                                    float *__restrict__ U_new,
                                                    float *__restrict__ norm_pdz_out, int kp_rt) {
    constexpr int LPN = S::LPN, CH = S::CH;
    constexpr int GPB = 256 / LPN;
    const int kp = S::kp(kp_rt);
    const int li = threadIdx.x % LPN;
    const int gid = threadIdx.x / LPN;
    for (i64 d = (i64)blockIdx.x * GPB + gid; d < n; d += (i64)gridDim.x * GPB) {
        const int i0 = ritem_first[d], i1 = ritem_first[d + 1];
        float4 acc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = zero4();
        for (int it = i0; it < i1; ++it) {
            float4 p[CH];
            load_row<S, true>(rpartial + (i64)it * kp, li, kp, p);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                acc[j].x += p[j].x; acc[j].y += p[j].y; acc[j].z += p[j].z; acc[j].w += p[j].w;
            }
        }
        float part = 0.f;
#pragma unroll
        for (int j = 0; j < CH; ++j) part += hsum(acc[j]);
        const float rown = group_sum<LPN>(part);
        if (norm_pdz_out && li == 0) norm_pdz_out[d] = rown;
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (S::ok(li, j, kp)) {
                float4 o = acc[j];
                if (rown > 0.f) { o.x /= rown; o.y /= rown; o.z /= rown; o.w /= rown; }
                st4(U_new + d * kp + S::c4(li, j), o);
            }
        }
    }
}

__global__ void k_ritem_fill(const int *__restrict__ indptr, const int *__restrict__ ritem_first, int n,
                             int seg, int *__restrict__ ritem_row, int *__restrict__ ritem_start) {
    const i64 d = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (d < n) {
        const int i0 = ritem_first[d], i1 = ritem_first[d + 1];
        for (int i = i0; i < i1; ++i) {
            ritem_row[i] = (int)d;
            ritem_start[i] = indptr[d] + (i - i0) * seg;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// k_col_pass: vocabulary-owned half of the M-step (plsa.py:190/296, 193/299) without atomics.  The
// active matrix is also held column-major (CSC: colptr, csc_row, csc_val, csc_pos = position of the
// entry in CSR order).  Long columns (Zipf head words) are cut into items of at most SEG entries;
// a group owns one item, accumulates x * P(z|w,d) [* sample_weight] in registers and writes one
// partial k-vector; k_col_reduce* add the partials of a column in item order (bit-reproducible).
// Items are visited through `item_order` (ascending first document) so that the groups running
// concurrently gather from the same band of U rows.  FROM_P = false recomputes the
// responsibilities from U (gather) and Vt (registers).
// ------------------------------------------------------------------------------------------------
// block-wide sum of per-thread k-vector chunks (float) in float64, fixed group order -> out_row[kp]
template <class S, int THREADS = 256>
__device__ __forceinline__ void block_colsum(const float4 (&csum)[S::CH], int li, int gid, int kp,
                                             double *sred /*[THREADS / LPN][kp]*/, double *__restrict__ out_row) {
    constexpr int GPB = THREADS / S::LPN;
#pragma unroll
    for (int j = 0; j < S::CH; ++j) {
        if (S::ok(li, j, kp)) {
            double *p = sred + gid * kp + S::c4(li, j);
            p[0] = (double)csum[j].x; p[1] = (double)csum[j].y; p[2] = (double)csum[j].z; p[3] = (double)csum[j].w;
        }
    }
    __syncthreads();
    for (int z = threadIdx.x; z < kp; z += THREADS) {
        double t = 0.0;
        for (int g = 0; g < GPB; ++g) t += sred[g * kp + z];
        out_row[z] = t;
    }
}

// one batch of UN entries of a column item: all UN gathers are issued before the first use
template <class S, bool FROM_P, int UN, bool TINY>
__device__ __forceinline__ void col_batch(int s0, int d_l, float x_l, int p_l, int li, int kp, float thresh,
                                          const float *__restrict__ U, const float *__restrict__ P,
                                          const float4 (&vt)[S::CH], float4 (&acc)[S::CH]) {
    constexpr int LPN = S::LPN, CH = S::CH;
    constexpr bool OWNER = !FROM_P && UN >= 4;   // x / norm formed once, in the lane that loaded the entry (see k_row_pass)
    float4 a[UN][CH];   // U rows (fused) or P rows (FROM_P)
    float x[UN];
#pragma unroll
    for (int q = 0; q < UN; ++q) {
        if (!OWNER) x[q] = __shfl(x_l, s0 + q, LPN);     // lanes beyond the item hold (doc 0, count 0): exact zeros
        if (FROM_P) {
            const int pos = __shfl(p_l, s0 + q, LPN);
            load_row<S, false>(P + (i64)pos * kp, li, kp, a[q]);
        } else {
            const int d = __shfl(d_l, s0 + q, LPN);
            gather_row<S>(U, d, li, kp, a[q]);
        }
    }
    if (OWNER) {
        float nmine = 0.f;
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            float unth;
            float norm = group_sum<LPN>(products<CH, false>(a[q], vt, thresh, a[q], unth));
            rescue_tiny(TINY, norm, a[q]);
            nmine = (li == s0 + q) ? norm : nmine;
        }
        const float xs_mine = x_l * inv_norm(nmine);
#pragma unroll
        for (int q = 0; q < UN; ++q) {
            const float xs = __shfl(xs_mine, s0 + q, LPN);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                acc[j].x += xs * a[q][j].x; acc[j].y += xs * a[q][j].y;
                acc[j].z += xs * a[q][j].z; acc[j].w += xs * a[q][j].w;
            }
        }
        return;
    }
#pragma unroll
    for (int q = 0; q < UN; ++q) {
        float4 pz[CH];
        if (FROM_P) {
#pragma unroll
            for (int j = 0; j < CH; ++j) pz[j] = S::ok(li, j, kp) ? a[q][j] : zero4();
        } else {
            float unth;
            float norm = group_sum<LPN>(products<CH, false>(a[q], vt, thresh, pz, unth));
            rescue_tiny(TINY, norm, pz);
            x[q] *= inv_norm(norm);
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            acc[j].x += x[q] * pz[j].x; acc[j].y += x[q] * pz[j].y;
            acc[j].z += x[q] * pz[j].z; acc[j].w += x[q] * pz[j].w;
        }
    }
}

// Schedule of the column pass.  The items are visited in ascending-first-document order (`item_rec`, one
// {column, start, end, partial slot} record per visiting position) in CHUNKS of GPB = 256 / LPN consecutive items:
// one chunk = one trip of a workgroup, one item per group.  Workgroup b runs on XCD b % 8 (observed dispatch rule; a
// different placement only costs speed) and XCD x walks the chunks [xcd_lo[x], xcd_lo[x+1]) -- a contiguous stretch
// of the list, so that all workgroups sharing an L2 gather from the same band of U rows.  The boundaries are
// MEASURED (plsa_hip.hip::ensure_balance: the TIMED instantiation records every workgroup's end time and the
// stretches are resized until the eight XCDs finish together): equal stretches left the XCD holding the
// rare-word items -- they sort to the front of the list and all their gathers miss -- 35 % behind the fastest one
// (round 3: 2.35 -> 2.06 ms at config 3 with 256-entry items, 1.93 ms with 128).
// Per chunk the workgroup also writes the float64 sum of its GPB accumulators (`chunk_sums`, the rows norm_pwz is
// added up from, in chunk order): every result is independent of the boundaries and of the grid.
template <class S, bool FROM_P, bool TIMED, bool TINY = false>
__global__ __launch_bounds__(256, PLSA_WAVES_COL) void k_col_pass(const int4 *__restrict__ item_rec, i64 n_items,
                                                  const int *__restrict__ xcd_lo,
                                                  const int *__restrict__ csc_row,
                                                  const float *__restrict__ csc_val,
                                                  const int *__restrict__ csc_pos,
                                                  const float *__restrict__ U,
                                                  const float *__restrict__ Vt,
                                                  const float *__restrict__ P,
                                                  const float *__restrict__ sw,
                                                  float *__restrict__ partial, int kp_rt, float thresh,
                                                  int xcd_split, double *__restrict__ chunk_sums,
                                                  unsigned long long *__restrict__ t_end) {
    constexpr int LPN = S::LPN, CH = S::CH, UNR = S::UNR_COL;
    constexpr int GPB = 256 / LPN;
    extern __shared__ double scol[];   // [GPB][kp]: sum of the chunk's accumulators (-> norm_pwz)
    const int kp = S::kp(kp_rt);
    const int li = threadIdx.x % LPN;
    const int gid = threadIdx.x / LPN;
    const int n_chunks = (int)((n_items + GPB - 1) / GPB);
    const int xcd = xcd_split ? (int)(blockIdx.x & 7) : 0;
    const int nq = xcd_split ? (int)((gridDim.x + 7 - xcd) / 8) : (int)gridDim.x;   // workgroups on this XCD
    const int q = xcd_split ? (int)(blockIdx.x >> 3) : (int)blockIdx.x;
    const int c_lo = xcd_split ? xcd_lo[xcd] : 0, c_hi = xcd_split ? xcd_lo[xcd + 1] : n_chunks;
    if (TIMED && blockIdx.x == 0 && threadIdx.x == 0) t_end[gridDim.x] = wall_clock64();   // launch start
    for (int chunk = c_lo + q; chunk < c_hi; chunk += nq) {
        const i64 io = (i64)chunk * GPB + gid;
        float4 acc[CH];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = zero4();
        if (io < n_items) {
            const int4 rec = item_rec[io];
            const int w = rec.x, j0 = rec.y, j1 = rec.z;
            float4 vt[CH];
            load_row<S, true, PLSA_NT_STREAMS>(Vt + (i64)w * kp, li, kp, vt);
            int d_n = (j0 + li < j1) ? ldi(csc_row + j0 + li) : 0;
            float x_n = (j0 + li < j1) ? ldf(csc_val + j0 + li) : 0.f;
            int p_n = (FROM_P && j0 + li < j1) ? ldi(csc_pos + j0 + li) : 0;
            for (int jb = j0; jb < j1; jb += LPN) {
                const int d_l = d_n, p_l = p_n;
                float x_l = x_n;
                const int jn = jb + LPN + li;
                d_n = jn < j1 ? ldi(csc_row + jn) : 0;
                x_n = jn < j1 ? ldf(csc_val + jn) : 0.f;
                if (FROM_P) p_n = jn < j1 ? ldi(csc_pos + jn) : 0;
                if (sw) x_l *= sw[d_l];  // t = s * sample_weight[d]  (plsa.py:294), folded into the count
                const int cnt = min(LPN, j1 - jb);
                // full batches of UNR gathers, then the remainder two at a time: most vocabulary columns
                // are short (Zipf tail) and must not pay for UNR padded gathers
                int s0 = 0;
                for (; s0 + UNR <= cnt; s0 += UNR)
                    col_batch<S, FROM_P, UNR, TINY>(s0, d_l, x_l, p_l, li, kp, thresh, U, P, vt, acc);
                constexpr int TAIL = (UNR >= 2 && LPN >= 2) ? 2 : 1;
                for (; s0 < cnt; s0 += TAIL)
                    col_batch<S, FROM_P, TAIL, TINY>(s0, d_l, x_l, p_l, li, kp, thresh, U, P, vt, acc);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j)
                if (S::ok(li, j, kp)) st4(partial + (i64)rec.w * kp + S::c4(li, j), acc[j]);
        }
        block_colsum<S>(acc, li, gid, kp, scol, chunk_sums + (i64)chunk * kp);
        __syncthreads();               // scol is rewritten by the next chunk
    }
    if (TIMED && threadIdx.x == 0) t_end[blockIdx.x] = wall_clock64();
}

// visiting-order item records of the column pass: rec[io] = {column, first entry, end, item id (partial slot)}
__global__ void k_item_records(const int *__restrict__ item_order, const int *__restrict__ item_col,
                               const int *__restrict__ item_start, const int *__restrict__ item_end, i64 n_items,
                               int4 *__restrict__ rec) {
    const i64 io = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (io < n_items) {
        const int it = item_order ? item_order[io] : (int)io;
        rec[io] = make_int4(item_col[it], item_start[it], item_end[it], it);
    }
}

// adds the item partials of each column (fixed order) into the un-normalised Vt_new.
// Columns with more than `heavy_items` items (the Zipf head) take a whole block: its groups stride
// over the column's items (fixed assignment, loads batched four deep, added in item order), then their
// sums are added in group order through LDS -> bit-reproducible.  Every other column takes one group.
// `block` of `nblocks`: the caller's position in the grid.  NORMALISE: divide by norm_pwz on the way out
// (plsa.py:196-199: true division, only where the norm is positive) -- snorm is its LDS copy.
__device__ __forceinline__ float div_pos(float v, float n) { return n > 0.f ? v / n : v; }

template <class S, bool NORMALISE>
__device__ __forceinline__ void col_reduce_body(const int *__restrict__ item_first, int m, int heavy_items,
                                                const int *__restrict__ heavy_cols, int n_heavy,
                                                const float *__restrict__ partial, float *__restrict__ Vt_new,
                                                int kp, int block, int nblocks, float *sacc /*[GPB][kp]*/,
                                                const float *snorm /*[kp], NORMALISE only*/) {
    constexpr int LPN = S::LPN, CH = S::CH;
    constexpr int GPB = 256 / LPN;
    const int li = threadIdx.x % LPN;
    const int gid = threadIdx.x / LPN;
    float4 acc[CH];
    for (int h = block; h < n_heavy; h += nblocks) {
        const int c = heavy_cols[h];
        const int i0 = item_first[c], i1 = item_first[c + 1];
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = zero4();
        constexpr int B = 4;   // rows in flight per group (16 measured slower: 281 against 285 iterations/s at config 3)
        for (int it = i0 + gid; it < i1; it += GPB * B) {
            float4 p[B][CH];
#pragma unroll
            for (int u = 0; u < B; ++u) {
                const int iu = it + u * GPB;
                if (iu < i1) {
                    load_row<S, true>(partial + (i64)iu * kp, li, kp, p[u]);
                } else {
#pragma unroll
                    for (int j = 0; j < CH; ++j) p[u][j] = zero4();
                }
            }
#pragma unroll
            for (int u = 0; u < B; ++u)
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    acc[j].x += p[u][j].x; acc[j].y += p[u][j].y; acc[j].z += p[u][j].z; acc[j].w += p[u][j].w;
                }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j)
            if (S::ok(li, j, kp)) st4(sacc + gid * kp + S::c4(li, j), acc[j]);
        __syncthreads();
        for (int z = threadIdx.x; z < kp; z += 256) {
            float t = 0.f;
            for (int g = 0; g < GPB; ++g) t += sacc[g * kp + z];
            Vt_new[(i64)c * kp + z] = NORMALISE ? div_pos(t, snorm[z]) : t;
        }
        __syncthreads();
    }
    for (i64 c = (i64)block * GPB + gid; c < m; c += (i64)nblocks * GPB) {
        const int i0 = item_first[c], i1 = item_first[c + 1];
        if (i1 - i0 > heavy_items) continue;
#pragma unroll
        for (int j = 0; j < CH; ++j) acc[j] = zero4();
        int it = i0;
        for (; it + 4 <= i1; it += 4) {        // four rows in flight, added in item order
            float4 p[4][CH];
#pragma unroll
            for (int u = 0; u < 4; ++u) load_row<S, true>(partial + (i64)(it + u) * kp, li, kp, p[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < CH; ++j) {
                    acc[j].x += p[u][j].x; acc[j].y += p[u][j].y; acc[j].z += p[u][j].z; acc[j].w += p[u][j].w;
                }
        }
        for (; it < i1; ++it) {
            float4 p[CH];
            load_row<S, true>(partial + (i64)it * kp, li, kp, p);
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                acc[j].x += p[j].x; acc[j].y += p[j].y; acc[j].z += p[j].z; acc[j].w += p[j].w;
            }
        }
#pragma unroll
        for (int j = 0; j < CH; ++j) {
            if (S::ok(li, j, kp)) {
                float4 o = acc[j];
                if (NORMALISE) {
                    const float *nn = snorm + S::c4(li, j);
                    o.x = div_pos(o.x, nn[0]); o.y = div_pos(o.y, nn[1]); o.z = div_pos(o.z, nn[2]); o.w = div_pos(o.w, nn[3]);
                }
                st4(Vt_new + c * kp + S::c4(li, j), o);
            }
        }
    }
}

// un-normalised sums -> Vacc (doc-sharded fit: the all-reduce comes next; plsa_em_accumulate)
template <class S>
__global__ __launch_bounds__(256) void k_col_reduce(const int *__restrict__ item_first, int m,
                                                    int heavy_items, const int *__restrict__ heavy_cols,
                                                    int n_heavy, const float *__restrict__ partial,
                                                    float *__restrict__ Vt_new, int kp_rt) {
    extern __shared__ float sacc[];  // [GPB][kp]
    col_reduce_body<S, false>(item_first, m, heavy_items, heavy_cols, n_heavy, partial, Vt_new, S::kp(kp_rt),
                              (int)blockIdx.x, (int)gridDim.x, sacc, nullptr);
}

// sums AND division in one pass: norm_pwz is already known when this runs, because the column pass
// itself adds up its accumulators (sum_w Vacc[w, z] == sum over all items of their partial rows), so the
// two extra sweeps over the [m, kp] accumulator (column sums, division) of the first version are gone.
template <class S>
__global__ __launch_bounds__(256) void k_col_reduce_norm(const int *__restrict__ item_first, int m,
                                                         int heavy_items, const int *__restrict__ heavy_cols,
                                                         int n_heavy, const float *__restrict__ partial,
                                                         const float *__restrict__ norm_pwz,
                                                         float *__restrict__ Vt_out, int kp_rt) {
    extern __shared__ float sdyn[];  // [GPB][kp] heavy-column sums, then [kp] norm_pwz
    const int kp = S::kp(kp_rt);
    constexpr int GPB = 256 / S::LPN;
    float *snorm = sdyn + GPB * kp;
    for (int z = threadIdx.x; z < kp; z += 256) snorm[z] = norm_pwz[z];
    __syncthreads();
    col_reduce_body<S, true>(item_first, m, heavy_items, heavy_cols, n_heavy, partial, Vt_out, kp,
                             (int)blockIdx.x, (int)gridDim.x, sdyn, snorm);
}

__global__ void k_heavy_list(const int *__restrict__ item_first, int m, int heavy_items,
                             int *__restrict__ heavy_cols, int *__restrict__ n_heavy) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m && item_first[c + 1] - item_first[c] > heavy_items)
        heavy_cols[atomicAdd(n_heavy, 1)] = (int)c;
}

// ------------------------------------------------------------------------------------------------
// P(w|z) normalisation, plsa.py:196-199: norm_pwz[z] = sum_w Vt_new[w,z], then divide.
//   k_colsum_partial : NORM_BLOCKS blocks, each sums a contiguous slab of words -> partials (f64)
//   k_colsum_final   : one block adds the NORM_BLOCKS partials in fixed order -> norm_pwz (f32)
//   k_v_normalise    : divides and writes the normalised topics into Vt.
// ------------------------------------------------------------------------------------------------
constexpr int NORM_BLOCKS = 256;

// slab `slab` of `n_slabs`: float32 strand sums of a contiguous range of words, added in strand order (f64)
__device__ __forceinline__ void colsum_slab_body(const float *__restrict__ Vt_new, int m, int kp, int slab,
                                                 int n_slabs, double *__restrict__ partials, double *sred /*[256]*/) {
    // thread t owns column z = t % span for rows t / span, t / span + rows_per_pass, ...
    const i64 per = ((i64)m + n_slabs - 1) / n_slabs;
    const i64 w0 = (i64)slab * per, w1 = min((i64)m, w0 + per);
    for (int zb = 0; zb < kp; zb += 256) {
        const int span = min(256, kp - zb);          // columns handled in this sweep
        const int rpp = 256 / span;                  // rows per pass
        const int z = zb + (int)threadIdx.x % span;
        const int ro = (int)threadIdx.x / span;
        float s = 0.f;
        if (ro < rpp)
            for (i64 w = w0 + ro; w < w1; w += rpp) s += Vt_new[w * kp + z];
        sred[threadIdx.x] = (ro < rpp) ? (double)s : 0.0;
        __syncthreads();
        if ((int)threadIdx.x < span) {
            double tot = 0.0;
            for (int r = 0; r < rpp; ++r) tot += sred[r * span + threadIdx.x];
            partials[(i64)slab * kp + zb + threadIdx.x] = tot;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_colsum_partial(const float *__restrict__ Vt_new, int m,
                                                        int kp, double *__restrict__ partials) {
    extern __shared__ double sred_dyn[];  // [256]
    colsum_slab_body(Vt_new, m, kp, (int)blockIdx.x, (int)gridDim.x, partials, sred_dyn);
}

// norm_pwz[z] = fixed-order sum of the slab partials (the 256 threads split the partials of each column
// into 256/span interleaved strands that are then added in strand order); out may be LDS or global
__device__ __forceinline__ void colsum_final_body(const double *__restrict__ partials, int n_partials, int kp,
                                                  float *out, double *sred /*[256]*/) {
    for (int zb = 0; zb < kp; zb += 256) {
        const int span = min(256, kp - zb);
        const int rpp = 256 / span;
        const int z = zb + (int)threadIdx.x % span;
        const int ro = (int)threadIdx.x / span;
        double tot = 0.0;
        if (ro < rpp) {
#pragma unroll 8
            for (int b = ro; b < n_partials; b += rpp) tot += partials[(i64)b * kp + z];
        }
        sred[threadIdx.x] = tot;
        __syncthreads();
        if ((int)threadIdx.x < span) {
            double t2 = 0.0;
            for (int r = 0; r < rpp; ++r) t2 += sred[r * span + threadIdx.x];
            out[zb + threadIdx.x] = (float)t2;
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_colsum_final(const double *__restrict__ partials, int n_partials,
                                                      int kp, float *__restrict__ norm_pwz) {
    __shared__ double sred[256];
    colsum_final_body(partials, n_partials, kp, norm_pwz, sred);
}

// stage 1 of norm_pwz from the column pass' per-block sums: `rows` rows of kp doubles -> gridDim.x rows
// (block b adds rows b*per .. in strand order; same thread layout as colsum_slab_body)
__global__ __launch_bounds__(256) void k_norm_reduce(const double *__restrict__ in, int rows, int kp,
                                                     double *__restrict__ out) {
    __shared__ double sred[256];
    const i64 per = ((i64)rows + gridDim.x - 1) / gridDim.x;
    const i64 r0 = (i64)blockIdx.x * per, r1 = min((i64)rows, r0 + per);
    for (int zb = 0; zb < kp; zb += 256) {
        const int span = min(256, kp - zb);
        const int rpp = 256 / span;
        const int z = zb + (int)threadIdx.x % span;
        const int ro = (int)threadIdx.x / span;
        double s = 0.0;
        if (ro < rpp) {
#pragma unroll 8
            for (i64 r = r0 + ro; r < r1; r += rpp) s += in[r * kp + z];
        }
        sred[threadIdx.x] = (ro < rpp) ? s : 0.0;
        __syncthreads();
        if ((int)threadIdx.x < span) {
            double tot = 0.0;
            for (int r = 0; r < rpp; ++r) tot += sred[r * span + threadIdx.x];
            out[(i64)blockIdx.x * kp + zb + threadIdx.x] = tot;
        }
        __syncthreads();
    }
}

// division by norm_pwz (snorm: LDS copy), plsa.py:196-199
__device__ __forceinline__ void v_normalise_body(const float *__restrict__ Vt_new, float *__restrict__ Vt, int m,
                                                 int kp, const float *snorm, int block, int nblocks) {
    const i64 total4 = (i64)m * kp / 4;
    const int kq = kp / 4;
    for (i64 i = (i64)block * 256 + threadIdx.x; i < total4; i += (i64)nblocks * 256) {
        const int z4 = (int)(i % kq) * 4;
        float4 v = ld4(Vt_new + i * 4);
        const float n0 = snorm[z4], n1 = snorm[z4 + 1], n2 = snorm[z4 + 2], n3 = snorm[z4 + 3];
        if (n0 > 0.f) v.x /= n0;
        if (n1 > 0.f) v.y /= n1;
        if (n2 > 0.f) v.z /= n2;
        if (n3 > 0.f) v.w /= n3;
        st4(Vt + i * 4, v);
    }
}

__global__ __launch_bounds__(256) void k_v_normalise(const float *__restrict__ Vt_new,
                                                     float *__restrict__ Vt, int m, int kp,
                                                     const float *__restrict__ norm_pwz) {
    extern __shared__ float snorm[];  // [kp]
    for (int z = threadIdx.x; z < kp; z += 256) snorm[z] = norm_pwz[z];
    __syncthreads();
    v_normalise_body(Vt_new, Vt, m, kp, snorm, (int)blockIdx.x, (int)gridDim.x);
}

// final, fixed-order sum of the per-block log-likelihood partials
__global__ void k_ll_final(const double *__restrict__ partials, int nb, double *__restrict__ out) {
    __shared__ double red[256];
    double s = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) s += partials[i];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// standalone log-likelihood, plsa.py:375-384 (row-owned; same traversal as k_row_pass)
template <class S>
__global__ __launch_bounds__(256, PLSA_WAVES) void k_loglik(const int *__restrict__ indptr,
                                                const int *__restrict__ colidx,
                                                const float *__restrict__ vals, int n,
                                                const int *__restrict__ row_order,
                                                const float *__restrict__ U,
                                                const float *__restrict__ Vt,
                                                const float *__restrict__ sw, int kp_rt,
                                                double *__restrict__ ll_partials) {
    constexpr int LPN = S::LPN, CH = S::CH, UNR = S::UNR;
    constexpr int GPB = 256 / LPN;
    const int kp = S::kp(kp_rt);
    const int li = threadIdx.x % LPN;
    const int gid = threadIdx.x / LPN;
    double ll = 0.0;
    for (i64 r = (i64)blockIdx.x * GPB + gid; r < n; r += (i64)gridDim.x * GPB) {
        const int d = row_order ? row_order[r] : (int)r;
        const int j0 = indptr[d], j1 = indptr[d + 1];
        float4 u[CH];
        load_row<S, true, PLSA_NT_STREAMS>(U + (i64)d * kp, li, kp, u);
        const float swd = sw ? sw[d] : 1.0f;
        int w_n = (j0 + li < j1) ? ldi(colidx + j0 + li) : 0;
        float x_n = (j0 + li < j1) ? ldf(vals + j0 + li) : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int w_l = w_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            w_n = jn < j1 ? ldi(colidx + jn) : 0;
            x_n = jn < j1 ? ldf(vals + jn) : 0.f;
            const int cnt = min(LPN, j1 - jb);
            for (int s0 = 0; s0 < cnt; s0 += UNR) {
                float4 vt[UNR][CH];
                float x[UNR];
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    const int w = __shfl(w_l, s0 + q, LPN);
                    x[q] = __shfl(x_l, s0 + q, LPN);
                    load_row<S, false>(Vt + (i64)w * kp, li, kp, vt[q]);
                }
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    float part = 0.f;
#pragma unroll
                    for (int j = 0; j < CH; ++j)
                        part += (vt[q][j].x * u[j].x + vt[q][j].y * u[j].y) + (vt[q][j].z * u[j].z + vt[q][j].w * u[j].w);
                    const float dot = group_sum<LPN>(part);
                    if (li == 0 && s0 + q < cnt) ll += (double)(x[q] * logf(dot) * swd);
                }
            }
        }
    }
    __shared__ double red[256];
    red[threadIdx.x] = ll;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ll_partials[blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------------------
// layout / corpus utility kernels
// ------------------------------------------------------------------------------------------------
// rowidx[j] = d for j in [indptr[d], indptr[d+1])  (COO row ids of the nnz-parallel E-step)
__global__ void k_expand_rows(const int *__restrict__ indptr, int n, int *__restrict__ rowidx) {
    const int lane = threadIdx.x & 63;
    const i64 wid = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nw = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 d = wid; d < n; d += nw) {
        const int j0 = indptr[d], j1 = indptr[d + 1];
        for (int j = j0 + lane; j < j1; j += 64) rowidx[j] = (int)d;
    }
}

// sort key of the document order: the row length (sorted DESCENDING), or -- range > 0, the XCD-contiguous document
// schedule (PLSA_ROW_XCD) -- documents of one RANGE of `range` consecutive documents first, longest first inside a range:
// key = (number of ranges - 1 - d / range) << 22 | min(length, 2^22 - 1), also sorted descending
__global__ void k_row_lengths(const int *__restrict__ indptr, int n, int *__restrict__ len,
                              int *__restrict__ ids, int range) {
    const i64 r = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (r < n) {
        const int l = indptr[r + 1] - indptr[r];
        len[r] = range > 0 ? (int)((unsigned)((n - 1) / range - (int)(r / range)) << 22 | (unsigned)min(l, (1 << 22) - 1)) : l;
        ids[r] = (int)r;
    }
}

// V [k,m] (reference layout) -> Vt [m,kp] (device layout), 32x32 tiles through LDS
__global__ void k_v_to_vt(const float *__restrict__ V, float *__restrict__ Vt, int k, int m, int kp) {
    __shared__ float tile[32][33];
    const int w0 = blockIdx.x * 32, z0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 256 threads: ty = 0..7
    for (int r = ty; r < 32; r += 8) {
        const int z = z0 + r, w = w0 + tx;
        tile[r][tx] = (z < k && w < m) ? V[(i64)z * m + w] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int w = w0 + r, z = z0 + tx;
        if (w < m && z < kp) Vt[(i64)w * kp + z] = tile[tx][r];
    }
}

__global__ void k_vt_to_v(const float *__restrict__ Vt, float *__restrict__ V, int k, int m, int kp) {
    __shared__ float tile[32][33];
    const int w0 = blockIdx.x * 32, z0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int r = ty; r < 32; r += 8) {
        const int w = w0 + r, z = z0 + tx;
        tile[r][tx] = (w < m && z < kp) ? Vt[(i64)w * kp + z] : 0.f;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int z = z0 + r, w = w0 + tx;
        if (z < k && w < m) V[(i64)z * m + w] = tile[tx][r];
    }
}

// plsa_upload_csr: the contract of include/plsa_hip.h ("indices must be < m, indptr non-decreasing") checked where the
// data already is -- one streaming pass over indptr and indices; bad |= 1 row pointers out of order / out of [0, nnz],
// bad |= 2 column index outside [0, m).  (The Python layer raises ValueError for the same; a C caller gets a status.)
__global__ void k_validate_csr(const int *__restrict__ indptr, const int *__restrict__ indices, i64 n, i64 m, i64 nnz,
                               int *__restrict__ bad) {
    int b = 0;
    const i64 stride = (i64)gridDim.x * blockDim.x;
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const int a = indptr[i], e = indptr[i + 1];
        if (a > e || a < 0 || (i64)e > nnz) b |= 1;
    }
    for (i64 j = (i64)blockIdx.x * blockDim.x + threadIdx.x; j < nnz; j += stride) {
        const int w = indices[j];
        if (w < 0 || (i64)w >= m) b |= 2;
    }
    if (b) atomicOr(bad, b);
}

// bootstrap (enstop_.py:87-88): out row i := base row idx[i]
__global__ void k_boot_lengths(const int *__restrict__ base_indptr, const i64 *__restrict__ idx,
                               i64 n_out, i64 n_base, int *__restrict__ lens, int *__restrict__ bad) {
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_out) {
        const i64 r = idx[i];
        if (r < 0 || r >= n_base) { lens[i] = 0; atomicOr(bad, 1); }
        else lens[i] = base_indptr[r + 1] - base_indptr[r];
    }
}

__global__ void k_boot_gather(const int *__restrict__ base_indptr, const int *__restrict__ base_col,
                              const float *__restrict__ base_val, const i64 *__restrict__ idx,
                              i64 n_out, const i64 *__restrict__ out_ptr64,
                              int *__restrict__ out_indptr, int *__restrict__ out_col,
                              float *__restrict__ out_val) {
    const int lane = threadIdx.x & 63;
    const i64 wid = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nw = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 i = wid; i <= n_out; i += nw) {
        if (lane == 0) out_indptr[i] = (int)out_ptr64[i];
        if (i == n_out) break;
        const i64 r = idx[i];
        const int s0 = base_indptr[r], len = base_indptr[r + 1] - s0;
        const i64 o0 = out_ptr64[i];
        for (int j = lane; j < len; j += 64) {
            out_col[o0 + j] = base_col[s0 + j];
            out_val[o0 + j] = base_val[s0 + j];
        }
    }
}

// CSC construction helpers
__global__ void k_iota(int *__restrict__ a, i64 nn) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nn; i += (i64)gridDim.x * blockDim.x)
        a[i] = (int)i;
}
// colptr[w] = first position of a key >= w in the sorted column keys (w = 0..m); no atomics -- an
// int-atomic histogram spent 12 ms on the Zipf head words' same-address contention
__global__ void k_colptr_from_sorted(const int *__restrict__ keys_sorted, i64 nnz, int m,
                                     int *__restrict__ colptr) {
    const i64 w = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (w > m) return;
    i64 lo = 0, hi = nnz;
    while (lo < hi) {
        const i64 mid = (lo + hi) >> 1;
        if (keys_sorted[mid] < (int)w) lo = mid + 1; else hi = mid;
    }
    colptr[w] = (int)lo;
}

// device-side factor initialisation for throughput runs (NOT the reference's MT19937 stream):
// counter-based uniform numbers, rows L1-normalised in place.  One group of 64 lanes per row.
__global__ void k_init_rows(float *__restrict__ A, i64 rows, int k, int kp, unsigned long long seed) {
    const int lane = threadIdx.x & 63;
    const i64 wid = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const i64 nw = ((i64)gridDim.x * blockDim.x) >> 6;
    for (i64 r = wid; r < rows; r += nw) {
        float part = 0.f;
        for (int z = lane; z < kp; z += 64) {
            float v = 0.f;
            if (z < k) {
                unsigned long long x = seed ^ ((unsigned long long)r * 0x9E3779B97F4A7C15ull + (unsigned long long)z);
                x += 0x9E3779B97F4A7C15ull; x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
                x = (x ^ (x >> 27)) * 0x94D049BB133111EBull; x ^= x >> 31;
                v = ((float)(x >> 40) + 0.5f) * (1.0f / 16777216.0f);
            }
            A[r * kp + z] = v;
            part += v;
        }
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        for (int z = lane; z < k; z += 64) A[r * kp + z] /= part;
    }
}
__global__ void k_csc_gather(const int *__restrict__ pos, const int *__restrict__ rowidx,
                             const float *__restrict__ vals, i64 nnz, int *__restrict__ csc_row,
                             float *__restrict__ csc_val) {
    for (i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x; i < nnz; i += (i64)gridDim.x * blockDim.x) {
        const int p = pos[i];
        csc_row[i] = rowidx[p];
        csc_val[i] = vals[p];
    }
}
__global__ void k_item_counts(const int *__restrict__ colptr, int m, int seg, int *__restrict__ cnt) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m) cnt[c] = (colptr[c + 1] - colptr[c] + seg - 1) / seg;
}
// item arrays + the first document of every item (sort key of the doc-band-major visiting order)
// Column items: a column is cut into chunks of <= seg entries (explicit [item_start, item_end) bounds).
__global__ void k_col_item_counts(const int *__restrict__ colptr, int m, int seg, int *__restrict__ cnt) {
    const i64 c = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (c < m) cnt[c] = (colptr[c + 1] - colptr[c] + seg - 1) / seg;
}

// Sort key of an item for the visiting order: (band of its first document, column length descending).  Band-major
// keeps the workgroups of an XCD inside one band of P(z|d) rows; inside a band the Zipf-head words go first (they
// touch every row of the band and pull it into the L2) and a chunk's 16 items are of one kind -- a chunk whose
// gathers all hit is not held up by one whose gathers all miss.  Measured at config 3 (64-entry items, balanced
// boundaries): plain first-document order 1.91 ms, bands of 512 / 1024 / 1536 / 2048 / 3072 / 4096 / 8192 documents
// 1.87 / 1.82 / 1.80 / 1.80 / 1.81 / 1.84 / 2.22 ms, ascending length inside a band 1.97 ms.  band <= 0: first document.
__global__ void k_item_fill(const int *__restrict__ colptr, const int *__restrict__ item_first, int m, int seg,
                            const int *__restrict__ csc_row, int *__restrict__ item_col,
                            int *__restrict__ item_start, int *__restrict__ item_end, int band, int len_bits,
                            unsigned long long *__restrict__ item_key, int *__restrict__ item_id, int n_items) {
    // one thread per ITEM (its column by binary search in item_first): a thread per column serialised the 15 k items
    // of a Zipf-head word in one lane (6.7 ms at config 3 against 0.05 ms)
    const i64 i = (i64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_items) return;
    int lo = 0, hi = m;                      // largest c with item_first[c] <= i
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (item_first[mid] <= (int)i) lo = mid; else hi = mid;
    }
    const int c = lo;
    const int st = colptr[c] + ((int)i - item_first[c]) * seg;
    item_col[i] = c;
    item_start[i] = st;
    item_end[i] = min(st + seg, colptr[c + 1]);
    const unsigned first = (unsigned)csc_row[st];
    // compact key (band index above `len_bits` bits of inverted column length): the radix sort then runs over the
    // significant bits only -- 3 passes instead of 8 at the 20NG shape, where the sort is a third of the structure build
    const unsigned long long inv_len = ((1ull << len_bits) - 1ull) - (unsigned long long)(colptr[c + 1] - colptr[c]);
    item_key[i] = band > 0 ? ((unsigned long long)(first / (unsigned)band) << len_bits) | inv_len
                           : (unsigned long long)first;
    item_id[i] = (int)i;
}

// ------------------------------------------------------------------------------------------------
// plsa_init(random) on the device with the REFERENCE'S random stream.  NumPy's legacy
// RandomState.rand() is MT19937: every double consumes two tempered 32-bit outputs,
// (a >> 5) * 2^26 + (b >> 6)) / 2^53.  k_mt19937_fill continues a given generator state (624 words
// + position) with one workgroup -- the recurrence is sequential across 624-word blocks but each
// block update is three data-parallel sweeps -- and leaves the advanced state behind so that the host
// generator can be set to exactly where the reference would be.  k_mt_init_v / k_mt_init_u turn the
// stream into the factors the way plsa.py:455-456, 510-511, 709-710 do: rand(k, m) first, then
// rand(n, k); float64 row sums accumulated left to right (enstop/utils.py:22-29), division, cast to
// float32 -- bit-identical to the host path.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ unsigned mt_temper(unsigned y) {
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}
__device__ __forceinline__ unsigned mt_mix(unsigned hi, unsigned lo, unsigned far) {
    const unsigned y = (hi & 0x80000000u) | (lo & 0x7fffffffu);
    return far ^ (y >> 1) ^ ((y & 1u) ? 0x9908b0dfu : 0u);
}

#ifndef PLSA_MT_THREADS
#define PLSA_MT_THREADS 256  // measured: 256 threads 142 ms, 128 threads 187 ms for 140 M words (config 3), one stream
#endif
// one 624-word block of the recurrence: o = current block, nw = next block (distinct LDS arrays).
// Three data-parallel sweeps -- the recurrence reaches back 227 words -- with one synchronisation
// each; on return every thread may read nw.
template <int T>
__device__ __forceinline__ void mt_next_block(const unsigned *o, unsigned *nw, int t) {
#pragma unroll
    for (int i = t; i < 227; i += T) nw[i] = mt_mix(o[i], o[i + 1], o[i + 397]);            // old words only
    __syncthreads();
#pragma unroll
    for (int i = t; i < 227; i += T) nw[i + 227] = mt_mix(o[i + 227], o[i + 228], nw[i]);   // new [0, 227)
    __syncthreads();
#pragma unroll
    for (int i = t; i < 169; i += T) nw[i + 454] = mt_mix(o[i + 454], o[i + 455], nw[i + 227]);   // new [227, 396)
    if (t == T - 1) nw[623] = mt_mix(o[623], nw[0], nw[396]);
    __syncthreads();
}

// Continues the generator whose 624-word block is states[0] (position pos0 inside it) for n_words
// outputs.  Workgroup r starts from states[r] -- the block r * blocks_per_stream blocks further on,
// placed there by k_mt_jump -- and produces the blocks r * blocks_per_stream + 1 ... of the stream;
// with n_streams = 1 this is the plain sequential generator.  The last workgroup leaves the
// advanced state (624 words + position) in final_state.
__global__ __launch_bounds__(PLSA_MT_THREADS) void k_mt19937_fill(const unsigned *__restrict__ states, unsigned *__restrict__ out,
                                                                 int pos0, i64 blocks_per_stream, i64 n_words, int n_streams,
                                                                 unsigned *__restrict__ final_state /*[625]*/) {
    constexpr int T = PLSA_MT_THREADS;
    __shared__ unsigned buf[2][624];
    const int t = threadIdx.x, r = blockIdx.x;
    for (int i = t; i < 624; i += T) buf[0][i] = states[(i64)r * 624 + i];
    int cur = 0;
    __syncthreads();
    const i64 first = min((i64)(624 - pos0), n_words);          // the rest of the current block
    if (r == 0)
        for (i64 i = t; i < first; i += T) out[i] = mt_temper(buf[0][pos0 + i]);
    int pos = pos0 + (int)first;
    const i64 total_blocks = (n_words - first + 623) / 624;
    const i64 jb = (i64)r * blocks_per_stream + 1, je = min(jb + blocks_per_stream - 1, total_blocks);
    for (i64 j = jb; j <= je; ++j) {
        mt_next_block<T>(buf[cur], buf[cur ^ 1], t);
        cur ^= 1;
        const i64 at = first + (j - 1) * 624;
        const i64 take = min((i64)624, n_words - at);
        for (i64 i = t; i < take; i += T) out[at + i] = mt_temper(buf[cur][i]);
        pos = (int)take;
        // the next block is written into the copy read two sweeps ago: every thread passed the third
        // synchronisation after its last read of that copy
    }
    if (r == n_streams - 1) {
        __syncthreads();
        for (int i = t; i < 624; i += T) final_state[i] = buf[cur][i];
        if (t == 0) final_state[624] = (unsigned)pos;
    }
}

// Jump-ahead (csrc/mt_jump.hpp): word j of the block J words further on is XOR_{i : g_i = 1} x[i + j]
// over the raw word sequence x of the source block.  blockIdx.x picks the source stream
// q = 2 * step * blockIdx.x and fills stream q + step (zeroed beforehand); blockIdx.y takes one
// slice of the polynomial's 624 coefficient words, regenerates the stretch of x it needs in LDS and
// XORs its share into the destination.
constexpr int MT_JUMP_SLICES = 16;
constexpr int MT_JUMP_GW = 624 / MT_JUMP_SLICES;                 // coefficient words per slice
constexpr int MT_JUMP_WIN = MT_JUMP_GW * 32 + 624;               // words of x a slice touches
__global__ __launch_bounds__(256) void k_mt_jump(const unsigned *__restrict__ g /*[624]*/, unsigned *__restrict__ states,
                                                 int step, int n_streams) {
    const int q = (int)blockIdx.x * 2 * step, dst = q + step;
    if (dst >= n_streams) return;
    __shared__ unsigned buf[2][624];
    __shared__ unsigned win[MT_JUMP_WIN + 8];
    __shared__ unsigned gw[MT_JUMP_GW];
    const int t = threadIdx.x;
    const int i0 = (int)blockIdx.y * MT_JUMP_GW * 32, i1 = i0 + MT_JUMP_WIN;
    for (int i = t; i < 624; i += 256) buf[0][i] = states[(i64)q * 624 + i];
    if (t < MT_JUMP_GW) gw[t] = g[blockIdx.y * MT_JUMP_GW + t];
    __syncthreads();
    int cur = 0;
    for (int base = 0;; base += 624) {
        for (int i = t; i < 624; i += 256) {
            const int gi = base + i;
            if (gi >= i0 && gi < i1) win[gi - i0] = buf[cur][i];
        }
        if (base + 624 >= i1) break;
        mt_next_block<256>(buf[cur], buf[cur ^ 1], t);
        cur ^= 1;
    }
    __syncthreads();
    unsigned a0 = 0, a1 = 0, a2 = 0;
    const bool third = t + 512 < 624;
    for (int w = 0; w < MT_JUMP_GW; ++w) {
        unsigned bits = gw[w];
        while (bits) {
            const unsigned *x = win + w * 32 + __builtin_ctz(bits);
            bits &= bits - 1;
            a0 ^= x[t];
            a1 ^= x[t + 256];
            if (third) a2 ^= x[t + 512];
        }
    }
    unsigned *d = states + (i64)dst * 624;
    atomicXor(d + t, a0);
    atomicXor(d + t + 256, a1);
    if (third) atomicXor(d + t + 512, a2);
}

__device__ __forceinline__ double mt_double(const unsigned *w, i64 idx) {
    const unsigned a = w[2 * idx] >> 5, b = w[2 * idx + 1] >> 6;
    return ((double)a * 67108864.0 + (double)b) / 9007199254740992.0;
}

// Sequential float64 marginal of each topic row, same rounding sequence as the reference's
// left-to-right sum (enstop/utils.py:24-29).  One wave per row: the 64 values of a chunk go through
// LDS so that every lane adds them in stream order (broadcast reads, one dependent add per value --
// the chain of m adds is the whole cost); chunks are prefetched four deep.  Padding adds +0.0,
// which leaves a non-negative sum unchanged.
__global__ __launch_bounds__(64) void k_mt_marginal_v(const unsigned *__restrict__ words, int k, int m,
                                                      double *__restrict__ marg) {
    constexpr int PF = 4;
    __shared__ double tile[2][64];
    const int z = blockIdx.x, lane = threadIdx.x;
    if (z >= k) return;
    const i64 base = (i64)z * m;
    double nxt[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const int w = p * 64 + lane;
        nxt[p] = w < m ? mt_double(words, base + w) : 0.0;
    }
    double s = 0.0;
    for (int w0 = 0; w0 < m; w0 += 64 * PF) {
        double cur[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) cur[p] = nxt[p];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const i64 w = (i64)w0 + 64 * PF + p * 64 + lane;
            nxt[p] = w < m ? mt_double(words, base + w) : 0.0;
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            if (w0 + p * 64 >= m) break;
            tile[p & 1][lane] = cur[p];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 64; ++i) s += tile[p & 1][i];
        }
    }
    if (lane == 0) marg[z] = s;
}

// ------------------------------------------------------------------------------------------------
// The same marginal WITHOUT the chain of m dependent adds (k_mt_marginal_v above: 1.9 ms for the 173 762 words of the
// 20NG shape, three quarters of a member's initialisation) -- bit for bit the same sequence of roundings.
//
// The draws are X * 2^-53 with integers X < 2^53 (numpy's random_sample), so while the running sum s stays inside one
// binade [2^e, 2^(e+1)), e >= 0, it is an integer multiple T of u = 2^(e-52) and one rounded addition is
//     T' = T + q + c,   q = X >> (e+1),  r = X mod 2^(e+1),   c = [r > half] or ([r == half] and (T + q) odd)
// (round to nearest even): the increment depends on T only through its PARITY.  A chunk of 64 draws therefore maps the
// parity at its start to a total increment -- a pair (A0, A1) that every chunk computes on its own (k_mt_chunk_pairs),
// for the binade its approximate prefix sum points at.  One wave per topic then walks the chunks: s += A[parity] is an
// integer addition on the bit pattern, checked on the spot (exponent of s equals the chunk's binade before AND after: no
// crossing inside, all 64 roundings were at that granularity); a chunk that fails the check -- the ~18 binade crossings,
// the first chunk (s < 1: exact additions), a wrong guess -- is added one draw at a time exactly as above.  The checks
// make the result independent of how good the guess is; the guess only decides how many chunks take the slow way.
// Validated against the sequential sum draw for draw by the NumPy-identity tests and a tie-heavy host prototype.
// ------------------------------------------------------------------------------------------------
constexpr int MT_SEQ_L = 64;      // draws per chunk = chunks per workgroup tile
typedef unsigned long long u64;
__device__ __forceinline__ u64 mt_int53(const unsigned *w, i64 idx) {
    return ((u64)(w[2 * idx] >> 5) << 26) | (u64)(w[2 * idx + 1] >> 6);
}

// tile[c][j] = draw j of chunk c0 + c of topic z (0 beyond the row): coalesced loads, conflict-free row reads
__device__ __forceinline__ void mt_load_tile(const unsigned *__restrict__ words, i64 base, int m, i64 c0,
                                             u64 (*tile)[MT_SEQ_L + 1]) {
    const int lane = threadIdx.x;
    for (int c = 0; c < MT_SEQ_L; ++c) {
        const i64 w = (c0 + c) * MT_SEQ_L + lane;
        tile[c][lane] = w < m ? mt_int53(words, base + w) : 0ull;
    }
    __syncthreads();
}

// exact integer sum of every chunk (< 2^59) and the approximate (float64) sum of every tile of 64 chunks:
// grid (ceil(nch / 64), k), 64 threads
__global__ __launch_bounds__(64) void k_mt_chunk_sums(const unsigned *__restrict__ words, int m, int nch,
                                                      u64 *__restrict__ csum, double *__restrict__ tsum) {
    __shared__ u64 tile[MT_SEQ_L][MT_SEQ_L + 1];
    const int z = blockIdx.y, lane = threadIdx.x;
    const i64 c0 = (i64)blockIdx.x * MT_SEQ_L;
    mt_load_tile(words, (i64)z * m, m, c0, tile);
    u64 t = 0;
#pragma unroll 8
    for (int j = 0; j < MT_SEQ_L; ++j) t += tile[lane][j];
    if (c0 + lane < nch) csum[(i64)z * nch + c0 + lane] = t;
    double tt = (c0 + lane < nch) ? (double)t : 0.0;
    for (int o = 32; o > 0; o >>= 1) tt += __shfl_xor(tt, o, 64);
    if (lane == 0) tsum[(i64)z * gridDim.x + blockIdx.x] = tt;
}

// per chunk: guessed binade e (from the approximate prefix sum of the chunk sums; -1: s < 1 or no guess) and the
// parity -> increment pair at that binade's granularity.  Same grid.
__global__ __launch_bounds__(64) void k_mt_chunk_pairs(const unsigned *__restrict__ words, int m, int nch,
                                                       const u64 *__restrict__ csum, const double *__restrict__ tsum,
                                                       u64 *__restrict__ pairs /*[2]*/, int *__restrict__ guess) {
    __shared__ u64 tile[MT_SEQ_L][MT_SEQ_L + 1];
    __shared__ double pre[MT_SEQ_L];
    const int z = blockIdx.y, lane = threadIdx.x;
    const i64 c0 = (i64)blockIdx.x * MT_SEQ_L;
    const u64 *cs = csum + (i64)z * nch;
    // approximate real prefix at this tile's first chunk -- the sums of the TILES in front of it (k_mt_chunk_sums; one
    // value per 4096 draws: linear in m per topic where summing the chunk sums was quadratic) -- then at each of its
    // chunks (any summation order will do: the walk checks every guess)
    double before = 0.0;
    const double *ts = tsum + (i64)z * gridDim.x;
    for (int t = lane; t < (int)blockIdx.x; t += 64) before += ts[t];
    for (int o = 32; o > 0; o >>= 1) before += __shfl_xor(before, o, 64);
    pre[lane] = (c0 + lane < nch) ? (double)cs[c0 + lane] : 0.0;
    mt_load_tile(words, (i64)z * m, m, c0, tile);     // (synchronises)
    double start = before;
    for (int j = 0; j < lane; ++j) start += pre[j];
    start *= 0x1p-53;
    const int e = start >= 1.0 ? (int)((__double_as_longlong(start) >> 52) & 0x7ff) - 1023 : -1;
    u64 a0 = 0, a1 = 0;
    if (e >= 0 && e <= 51) {
        const int sh = e + 1;
        const u64 mask = (1ull << sh) - 1, half = 1ull << (sh - 1);
        u64 t0 = 0, t1 = 1;
#pragma unroll 4
        for (int j = 0; j < MT_SEQ_L; ++j) {
            const u64 x = tile[lane][j], q = x >> sh, r = x & mask;
            const u64 up = r > half, tie = r == half;
            t0 += q + (up | (tie & ((t0 + q) & 1)));
            t1 += q + (up | (tie & ((t1 + q) & 1)));
        }
        a0 = t0; a1 = t1 - 1;
    }
    if (c0 + lane < nch) {
        const i64 o = (i64)z * nch + c0 + lane;
        pairs[2 * o] = a0; pairs[2 * o + 1] = a1;
        guess[o] = (e >= 0 && e <= 51) ? e : -1;
    }
}

// one wave per topic walks its chunks; marg[z] = the reference's left-to-right float64 sum.  The walk is wave-uniform:
// lane l fetches the pair and the guess of chunk c0 + l (the next 64 are requested before the current 64 are walked --
// they do not depend on the sum), v_readlane moves chunk j's values to scalar registers and the dependent chain per chunk
// is a handful of SCALAR instructions (parity, select, 64-bit add, two range checks); only the draw-by-draw fallback
// touches the vector units.
__device__ __forceinline__ u64 readlane64(u64 v, int l) {
    return (u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v & 0xffffffffull), l) |
           ((u64)(unsigned)__builtin_amdgcn_readlane((int)(unsigned)(v >> 32), l) << 32);
}
__global__ __launch_bounds__(64) void k_mt_marginal_walk(const unsigned *__restrict__ words, int m, int nch,
                                                         const u64 *__restrict__ pairs, const int *__restrict__ guess,
                                                         double *__restrict__ marg) {
    __shared__ double sx[MT_SEQ_L];
    const int z = blockIdx.x, lane = threadIdx.x;
    const i64 base = (i64)z * m;
    const u64 *pz = pairs + 2 * (i64)z * nch;
    const int *gz = guess + (i64)z * nch;
    u64 b = 0;      // bit pattern of the running sum (+0.0)
    u64 n0 = 0, n1 = 0;
    int ne = -1;
    if (lane < nch) { n0 = pz[2 * lane]; n1 = pz[2 * lane + 1]; ne = gz[lane]; }
    for (int c0 = 0; c0 < nch; c0 += 64) {
        const u64 v0 = n0, v1 = n1;
        const int ve = ne;
        const int cn = c0 + 64 + lane;
        if (cn < nch) { n0 = pz[2 * (i64)cn]; n1 = pz[2 * (i64)cn + 1]; ne = gz[cn]; }
        const int cnt = min(64, nch - c0);
#pragma unroll 4
        for (int j = 0; j < cnt; ++j) {
            const int e = __builtin_amdgcn_readlane(ve, j);
            const u64 a0 = readlane64(v0, j), a1 = readlane64(v1, j);      // (independent of the sum: issued ahead)
            // binade e = bit patterns [lo, lo + 2^52) with lo = (e + 1023) << 52: compared on the high words.  The sum may
            // take the pair iff it starts and ends inside
            const unsigned lo_hi = (unsigned)(e + 1023) << 20;
            const u64 nb = b + ((b & 1) ? a1 : a0);
            if (e >= 0 && (unsigned)(b >> 32) >= lo_hi && (unsigned)(nb >> 32) < lo_hi + (1u << 20)) { b = nb; continue; }
            // one draw at a time (uniform branch: every lane walks the same sum)
            const i64 w = (i64)(c0 + j) * MT_SEQ_L + lane;
            __syncthreads();
            sx[lane] = w < m ? mt_double(words, base + w) : 0.0;
            __syncthreads();
            double sd = __longlong_as_double((long long)b);
#pragma unroll 8
            for (int i = 0; i < MT_SEQ_L; ++i) sd += sx[i];
            b = readlane64((u64)__double_as_longlong(sd), 0);      // (uniform; back to scalar registers)
        }
    }
    if (lane == 0) marg[z] = __longlong_as_double((long long)b);
}

// V[z, w] = float32(draw / marginal_z) in the reference layout, for the transpose kernel
__global__ __launch_bounds__(256) void k_mt_scale_v(const unsigned *__restrict__ words, const double *__restrict__ marg,
                                                    int k, int m, float *__restrict__ V) {
    const i64 total = (i64)k * m;
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < total; i += (i64)gridDim.x * 256) {
        const double mg = marg[i / m];
        double v = mt_double(words, i);
        if (mg > 0.0) v /= mg;
        V[i] = (float)v;
    }
}

// 64 documents per wave: doubles [off + d*k, off + (d+1)*k) of the stream per document.  The draws
// are read coalesced (the 64 documents' values are one contiguous range) into an LDS tile of
// MT_U_COLS columns at a time, lane d adds document d's values in stream order, then the range is
// read again, divided and written as U[d, :] (pad columns zeroed) -- again coalesced.
constexpr int MT_U_COLS = 32;
__global__ __launch_bounds__(64) void k_mt_init_u(const unsigned *__restrict__ words, i64 off, i64 n, int k, int kp,
                                                  float *__restrict__ U) {
    __shared__ double tile[64][MT_U_COLS + 1];
    __shared__ double marg[64];
    const int lane = threadIdx.x;
    const i64 d0 = (i64)blockIdx.x * 64;
    const int docs = (int)min((i64)64, n - d0);
    double s = 0.0;
    for (int kc = 0; kc < k; kc += MT_U_COLS) {
        const int cols = min(MT_U_COLS, k - kc);
        const int cnt = docs * cols;
#pragma unroll 4
        for (int e = lane; e < cnt; e += 64) {
            const int r = e / cols, col = e - r * cols;
            tile[r][col] = mt_double(words, off + (d0 + r) * k + kc + col);
        }
        __syncthreads();
        if (lane < docs) {
#pragma unroll 8
            for (int i = 0; i < cols; ++i) s += tile[lane][i];
        }
        __syncthreads();
    }
    marg[lane] = s;
    __syncthreads();
    const int cnt = docs * kp;
#pragma unroll 4
    for (int e = lane; e < cnt; e += 64) {
        const int r = e / kp, z = e - r * kp;
        double v = 0.0;
        if (z < k) {
            v = mt_double(words, off + (d0 + r) * k + z);
            const double mg = marg[r];
            if (mg > 0.0) v /= mg;
        }
        U[d0 * kp + e] = (float)v;
    }
}

// streaming-bandwidth probes (measurement only): fill with plain / non-temporal stores, copy
template <bool NT>
__global__ __launch_bounds__(256) void k_probe_fill(float *__restrict__ p, i64 n4) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) {
        if (NT) st4_nt(p + i * 4, v); else st4(p + i * 4, v);
    }
}
// the E-step's store order: each wave writes `rows` consecutive 1-KB rows (one 16-B store per lane
// and row) before moving to its next tile
__global__ __launch_bounds__(256) void k_probe_fill_tiled(float *__restrict__ p, i64 n4, int rows) {
    const float4 v = make_float4(1.f, 2.f, 3.f, 4.f);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 tiles = n4 / (64 * (i64)rows);
    for (i64 t = (i64)blockIdx.x * 4 + wave; t < tiles; t += (i64)gridDim.x * 4) {
        float *base = p + (t * rows * 64 + lane) * 4;
        for (int r = 0; r < rows; ++r) st4_nt(base + (i64)r * 256, v);
    }
}
// read-only stream: every lane accumulates its float4s, one store per thread at the end
__global__ __launch_bounds__(256) void k_probe_read(const float *__restrict__ a, float *__restrict__ sink, i64 n4) {
    float4 acc = zero4();
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256) {
        const float4 v = ld4(a + i * 4);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
    if (acc.x + acc.y + acc.z + acc.w == -1.2345f) sink[0] = acc.x;   // keeps the loads alive
}
__global__ __launch_bounds__(256) void k_probe_copy(const float *__restrict__ a, float *__restrict__ b, i64 n4) {
    for (i64 i = (i64)blockIdx.x * 256 + threadIdx.x; i < n4; i += (i64)gridDim.x * 256)
        st4(b + i * 4, ld4(a + i * 4));
}

// ------------------------------------------------------------------------------------------------
// All-pairs Hellinger distance between topic vectors (the precomputed metric of the ensemble's topic
// combination, enstop/enstop_.py:258-266):  D_ij = sqrt(1 - sum_w sqrt(p_i[w] p_j[w]) / sqrt(|p_i|_1 |p_j|_1)).
//   k_hell_prepare : per topic, |p|_1 in float64 and sqrt() in place
//   k_hell_gram    : 64 x 64 tile of R R^T over one slice of the vocabulary (upper-triangular tiles
//                    only); float32 products, flushed into float64 every 256 words
//   k_hell_finish  : adds the slice partials in fixed order, forms the distance
// A plain dense contraction on the vector ALUs: t = n_runs * k is a few hundred, the whole job is tens
// of GFLOP -- milliseconds -- so no matrix-core path is warranted.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_hell_prepare(float *__restrict__ R, int t, i64 m, double *__restrict__ l1) {
    __shared__ double red[256];
    const int i = blockIdx.x;
    float *row = R + (i64)i * m;
    double s = 0.0;
    for (i64 w = threadIdx.x; w < m; w += 256) {
        const float p = row[w];
        s += (double)p;
        row[w] = sqrtf(fmaxf(p, 0.f));
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) l1[i] = red[0];
}

constexpr int HELL_TILE = 64;      // rows / columns of the output tile
constexpr int HELL_KSTEP = 32;     // words staged per step
__global__ __launch_bounds__(256) void k_hell_gram(const float *__restrict__ R, int t, i64 m, i64 slice,
                                                   const int *__restrict__ tile_i, const int *__restrict__ tile_j,
                                                   double *__restrict__ partial /*[slices][t][t]*/) {
    __shared__ float sa[HELL_KSTEP][HELL_TILE + 1], sb[HELL_KSTEP][HELL_TILE + 1];
    const int bi = tile_i[blockIdx.x] * HELL_TILE, bj = tile_j[blockIdx.x] * HELL_TILE;
    const i64 w0 = (i64)blockIdx.y * slice, w1 = min(m, w0 + slice);
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;          // 16 x 16 threads, 4 x 4 outputs each
    double acc[4][4];
    float facc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { acc[a][b] = 0.0; facc[a][b] = 0.f; }
    int since_flush = 0;
    for (i64 w = w0; w < w1; w += HELL_KSTEP) {
        // stage 64 rows x 32 words of both operands (each row segment is 128 contiguous bytes)
        for (int e = threadIdx.x; e < HELL_TILE * HELL_KSTEP; e += 256) {
            const int r = e / HELL_KSTEP, kk = e % HELL_KSTEP;
            const i64 ww = w + kk;
            sa[kk][r] = (bi + r < t && ww < w1) ? R[(i64)(bi + r) * m + ww] : 0.f;
            sb[kk][r] = (bj + r < t && ww < w1) ? R[(i64)(bj + r) * m + ww] : 0.f;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < HELL_KSTEP; ++kk) {
            float av[4], bv[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) { av[a] = sa[kk][ty * 4 + a]; bv[a] = sb[kk][tx * 4 + a]; }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) facc[a][b] += av[a] * bv[b];
        }
        __syncthreads();
        since_flush += HELL_KSTEP;
        if (since_flush >= 256) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { acc[a][b] += (double)facc[a][b]; facc[a][b] = 0.f; }
            since_flush = 0;
        }
    }
    double *out = partial + (i64)blockIdx.y * t * t;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = bi + ty * 4 + a, j = bj + tx * 4 + b;
            if (i < t && j < t) out[(i64)i * t + j] = acc[a][b] + (double)facc[a][b];
        }
}

__global__ __launch_bounds__(256) void k_hell_finish(const double *__restrict__ partial, int slices, int t,
                                                     const double *__restrict__ l1, double *__restrict__ D) {
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= (i64)t * t) return;
    const int i = (int)(e / t), j = (int)(e % t);
    const int a = min(i, j), b = max(i, j);                // only the upper triangle was computed
    double g = 0.0;
    for (int s = 0; s < slices; ++s) g += partial[(i64)s * t * t + (i64)a * t + b];
    const double li = l1[i], lj = l1[j];
    double d;
    if (i == j || (li == 0.0 && lj == 0.0)) d = 0.0;
    else if (li == 0.0 || lj == 0.0) d = 1.0;
    else d = sqrt(fmax(1.0 - g / sqrt(li * lj), 0.0));
    D[e] = d;
}

// ------------------------------------------------------------------------------------------------
// All-pairs KL divergence of the stacked ensemble topics, enstop/enstop_.py:234-253:
//   D[i, j] = sum over words with p_i > 0 and p_j > 0 of  p_i * (log2 p_i - log2 p_j)      (bits)
// Not symmetric: every (i, j) tile is computed.  Same staging as k_hell_gram; log2 (v_log_f32 IS a
// base-2 logarithm) is evaluated while a tile is staged into LDS, the pair term is formed directly
// (no cancellation between two large dot products), float partial sums are flushed into float64
// every 256 words.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_kl_gram(const float *__restrict__ T, int t, i64 m, i64 slice,
                                                 double *__restrict__ partial /*[slices][t][t]*/) {
    __shared__ float sp[HELL_KSTEP][HELL_TILE + 1], sl[HELL_KSTEP][HELL_TILE + 1];   // row side: p_i, log2 p_i
    __shared__ float sq[HELL_KSTEP][HELL_TILE + 1], sm[HELL_KSTEP][HELL_TILE + 1];   // column side: log2 p_j, [p_j > 0]
    const int nt = (t + HELL_TILE - 1) / HELL_TILE;
    const int bi = (int)(blockIdx.x / nt) * HELL_TILE, bj = (int)(blockIdx.x % nt) * HELL_TILE;
    const i64 w0 = (i64)blockIdx.y * slice, w1 = min(m, w0 + slice);
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    double acc[4][4];
    float facc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) { acc[a][b] = 0.0; facc[a][b] = 0.f; }
    int since_flush = 0;
    for (i64 w = w0; w < w1; w += HELL_KSTEP) {
        for (int e = threadIdx.x; e < HELL_TILE * HELL_KSTEP; e += 256) {
            const int r = e / HELL_KSTEP, kk = e % HELL_KSTEP;
            const i64 ww = w + kk;
            const float pi = (bi + r < t && ww < w1) ? T[(i64)(bi + r) * m + ww] : 0.f;
            const float pj = (bj + r < t && ww < w1) ? T[(i64)(bj + r) * m + ww] : 0.f;
            sp[kk][r] = pi > 0.f ? pi : 0.f;
            sl[kk][r] = pi > 0.f ? __log2f(pi) : 0.f;
            sq[kk][r] = pj > 0.f ? __log2f(pj) : 0.f;
            sm[kk][r] = pj > 0.f ? 1.f : 0.f;
        }
        __syncthreads();
#pragma unroll 4
        for (int kk = 0; kk < HELL_KSTEP; ++kk) {
            float pa[4], la[4], lb[4], mb[4];
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                pa[a] = sp[kk][ty * 4 + a]; la[a] = sl[kk][ty * 4 + a];
                lb[a] = sq[kk][tx * 4 + a]; mb[a] = sm[kk][tx * 4 + a];
            }
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) facc[a][b] += (pa[a] * mb[b]) * (la[a] - lb[b]);
        }
        __syncthreads();
        since_flush += HELL_KSTEP;
        if (since_flush >= 256) {
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) { acc[a][b] += (double)facc[a][b]; facc[a][b] = 0.f; }
            since_flush = 0;
        }
    }
    double *out = partial + (i64)blockIdx.y * t * t;
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const int i = bi + ty * 4 + a, j = bj + tx * 4 + b;
            if (i < t && j < t) out[(i64)i * t + j] = acc[a][b] + (double)facc[a][b];
        }
}

__global__ __launch_bounds__(256) void k_sum_slices(const double *__restrict__ partial, int slices, i64 count,
                                                    double *__restrict__ D) {
    const i64 e = (i64)blockIdx.x * 256 + threadIdx.x;
    if (e >= count) return;
    double g = 0.0;
    for (int s = 0; s < slices; ++s) g += partial[(i64)s * count + e];
    D[e] = g;
}

// ------------------------------------------------------------------------------------------------
// Cluster representatives of the topic combination, enstop/enstop_.py:299-308, 340-345, 385-393:
//   rep[c] = ( sum_{i in c} w_i sqrt(p_i) / sum_{i in c} w_i )^2, then L1-normalised
// (w == nullptr: plain mean).  members[] lists the topic rows of each cluster back to back
// (first[c] .. first[c+1]); rows are added in member order (fixed order: reproducible).
//   k_rep_accumulate : one thread per (cluster, word); per-block float64 partial of the row sum
//   k_rep_normalise  : fixed-order sum of the block partials, division, float32 output
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rep_accumulate(const float *__restrict__ T, i64 m,
                                                        const int *__restrict__ first, const int *__restrict__ members,
                                                        const double *__restrict__ w, double *__restrict__ rep /*[c][m]*/,
                                                        double *__restrict__ block_sums /*[c][gridDim.x]*/) {
    const int c = blockIdx.y;
    const int i0 = first[c], i1 = first[c + 1];
    const i64 word = (i64)blockIdx.x * 256 + threadIdx.x;
    double v = 0.0;
    if (word < m) {
        double num = 0.0, den = 0.0;
        for (int i = i0; i < i1; ++i) {
            const double wi = w ? w[members[i]] : 1.0;
            num += wi * (double)sqrtf(T[(i64)members[i] * m + word]);
            den += wi;
        }
        const double mean = den > 0.0 ? num / den : 0.0;
        v = mean * mean;
        rep[(i64)c * m + word] = v;
    }
    __shared__ double red[256];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) block_sums[(i64)c * gridDim.x + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void k_rep_normalise(const double *__restrict__ rep, i64 m,
                                                       const double *__restrict__ block_sums, int n_blocks,
                                                       float *__restrict__ out /*[c][m]*/) {
    const int c = blockIdx.y;
    __shared__ double total;
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int b = 0; b < n_blocks; ++b) s += block_sums[(i64)c * n_blocks + b];
        total = s;
    }
    __syncthreads();
    const i64 word = (i64)blockIdx.x * 256 + threadIdx.x;
    if (word < m) out[(i64)c * m + word] = (float)(rep[(i64)c * m + word] / total);
}

}  // namespace plsa
