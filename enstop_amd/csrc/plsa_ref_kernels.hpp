// plsa_ref_kernels.hpp -- the REFERENCE'S float32 arithmetic, rounding for rounding (PLSA_REFERENCE_SUMS / PLSA_REFERENCE_LL).
//
// The default kernels (plsa_kernels.hpp) add every corpus-long sum in a fixed order of their own, norm_pwz and the
// log-likelihood in float64: more accurate than the reference, and therefore 1e-2 away from the reference's OWN output
// from ~1e6 non-zeros on.  The kernels below evaluate the reference's statements with the
// reference's roundings instead -- every sum one float32 accumulator, added in the order the reference's loops add --
// so that a fit in this mode returns the bits `enstop/plsa.py` computes when its source is executed statement by
// statement (the fixtures under tests/golden, produced by the reference itself, are met bit for bit):
//
//   reference statement (enstop/plsa.py)                       here
//   :96-105  v = P(w|z) P(z|d); norm += v (z = 0 .. k-1);      k_ref_e_step_tiled  one lane per non-zero adds the products in
//            P(z|w,d) = v / norm                                                topic order (from an LDS tile); true division
//   :188     s = x * P(z|w,d)          (:294 t = s * sw[d])    every kernel     product rounded BEFORE it is added: the whole
//                                                                               file is compiled with fp contraction off
//   :190     p_w_given_z[z, w] += s    over nz = 0 .. nnz-1    k_ref_col_pass   a group owns a whole column; its entries in
//                                                                               document order (stable CSC) = COO order; long
//                                                                               columns: k_ref_norm_chain<GATHER>, one each
//   :191     p_z_given_d[d, z] += s                            k_ref_row_pass   a group owns a document, entries in order
//   :194     norm_pdz[d] += s          (entry-major, z-minor)  k_ref_row_pass   the row's k * len products added one by one
//                                                                               (through LDS, every lane of the group)
//   :193     norm_pwz[z] += s          over ALL nz             k_ref_pair_*     the chain's BITS without the chain: chunks ->
//                                                                               (parity -> increment) pairs, one checking walk
//                                                              k_ref_norm_chain ... or the chain itself (small corpora, the
//                                                                               fallback): lane = topic, a workgroup streams
//                                                                               x * P through LDS for its one adding wave
//   :196-202 division by the norms where positive              k_v_normalise (plsa_kernels.hpp) / k_ref_row_pass
//   :378-384 dot += P(w|z) P(z|d); result += x log(dot) sw     k_ref_ll_terms + k_ref_pair_* / k_ref_ll_chain (PLSA_REFERENCE_LL
//                                                                               only): one float32 running sum over all non-zeros
//
// A PARITY mode (sums in a prescribed order are the point): 1.8 / 4.3 / 9.4 / 35 ms per iteration at BASELINE config 1 / config 2 /
// the config-3 150 k sample / config 3 whole -- 20 ... 35 times the default arithmetic's; the numba-compiled reference takes ~490 /
// 1 900 / ~9 000 ms on the build container's 8 cores (DESIGN.md section 4 has the table and how each kernel got there).  Layouts are the engine's (U [n,kp], Vt [m,kp] word-major, P [nnz,kp], pad entries zero: a zero product
// adds +0.0, which changes no sum).
#pragma once
#include <hip/hip_runtime.h>
#include <hipcub/hipcub.hpp>
#include <stdint.h>

#pragma clang fp contract(off)   // s = x * p is rounded, THEN added (plsa.py:188-194): no fused multiply-add in this file

namespace plsa {
namespace ref {

typedef long long i64;

__device__ __forceinline__ void wave_lds_fence() {
    // LDS traffic of one wave executes in program order; this only keeps the COMPILER from moving accesses across
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
}

// ------------------------------------------------------------------------------------------------
// plsa.py:91-105.  One lane per non-zero: the topics are walked in order z = 0 .. k-1 with ONE float32
// norm (plsa.py:33 types it float32), kept products divided by it (true division) when it is positive.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ref_e_step(const int *__restrict__ rowidx, const int *__restrict__ colidx,
                                                    i64 nnz, const float *__restrict__ U, const float *__restrict__ Vt,
                                                    float *__restrict__ P, int kp, float thresh) {
    for (i64 nz = (i64)blockIdx.x * 256 + threadIdx.x; nz < nnz; nz += (i64)gridDim.x * 256) {
        const float4 *u = reinterpret_cast<const float4 *>(U + (i64)rowidx[nz] * kp);
        const float4 *v = reinterpret_cast<const float4 *>(Vt + (i64)colidx[nz] * kp);
        float4 *p = reinterpret_cast<float4 *>(P + nz * kp);
        const int q = kp >> 2;
        float norm = 0.0f;
        for (int c = 0; c < q; ++c) {
            const float4 a = v[c], b = u[c];
            const float t0 = a.x * b.x, t1 = a.y * b.y, t2 = a.z * b.z, t3 = a.w * b.w;
            if (t0 > thresh) norm += t0;      // plsa.py:97-100 (a pad product is 0: it adds +0.0 at most)
            if (t1 > thresh) norm += t1;
            if (t2 > thresh) norm += t2;
            if (t3 > thresh) norm += t3;
        }
        const bool pos = norm > 0.0f;         // plsa.py:104
        for (int c = 0; c < q; ++c) {
            const float4 a = v[c], b = u[c];
            float4 o;
            o.x = a.x * b.x; o.y = a.y * b.y; o.z = a.z * b.z; o.w = a.w * b.w;
            o.x = o.x > thresh ? o.x : 0.0f; o.y = o.y > thresh ? o.y : 0.0f;
            o.z = o.z > thresh ? o.z : 0.0f; o.w = o.w > thresh ? o.w : 0.0f;
            if (pos) { o.x = o.x / norm; o.y = o.y / norm; o.z = o.z / norm; o.w = o.w / norm; }
            p[c] = o;
        }
    }
}

// The same statements with the memory traffic of a streaming kernel.  The lane-per-non-zero form above reads its two factor rows
// and writes its row of P 16 bytes at a time from 64 different rows per instruction (and reads the factors twice): 0.8 TB/s.  Here a
// wave takes TJ = 64 / NZ non-zeros: (A) lane = topic, the factor rows loaded and the products formed, thresholded and written to
// an LDS tile row by row -- every access a whole row; (B) lane = non-zero, the tile read back TRANSPOSED (row stride kp + 1: no bank
// conflicts) and the norm summed over the topics in order, one float32, exactly the loop of plsa.py:96-100 (a product the threshold
// drops is stored as + 0.0: adding it changes nothing); (C) lane = topic again, every row divided by its norm (v_readlane) and
// stored, 256 bytes per instruction.
// SUMS: the wave also leaves the sums of its tile's addends t = x * P(z|w,d) [* sw[d]] per topic (tsum[tile][z], float32) -- what
// k_ref_pair_sums would read the whole of P again for (the chunk sums that guess the binades of the norm_pwz chain: any order, any
// precision will do for a guess).  The counts and weights are requested at the top of the tile: asked for where they are used, their
// latency stood exposed once per tile and the kernel took 1.8 times as long.
template <int NZ, bool SUMS>
__global__ __launch_bounds__(128) void k_ref_e_step_tiled(const int *__restrict__ rowidx, const int *__restrict__ colidx,
                                                          i64 nnz, const float *__restrict__ U, const float *__restrict__ Vt,
                                                          float *__restrict__ P, int kp, float thresh,
                                                          const float *__restrict__ vals, const float *__restrict__ sw,
                                                          float *__restrict__ tsum) {
    constexpr int TJ = 64 / NZ;                                  // non-zeros per wave tile
    constexpr int HB = TJ >= 32 ? 32 : TJ;                       // rows whose loads are in flight together (2 HB NZ <= 64 loads)
    extern __shared__ float e_lds[];                             // [2 waves][TJ][kp + 1]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int stride = kp + 1;
    float *tile = e_lds + wv * TJ * stride;
    const i64 n_tiles = (nnz + TJ - 1) / TJ;
    for (i64 ti = (i64)blockIdx.x * 2 + wv; ti < n_tiles; ti += (i64)gridDim.x * 2) {
        const i64 nz0 = ti * TJ;
        const int cnt = (int)min((i64)TJ, nnz - nz0);            // (uniform)
        const int jl = min(lane, cnt - 1);
        const int dl = rowidx[nz0 + jl], wl = colidx[nz0 + jl];  // lane j: the document and the word of non-zero j
        float xs = 0.0f, ws = 1.0f;                              // ... (SUMS) its count and its document's weight
        if (SUMS) {
            xs = vals[nz0 + jl];
            if (sw) ws = sw[dl];
        }
        float pr[TJ][NZ];
#pragma unroll
        for (int h = 0; h < TJ; h += HB) {
            float ua[HB][NZ], va[HB][NZ];
#pragma unroll
            for (int j = 0; j < HB; ++j) {
                const i64 d = __builtin_amdgcn_readlane(dl, h + j), w = __builtin_amdgcn_readlane(wl, h + j);
#pragma unroll
                for (int q = 0; q < NZ; ++q) {
                    const int z = lane + 64 * q, zz = z < kp ? z : 0;
                    ua[j][q] = U[d * kp + zz];
                    va[j][q] = Vt[w * kp + zz];
                }
            }
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < HB; ++j)
#pragma unroll
                for (int q = 0; q < NZ; ++q) {
                    const int z = lane + 64 * q;
                    float t = va[j][q] * ua[j][q];               // plsa.py:96
                    t = t > thresh ? t : 0.0f;                   // plsa.py:97 / :101-102
                    pr[h + j][q] = t;
                    if (z < kp) tile[(h + j) * stride + z] = t;
                }
        }
        wave_lds_fence();
        float norm = 0.0f;
        {
            const float *row = tile + (lane < TJ ? lane : 0) * stride;
#pragma unroll 8
            for (int z = 0; z < kp; ++z) norm += row[z];         // plsa.py:98-100: one float32 sum, z ascending
        }
        wave_lds_fence();
        float ts[NZ];
#pragma unroll
        for (int q = 0; q < NZ; ++q) ts[q] = 0.0f;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            const float nj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(norm), j));
            if (j < cnt) {                                       // (uniform)
                const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xs), j));
                const float w = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(ws), j));
#pragma unroll
                for (int q = 0; q < NZ; ++q) {
                    const int z = lane + 64 * q;
                    const float pz = nj > 0.0f ? pr[j][q] / nj : pr[j][q];                        // plsa.py:104-105
                    if (z < kp) P[(nz0 + j) * kp + z] = pz;
                    if (SUMS) {
                        float t = x * pz;                        // plsa.py:188
                        if (sw) t = t * w;                       // plsa.py:294
                        ts[q] += t;
                    }
                }
            }
        }
        if (SUMS) {
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                const int z = lane + 64 * q;
                if (z < kp) tsum[ti * kp + z] = ts[q];
            }
        }
    }
}

// chunk sums from the E-step's tile sums: csum[z][c] = the sum of the chunk's tiles (tiles_per_chunk = PAIR_L / TJ)
__global__ __launch_bounds__(256) void k_ref_pair_sums_from_tiles(const float *__restrict__ tsum, int kp, int tiles_per_chunk, i64 n_tiles,
                                                                  i64 n_chunks, i64 n_pad, double *__restrict__ csum) {
    const i64 total = n_chunks * kp;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 c = idx / kp;
        const int z = (int)(idx - c * kp);
        double s2 = 0.0;
        for (int i = 0; i < tiles_per_chunk; ++i) {
            const i64 tile = c * tiles_per_chunk + i;
            if (tile < n_tiles) s2 += (double)tsum[tile * kp + z];
        }
        csum[(i64)z * n_pad + c] = s2;
    }
}

// ------------------------------------------------------------------------------------------------
// Document half of the M-step, plsa.py:188, 191, 194, 200-202 (and the refit M-step, :806-814).  A group of G lanes
// owns a document; lane li holds the topics z = li + G t (t < NZ).  Per entry: s = x * P(z|w,d) (rounded), the
// lane's own sums p_z_given_d[d, z] += s, and norm_pdz[d] += s for z = 0 .. k-1 IN THAT ORDER: the group's products go
// through LDS and every lane adds them one by one (all lanes of a group carry the same chain).
// ------------------------------------------------------------------------------------------------
template <int G, int NZ>
__global__ __launch_bounds__(256) void k_ref_row_pass(const int *__restrict__ indptr, const float *__restrict__ vals,
                                                      int n, const int *__restrict__ row_order,
                                                      const float *__restrict__ P, float *__restrict__ U_new,
                                                      float *__restrict__ norm_pdz_out, int kp) {
    extern __shared__ float s_lds[];          // [256 / G][kp]
    constexpr int GPB = 256 / G;
    const int li = threadIdx.x % G, gid = threadIdx.x / G;
    float *mine = s_lds + gid * kp;
    const float4 *mine4 = reinterpret_cast<const float4 *>(mine);
    const int q = kp >> 2;
    for (i64 r = (i64)blockIdx.x * GPB + gid; r < n; r += (i64)gridDim.x * GPB) {
        const int d = row_order ? row_order[r] : (int)r;
        const int j0 = indptr[d], j1 = indptr[d + 1];
        float acc[NZ];
#pragma unroll
        for (int t = 0; t < NZ; ++t) acc[t] = 0.0f;
        float npdz = 0.0f;
        // software pipeline: the next entry's count and P(z|w,d) values are requested before the current entry's chain
        float xn = 0.0f, pn[NZ];
#pragma unroll
        for (int t = 0; t < NZ; ++t) pn[t] = 0.0f;
        if (j0 < j1) {
            xn = vals[j0];
#pragma unroll
            for (int t = 0; t < NZ; ++t) { const int z = li + G * t; if (z < kp) pn[t] = P[(i64)j0 * kp + z]; }
        }
        for (int j = j0; j < j1; ++j) {
            const float x = xn;
            float pc[NZ];
#pragma unroll
            for (int t = 0; t < NZ; ++t) pc[t] = pn[t];
            if (j + 1 < j1) {
                xn = vals[j + 1];
#pragma unroll
                for (int t = 0; t < NZ; ++t) { const int z = li + G * t; if (z < kp) pn[t] = P[(i64)(j + 1) * kp + z]; }
            }
#pragma unroll
            for (int t = 0; t < NZ; ++t) {
                const int z = li + G * t;
                if (z < kp) {
                    const float s = x * pc[t];        // plsa.py:188
                    acc[t] += s;                      // plsa.py:191
                    mine[z] = s;
                }
            }
            wave_lds_fence();
            for (int c = 0; c < q; ++c) {             // plsa.py:194, z ascending
                const float4 v = mine4[c];
                npdz += v.x; npdz += v.y; npdz += v.z; npdz += v.w;
            }
            wave_lds_fence();
        }
        if (norm_pdz_out && li == 0) norm_pdz_out[d] = npdz;
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            const int z = li + G * t;
            if (z < kp) U_new[(i64)d * kp + z] = npdz > 0.0f ? acc[t] / npdz : acc[t];   // plsa.py:200-202
        }
    }
}

// The document half with the tile of the E-step above: a wave takes TJ = 64 / NZ documents (neighbours in the length-sorted order);
// per entry slot e: (A) lane = topic -- for every document that has an e-th entry its row of P is loaded whole, s = x * P(z|w,d)
// formed, added to the document's own accumulators (registers, lane = topic) and parked in the LDS tile, a document past its end
// parks + 0.0; (B) lane = document -- norm_pdz[d] += s over the tile row, z ascending: the reference's chain (entry-major, z-minor),
// ONE lane per document adding it where the group kernel above has all its lanes carry the same chain, sixteen documents' worth of
// dependent additions per wave instruction instead of four (config 3 whole: 22 ms at 1.1 TB/s).
template <int NZ>
__global__ __launch_bounds__(128) void k_ref_row_pass_tiled(const int *__restrict__ indptr, const float *__restrict__ vals,
                                                            int n, const int *__restrict__ row_order,
                                                            const float *__restrict__ P, float *__restrict__ U_new,
                                                            float *__restrict__ norm_pdz_out, int kp) {
    constexpr int TJ = 64 / NZ;
    extern __shared__ float r_lds[];                             // [2 waves][TJ][kp + 1]
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int stride = kp + 1;
    float *tile = r_lds + wv * TJ * stride;
    const i64 n_tiles = ((i64)n + TJ - 1) / TJ;
    for (i64 ti = (i64)blockIdx.x * 2 + wv; ti < n_tiles; ti += (i64)gridDim.x * 2) {
        const i64 r0 = ti * TJ;
        const int cnt = (int)min((i64)TJ, (i64)n - r0);          // (uniform)
        // lane j < cnt: its document, first entry, length
        const bool has = lane < cnt;
        const int dj = has ? (row_order ? row_order[r0 + lane] : (int)(r0 + lane)) : 0;
        const int j0 = has ? indptr[dj] : 0;
        const int len = has ? indptr[dj + 1] - j0 : 0;
        int maxlen = len;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) maxlen = max(maxlen, __shfl_xor(maxlen, o));
        float acc[TJ][NZ];
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int q = 0; q < NZ; ++q) acc[j][q] = 0.0f;
        float npdz = 0.0f;
        const float *row = tile + (lane < TJ ? lane : 0) * stride;
        // software pipeline over the entry slots: slot e + 1's rows and counts are requested before slot e's chain is added
        float pn[TJ][NZ], xn;
        auto request = [&](int e, float (&pv)[TJ][NZ], float &xl) {
            xl = e < len ? vals[j0 + e] : 0.0f;                  // lane = document: the count of its e-th entry
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const int lj = __builtin_amdgcn_readlane(len, j);
                const i64 at = (i64)(__builtin_amdgcn_readlane(j0, j) + (e < lj ? e : 0)) * kp;   // (a finished document: its first row
#pragma unroll                                                                                    //  again -- cached, never used)
                for (int q = 0; q < NZ; ++q) {
                    const int z = lane + 64 * q;
                    pv[j][q] = P[at + (z < kp ? z : 0)];
                }
            }
        };
        request(0, pn, xn);
        for (int e = 0; e < maxlen; ++e) {
            float pv[TJ][NZ];
            const float xl = xn;
#pragma unroll
            for (int j = 0; j < TJ; ++j)
#pragma unroll
                for (int q = 0; q < NZ; ++q) pv[j][q] = pn[j][q];
            request(e + 1 < maxlen ? e + 1 : e, pn, xn);
            asm volatile("" ::: "memory");
#pragma unroll
            for (int j = 0; j < TJ; ++j) {
                const bool live = e < __builtin_amdgcn_readlane(len, j);                          // (uniform)
                if (live) {
                    const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xl), j));
#pragma unroll
                    for (int q = 0; q < NZ; ++q) {
                        const int z = lane + 64 * q;
                        const float sv = x * pv[j][q];           // plsa.py:188
                        acc[j][q] += sv;                         // plsa.py:191
                        if (z < kp) tile[j * stride + z] = sv;
                    }
                } else if (e == __builtin_amdgcn_readlane(len, j)) {                              // just finished: its tile row becomes + 0.0
#pragma unroll
                    for (int q = 0; q < NZ; ++q) {
                        const int z = lane + 64 * q;
                        if (z < kp) tile[j * stride + z] = 0.0f;
                    }
                }
            }
            wave_lds_fence();
#pragma unroll 8
            for (int z = 0; z < kp; ++z) npdz += row[z];         // plsa.py:194, z ascending (a finished document adds + 0.0)
            wave_lds_fence();
        }
        if (norm_pdz_out && has) norm_pdz_out[dj] = npdz;
#pragma unroll
        for (int j = 0; j < TJ; ++j) {
            if (j < cnt) {                                       // (uniform)
                const float nj = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(npdz), j));
                const i64 d = __builtin_amdgcn_readlane(dj, j);
#pragma unroll
                for (int q = 0; q < NZ; ++q) {
                    const int z = lane + 64 * q;
                    if (z < kp) U_new[d * kp + z] = nj > 0.0f ? acc[j][q] / nj : acc[j][q];       // plsa.py:200-202
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Vocabulary half, plsa.py:190 (296 with weights): p_w_given_z[z, w] += s over the non-zeros in COO order.  For one
// word those are its column's entries in document order -- what the stable CSC holds -- so a group that owns the WHOLE
// column (no items, no partial sums) and adds entry by entry reproduces the reference's accumulator bit for bit.
// Un-normalised sums -> Vacc [m, kp]; B rows of P are in flight per group.  Columns of heavy_min entries and more are left to
// k_ref_norm_chain<.., GATHER> (one workgroup per column: ~6 ns per entry where this walk costs ~160).
// ------------------------------------------------------------------------------------------------
template <int G, int NZ>
__global__ __launch_bounds__(256) void k_ref_col_pass(const int *__restrict__ colptr, const int *__restrict__ csc_row,
                                                      const float *__restrict__ csc_val, const int *__restrict__ csc_pos,
                                                      int m, const float *__restrict__ P, const float *__restrict__ sw,
                                                      float *__restrict__ Vacc, int kp, int heavy_min) {
    constexpr int GPB = 256 / G;
    constexpr int B = NZ <= 2 ? 16 : 8;         // rows of P in flight per group
    const int li = threadIdx.x % G, gid = threadIdx.x / G;
    for (i64 w = (i64)blockIdx.x * GPB + gid; w < m; w += (i64)gridDim.x * GPB) {
        const int j0 = colptr[w], j1 = colptr[w + 1];
        if (j1 - j0 >= heavy_min) continue;     // a long column: a workgroup of its own (k_ref_norm_chain<.., GATHER>)
        float acc[NZ];
#pragma unroll
        for (int t = 0; t < NZ; ++t) acc[t] = 0.0f;
        if (j0 >= j1) {
#pragma unroll
            for (int t = 0; t < NZ; ++t) { const int z = li + G * t; if (z < kp) Vacc[w * kp + z] = 0.0f; }
            continue;
        }
        // The Zipf-head columns are the long pole (one group walks up to n entries): the entry records (position in CSR order,
        // count, document weight) of the NEXT batch are requested while the current batch's rows of P are in flight, so a batch
        // costs one memory latency, not the two of "index, then the row it points to" (first version: 5.4 / 42 / 63 ms at
        // config 1 / config 2 / the config-3 150 k sample).
        i64 pos_n[B];
        float x_n[B], wd_n[B];
        auto load_records = [&](int jb) {
#pragma unroll
            for (int b = 0; b < B; ++b) {
                const int j = min(jb + b, j1 - 1);
                pos_n[b] = csc_pos[j];
                x_n[b] = csc_val[j];
                wd_n[b] = sw ? sw[csc_row[j]] : 1.0f;
            }
        };
        load_records(j0);
        for (int jb = j0; jb < j1; jb += B) {
            float p[B][NZ], x[B], wd[B];
#pragma unroll
            for (int b = 0; b < B; ++b) {
                x[b] = x_n[b]; wd[b] = wd_n[b];
#pragma unroll
                for (int t = 0; t < NZ; ++t) {
                    const int z = li + G * t;
                    p[b][t] = z < kp ? P[pos_n[b] * kp + z] : 0.0f;
                }
            }
            load_records(min(jb + B, j1 - 1));     // (past the end: the last entry again, never added)
#pragma unroll
            for (int b = 0; b < B; ++b) {
                if (jb + b < j1) {
#pragma unroll
                    for (int t = 0; t < NZ; ++t) {
                        float s = x[b] * p[b][t];          // plsa.py:188
                        if (sw) s = s * wd[b];             // plsa.py:294
                        acc[t] += s;                       // plsa.py:190 / :296
                    }
                }
            }
        }
#pragma unroll
        for (int t = 0; t < NZ; ++t) {
            const int z = li + G * t;
            if (z < kp) Vacc[w * kp + z] = acc[t];
        }
    }
}

// the columns with heavy_min entries or more (any order)
__global__ void k_ref_heavy_cols(const int *__restrict__ colptr, int m, int heavy_min, int *__restrict__ list, int *__restrict__ count) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < m && colptr[w + 1] - colptr[w] >= heavy_min) list[atomicAdd(count, 1)] = w;
}

// ------------------------------------------------------------------------------------------------
// norm_pwz[z] += s over ALL non-zeros, plsa.py:193 (:299 with weights): one float32 accumulator per topic and a chain
// of nnz dependent additions -- the statement that carries the reference 1e-2 away from exact arithmetic at 3 M
// non-zeros, and the one a drop-in has to reproduce to land on the reference's own numbers.
//
// ONE workgroup of 1024 lanes.  All sixteen waves stream t = (x * P(z|w,d)) [* sw[d]] -- tiles of CHAIN_TILE floats
// of the P array, DEPTH tiles ahead in registers -- into a double-buffered LDS tile; wave 0 alone walks each tile row by
// row, lane = topic (z = lane + 64 t), one dependent v_add_f32 per row and chain.  One barrier per tile: the tile being
// refilled is the one wave 0 finished before it arrived at the previous barrier.
// ------------------------------------------------------------------------------------------------
constexpr int CHAIN_THREADS = 1024;
constexpr int CHAIN_TILE = 8192;                       // floats per LDS tile (32 KB; two tiles)
constexpr int CHAIN_PRODUCERS = 768;                   // lanes that stream tiles in: the waves NOT on the adding wave's SIMD
constexpr int CHAIN_F4 = (CHAIN_TILE / 4 + CHAIN_PRODUCERS - 1) / CHAIN_PRODUCERS;   // float4 per producer lane and tile (3)
constexpr int CHAIN_DEPTH = 4;                         // tiles in flight in registers

// The adding wave (wave 0) is bound by its own instruction stream -- one wave: every instruction costs four cycles -- so
//  * a tile row sits in LDS at a COMPILE-TIME stride of 64 NZ floats whatever kp is: the row loop is nothing but ds_read_b32 with
//    immediate offsets and the dependent v_add_f32 (lanes beyond kp read padding into accumulators nobody stores);
//  * wave 0 does nothing else, and the waves that share its SIMD (4, 8, 12: waves of a workgroup go round the four SIMDs)
//    only keep the barriers: the twelve others stream the tiles in;
//  * the producers keep the RAW data of CHAIN_DEPTH tiles in registers (P values, count, weight) and form the products when a
//    tile is written to LDS, with branch-free loads from per-tile scalar bases + loop-invariant 32-bit lane offsets: a wave only
//    ever waits for its OLDEST loads (s_waitcnt vmcnt(n > 0)) and spends ~17 instructions per float4.
// History on the GPU (ms per EM iteration in the reference arithmetic, config 1 / config 2 / config-3 150 k sample; this kernel
// runs beside the document and column passes and is the longer pole): run-time LDS stride + predicated reads 63 / 240 / 447;
// compile-time stride 37 / 132 / 262; loads CHAIN_DEPTH tiles ahead 29 / 98 / 162; dedicated adding wave with its LDS reads one
// batch ahead of its adds ~21 / 66 / 110 (the kernel alone: 19 / 64 / 101 ms = 6.4 ns per non-zero; a dependent v_add_f32 issues
// every ~8 cycles, the sequential likelihood chain below runs at 3.7 ns per term).
//
// GATHER (the long columns of the vocabulary half, plsa.py:190): the same chain over ONE COLUMN's entries -- workgroup b owns column
// heavy[b]; `vals` / `rowidx` are the CSC's counts / documents from the column's first entry on (contiguous), the row of P an entry
// multiplies is found through `pos` (its position in COO order), and the sums go to Vacc[w].  A load that depends on a load in a pipe
// whose counter retires in order: the positions are requested 2 CHAIN_DEPTH tiles ahead, the rows they point to CHAIN_DEPTH tiles
// ahead, so that waiting for a position never means waiting for the rows requested after it.
template <int NZ, bool HAS_SW, bool GATHER = false>
__global__ __launch_bounds__(CHAIN_THREADS) void k_ref_norm_chain(const int *__restrict__ rowidx,
                                                                  const float *__restrict__ vals, i64 nnz,
                                                                  const float *__restrict__ P,
                                                                  const float *__restrict__ sw, int kp,
                                                                  float *__restrict__ norm_pwz,
                                                                  const int *__restrict__ pos = nullptr,
                                                                  const int *__restrict__ heavy = nullptr,
                                                                  const int *__restrict__ colptr = nullptr) {
    constexpr int STRIDE = 64 * NZ;                            // floats between two rows of a tile in LDS
    constexpr int ROWS = CHAIN_TILE / STRIDE;                  // rows per tile (128 / NZ)
    __shared__ float4 tile[2][CHAIN_TILE / 4];
    if (GATHER) {                                              // (uniform: scalar registers)
        const int w = heavy[blockIdx.x], j0 = colptr[w];
        nnz = colptr[w + 1] - j0;
        rowidx += j0; vals += j0; pos += j0;
        norm_pwz += (i64)w * kp;
    }
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // (wave-uniform by construction: scalar branches below)
    const bool producer = (wave & 3) != 0;                     // waves 1-3, 5-7, 9-11, 13-15
    const int ptid = (wave - 1 - (wave >> 2)) * 64 + (tid & 63);   // 0 .. 767 among the producers
    const int kq = kp >> 2;
    const int f4_per_tile = ROWS * kq;                         // float4 of P one tile covers (<= 2048)
    const i64 n_tiles = (nnz + ROWS - 1) / ROWS;
    // a producer lane's float4 slots inside a tile (the same for every tile): validity, tile-relative row, offset into the
    // tile's part of P (floats), LDS position
    bool sval[CHAIN_F4];
    int srow[CHAIN_F4], soff[CHAIN_F4], spos[CHAIN_F4], scol[CHAIN_F4];
#pragma unroll
    for (int s = 0; s < CHAIN_F4; ++s) {
        const int f = ptid + CHAIN_PRODUCERS * s;
        sval[s] = producer && f < f4_per_tile;
        srow[s] = sval[s] ? f / kq : 0;
        soff[s] = sval[s] ? 4 * f : 0;
        scol[s] = soff[s] - srow[s] * kp;                      // float offset inside the row of P
        spos[s] = srow[s] * (STRIDE / 4) + (sval[s] ? f - srow[s] * kq : 0);
    }
    float4 rp_[CHAIN_DEPTH][CHAIN_F4];
    float rx_[CHAIN_DEPTH][CHAIN_F4], rw_[CHAIN_DEPTH][CHAIN_F4];
    int rq_[CHAIN_DEPTH][CHAIN_F4];                            // GATHER: positions, 2 CHAIN_DEPTH tiles ahead
    // rows of tile t that exist (0 beyond the corpus); the loads of a tile beyond the end go to the last tile (valid addresses)
    auto rows_of = [&](i64 t) { return (int)max((i64)0, min((i64)ROWS, nnz - t * ROWS)); };
    auto fetch_pos = [&](i64 t, int (&dq)[CHAIN_F4]) {         // (GATHER)
        const i64 row0 = min(t, n_tiles - 1) * ROWS;
        const int rows = t < n_tiles ? rows_of(t) : 0;
#pragma unroll
        for (int s = 0; s < CHAIN_F4; ++s) dq[s] = pos[row0 + (sval[s] && srow[s] < rows ? srow[s] : 0)];
    };
    auto fetch = [&](i64 t, float4 (&dp)[CHAIN_F4], float (&dx)[CHAIN_F4], float (&dw)[CHAIN_F4], const int (&dq)[CHAIN_F4]) {
        const i64 row0 = min(t, n_tiles - 1) * ROWS;           // (uniform: scalar registers)
        const int rows = t < n_tiles ? rows_of(t) : 0;
        const float *Pt = P + row0 * kp;
        const float *xt = vals + row0;
        const int *rt = rowidx + row0;
#pragma unroll
        for (int s = 0; s < CHAIN_F4; ++s) {
            const bool ok = sval[s] && srow[s] < rows;
            if (GATHER) dp[s] = *reinterpret_cast<const float4 *>(P + (i64)dq[s] * kp + scol[s]);   // (dq: a valid entry's position always)
            else dp[s] = *reinterpret_cast<const float4 *>(Pt + (ok ? soff[s] : 0));
            dx[s] = xt[ok ? srow[s] : 0];
            dw[s] = HAS_SW ? sw[rt[ok ? srow[s] : 0]] : 1.0f;
        }
    };
    auto products = [&](i64 t, const float4 &p, float x, float wd, int s) {
        const bool ok = sval[s] && srow[s] < rows_of(t);
        float4 o;
        o.x = x * p.x; o.y = x * p.y; o.z = x * p.z; o.w = x * p.w;                         // plsa.py:188
        if (HAS_SW) { o.x = o.x * wd; o.y = o.y * wd; o.z = o.z * wd; o.w = o.w * wd; }     // plsa.py:294
        o.x = ok ? o.x : 0.0f; o.y = ok ? o.y : 0.0f; o.z = ok ? o.z : 0.0f; o.w = ok ? o.w : 0.0f;
        return o;
    };
    // THREE loops, one per role, each free of inner branches (so that the producers' waits stay "oldest loads only"); every wave
    // passes the same number of barriers: one per tile, the tile count rounded up to a multiple of CHAIN_DEPTH.
    const i64 n_padded = (n_tiles + CHAIN_DEPTH - 1) / CHAIN_DEPTH * CHAIN_DEPTH;
    float acc[NZ];
#pragma unroll
    for (int t = 0; t < NZ; ++t) acc[t] = 0.0f;
    if (wave == 0) {
        // the adding wave: tile t is complete in LDS after barrier t; the producers refill that buffer after barrier t + 1 at
        // the earliest, which this wave reaches only when it is done with tile t
        for (i64 t = 0; t < n_padded; ++t) {
            __syncthreads();
            const float *rp = reinterpret_cast<const float *>(tile[t & 1]) + tid;
            const int rows = rows_of(t);
            // software pipeline: the LDS reads of the NEXT batch of UN rows are in flight while this batch is added (one batch at
            // a time left ~130 cycles of LDS latency exposed per 8 rows: 19 cycles = 9 ns per row; this kernel took 26 / 89 / 137 ms
            // per iteration at config 1 / config 2 / the config-3 150 k sample)
            constexpr int UN = NZ >= 8 ? 2 : (NZ == 4 ? 4 : 8);
            float va[UN][NZ], vb[UN][NZ];
            auto load = [&](float (&v)[UN][NZ], int r0) {
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int c = 0; c < NZ; ++c) v[u][c] = rp[(r0 + u) * STRIDE + 64 * c];
            };
            auto add = [&](const float (&v)[UN][NZ]) {
#pragma unroll
                for (int u = 0; u < UN; ++u)
#pragma unroll
                    for (int c = 0; c < NZ; ++c) acc[c] += v[u][c];          // plsa.py:193
            };
            // ALL rows of the tile, unguarded: the producers write every row of every tile -- products, or +0.0 for the rows past
            // the end of the corpus, and x + 0.0 == x -- so the loop has a compile-time trip count and the chain is one v_add_f32
            // per row and topic (a per-row `r < rows` test put a v_cndmask behind every add: twice the dependent chain)
            (void)rows;
            load(va, 0);
#pragma unroll 2
            for (int r = 0; r < ROWS; r += 2 * UN) {
                load(vb, r + UN);
                add(va);
                if (r + 2 * UN < ROWS) load(va, r + 2 * UN);
                add(vb);
            }
        }
    } else if (!producer) {
        for (i64 t = 0; t < n_padded; ++t) __syncthreads();          // waves 4, 8, 12: they would share the adding wave's SIMD
    } else {
        if (GATHER) {
#pragma unroll
            for (int dd = 0; dd < CHAIN_DEPTH; ++dd) fetch_pos(dd, rq_[dd]);
        }
#pragma unroll
        for (int dd = 0; dd < CHAIN_DEPTH; ++dd) {
            fetch(dd, rp_[dd], rx_[dd], rw_[dd], rq_[dd]);
            if (GATHER) fetch_pos(dd + CHAIN_DEPTH, rq_[dd]);
        }
        for (i64 t0 = 0; t0 < n_padded; t0 += CHAIN_DEPTH) {
#pragma unroll
            for (int dd = 0; dd < CHAIN_DEPTH; ++dd) {       // static register indices; tile t = t0 + dd
                const i64 t = t0 + dd;                       // (a tile past the end holds zeros and has no rows)
                float4 *buf = tile[dd & 1];                  // CHAIN_DEPTH is even: buffer parity == parity of t
#pragma unroll
                for (int s = 0; s < CHAIN_F4; ++s)
                    if (sval[s]) buf[spos[s]] = products(t, rp_[dd][s], rx_[dd][s], rw_[dd][s], s);
                __syncthreads();
                fetch(t + CHAIN_DEPTH, rp_[dd], rx_[dd], rw_[dd], rq_[dd]);
                if (GATHER) fetch_pos(t + 2 * CHAIN_DEPTH, rq_[dd]);
            }
        }
    }
    if (tid < 64) {
#pragma unroll
        for (int c = 0; c < NZ; ++c) {
            const int z = tid + 64 * c;
            if (z < kp) norm_pwz[z] = acc[c];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same chain WITHOUT nnz dependent additions: norm_pwz[z] from per-chunk (parity -> increment) pairs, the technique of
// k_mt_chunk_pairs / k_mt_marginal_walk (plsa_kernels.hpp; there: float64 sums of 53-bit draws) carried over to float32 sums of
// arbitrary non-negative float32 addends.
//
// While the running sum S stays inside one binade [2^e, 2^(e+1)) it is an integer multiple M of u = 2^(e-23), 2^23 <= M < 2^24, and
// one rounded addition of t = mt * 2^(Et - 150) (mt the 24-bit significand, Et the biased exponent, 1 for denormals) is
//     M' = M + q + c,    sh = E - Et,  q = mt >> sh,  r = mt mod 2^sh,  c = [r > 2^(sh-1)] or ([r == 2^(sh-1)] and (M + q) odd)
// (round to nearest even; sh >= 1: an addend from S's own binade or above pushes S out of it; sh >= 26: t < u / 2 changes
// nothing): the increment depends on M only through its PARITY.  A chunk of PAIR_L addends therefore maps the parity at its start to
// a total increment -- a pair (T0, T1) every chunk computes ON ITS OWN (k_ref_pair_build) for the binade the chunk's approximate
// real prefix sum points at AND for a neighbouring binade: the one below (the float32 chain drifts down from the real sum: a few
// per cent at BASELINE sizes, a third at the whole of config 3), the one above when the prefix lies within ~12 % of the upper edge.
// One wave per 64 topics then walks the chunks, lane = topic
// (k_ref_pair_walk): S's bit pattern takes the pair of the candidate binade that IS S's binade, iff the pair is valid and M + T stays
// below 2^24 (no crossing inside the chunk: T only grows); otherwise -- the ~25 binade crossings per topic, the first chunk (S = 0), a
// drift beyond the window, a negative or non-finite addend -- the addends of that chunk are added one by one with real float32
// additions (all lanes load the rows, only the lanes that need it add).  The checks make the result independent of the guesses: the
// bits are those of the sequential chain above (tests: every bitwise test of the mode runs through this path, and both paths are
// compared on tie-heavy dyadic inputs); the guesses only decide how many chunks take the slow way (counted, reported, and a context
// whose walks fail on more than a quarter of their chunks goes back to the serial chain).  Two refinements below: the pairs of 16
// consecutive chunks COMPOSE into one (k_ref_pair_compose: the walk steps through groups and descends into a group only where it has no
// valid pair), and a chunk's pairs are computed by the adder itself (two proxy sums per binade, k_ref_pair_build).
// ------------------------------------------------------------------------------------------------
// Addends per chunk: a kernel argument (a multiple of 64).  A walk step costs ~115 ns whatever the length, a chunk that goes the
// slow way ~22 ns per addend, and the number of such chunks hardly depends on the corpus (~25 binade crossings per topic, most of them
// shared by the 64 topics of a group: ~600 chunks per walk at k = 64).  With the groups of PAIR_R chunks the length is 256 everywhere;
// walking chunks only (PLSA_REF_LEVELS=1) the best length grows with sqrt(nnz) and the host takes 1024 from 48 M non-zeros on
// (config-3 sample, 15 M: 64 / 128 / 256 -> the four kernels together 39.9 / 27.5 / 24.5 ms in their first form; config 3 whole,
// 100 M: the walk 49 ms at 256, 24 ms at 1024, 9 ms with the groups).
constexpr int PAIR_L_SMALL = 256, PAIR_L_LARGE = 1024;
constexpr long long PAIR_L_LARGE_FROM = 48000000;
constexpr int PAIR_SC = 8;            // chunks a wave handles back to back (8 consecutive float64 chunk sums per lane: one 64-B line)
constexpr unsigned PAIR_INVALID = 0x7F000000u;                    // a total no valid pair holds (those stay below 2^25): M + T >= 2^24 by itself
constexpr unsigned PAIR_NO_BINADE = 0x7FFFu;                      // exps field "no candidate": equals no biased exponent
constexpr unsigned PAIR_NOOP = 0x80000000u;                       // exps bit: every addend of the chunk is zero (any sum stays what it is)

// The addend of (row, z).  KIND 0 / 1: t exactly as every other kernel of this file forms it, without / with sample weights.
// KIND 2 (the sequential likelihood, plsa.py:322 / :384: ONE chain, kp = 1, P = the terms k_ref_ll_terms wrote): MINUS the term --
// the terms are x * log(p) <= 0 wherever p <= 1, the pairs want non-negative addends, and rounding to nearest even is symmetric:
// -(a + b) == (-a) + (-b) bit for bit, so the chain of the negated terms is the negated chain (a positive term -- p > 1, which the
// reference's unnormalised topics do produce -- is a negative addend here: that chunk goes the slow way).
constexpr int PAIR_PLAIN = 0, PAIR_WEIGHTED = 1, PAIR_NEG_TERMS = 2;
template <int KIND>
__device__ __forceinline__ float pair_addend(const float *__restrict__ P, const float *__restrict__ vals,
                                             const int *__restrict__ rowidx, const float *__restrict__ sw, i64 row, int kp, int z) {
    if (KIND == PAIR_NEG_TERMS) return -P[row];
    float t = vals[row] * P[row * kp + z];                    // plsa.py:188
    if (KIND == PAIR_WEIGHTED) t = t * sw[rowidx[row]];       // plsa.py:294
    return t;
}

// SB addends of a chunk for this lane's NZ topics, the loads all issued before the first product is formed (the compiler left to itself
// multiplies each value as it arrives and keeps a handful of loads in flight: both kernels below ran at 0.8 TB/s that way).
// rows: the chunk's rows that exist (>= 1); rows beyond repeat the last one (the callers skip them).
template <int NZ, int KIND, int SB>
__device__ __forceinline__ void pair_addend_batch(const float *__restrict__ Pc, const float *__restrict__ xc, const int *__restrict__ dc,
                                                  const float *__restrict__ sw, int kp, int lane, int j0, int rows, float (&t)[SB][NZ]) {
    float xv[SB], wv[SB];
#pragma unroll
    for (int u = 0; u < SB; ++u) {
        const int jj = min(j0 + u, rows - 1);
        xv[u] = KIND != PAIR_NEG_TERMS ? xc[jj] : 1.0f;
        wv[u] = KIND == PAIR_WEIGHTED ? sw[dc[jj]] : 1.0f;
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            const int z = lane + 64 * q;
            t[u][q] = Pc[jj * kp + (z < kp ? z : 0)];
        }
    }
    asm volatile("" ::: "memory");
#pragma unroll
    for (int u = 0; u < SB; ++u)
#pragma unroll
        for (int q = 0; q < NZ; ++q) {
            float v = KIND != PAIR_NEG_TERMS ? xv[u] * t[u][q] : -t[u][q];     // plsa.py:188 (: 322)
            if (KIND == PAIR_WEIGHTED) v = v * wv[u];                          // plsa.py:294
            t[u][q] = v;
        }
}
constexpr int pair_batch(int nz) { return nz >= 8 ? 4 : (nz == 4 ? 8 : (nz == 2 ? 16 : 32)); }

// float64 sum of every chunk's addends per topic: csum[z][c], chunk index fastest (n_pad chunks per topic)
template <int NZ, int KIND>
__global__ __launch_bounds__(256) void k_ref_pair_sums(const int *__restrict__ rowidx, const float *__restrict__ vals, i64 nnz,
                                                       const float *__restrict__ P, const float *__restrict__ sw, int kp, int PAIR_L,
                                                       i64 n_chunks, i64 n_pad, double *__restrict__ csum) {
    constexpr int SB = pair_batch(NZ);
    const int lane = threadIdx.x & 63;
    const i64 wid = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((i64)gridDim.x * blockDim.x) >> 6;
    const i64 n_super = (n_chunks + PAIR_SC - 1) / PAIR_SC;
    for (i64 sc = wid; sc < n_super; sc += nw) {
        for (int c8 = 0; c8 < PAIR_SC; ++c8) {
            const i64 c = sc * PAIR_SC + c8;
            double sum[NZ];
#pragma unroll
            for (int q = 0; q < NZ; ++q) sum[q] = 0.0;
            if (c < n_chunks) {                                  // (uniform)
                const i64 row0 = c * PAIR_L;
                const int rows = (int)min((i64)PAIR_L, nnz - row0);
                const float *Pc = P + row0 * kp, *xc = vals + row0;
                const int *dc = rowidx + row0;
                for (int j0 = 0; j0 < rows; j0 += SB) {
                    float t[SB][NZ];
                    pair_addend_batch<NZ, KIND, SB>(Pc, xc, dc, sw, kp, lane, j0, rows, t);
#pragma unroll
                    for (int u = 0; u < SB; ++u)
                        if (j0 + u < rows) {                     // (uniform)
#pragma unroll
                            for (int q = 0; q < NZ; ++q) sum[q] += (double)t[u][q];
                        }
                }
            }
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                const int z = lane + 64 * q;
                if (z < kp) csum[(i64)z * n_pad + c] = sum[q];
            }
        }
    }
}

// csum[z][:] -> its exclusive prefix sums, in place; one workgroup per topic, tiles of 2048 chunks (8 consecutive sums per lane, a
// block-wide scan of the lanes' totals, a running carry).  Any summation order will do: the walk checks every guess.  (First
// version: every lane scanned a slab of its own, then added up the slabs in front of it: 4 ms for the 392 k chunks of config 3.)
__global__ __launch_bounds__(256) void k_ref_pair_prefix(double *__restrict__ csum, i64 n_chunks, i64 n_pad) {
    typedef hipcub::BlockScan<double, 256> Scan;
    __shared__ typename Scan::TempStorage tmp;
    double *a = csum + (i64)blockIdx.x * n_pad;
    double carry = 0.0;
    for (i64 base = 0; base < n_chunks; base += 2048) {
        const i64 i0 = base + (i64)threadIdx.x * 8;
        double v[8], tot = 0.0;
#pragma unroll
        for (int u = 0; u < 8; ++u) { v[u] = i0 + u < n_chunks ? a[i0 + u] : 0.0; tot += v[u]; }
        double before, tile_total;
        Scan(tmp).ExclusiveSum(tot, before, tile_total);
        __syncthreads();
        double run = carry + before;
#pragma unroll
        for (int u = 0; u < 8; ++u) { if (i0 + u < n_chunks) a[i0 + u] = run; run += v[u]; }
        carry += tile_total;
    }
}

// binade candidates of a chunk from its approximate prefix: biased float32 exponents, 0 = none
__device__ __forceinline__ void pair_candidates(double prefix, int &ea, int &eb) {
    ea = eb = 0;
    const float pf = (float)prefix;
    if (!(pf > 0.0f)) return;
    const unsigned b = __float_as_uint(pf);
    const int e = (int)(b >> 23);
    if (e < 1 || e > 252) return;
    ea = e;
    const unsigned frac = b & 0x7FFFFFu;                          // position inside the binade: 0 .. 2^23
    // the second candidate: the float32 chain drifts DOWN from the real sum (an addend below half a unit of the sum is lost, and at
    // the whole of config 3 the reference's norms end ~35 % short of the real ones), so the binade below -- except right under the
    // upper edge, where ties may have carried S across first
    eb = frac > (7u << 20) ? e + 1 : e - 1;
    if (eb < 1 || eb > 252) eb = 0;
}

// The pairs of a chunk are computed BY THE ADDER: inside binade E the chain's increments depend on the sum only through the parity of
// its significand, so two proxy sums -- 2^E (M = 2^23, even) and 2^E (1 + 2^-23) (odd) -- are carried through the chunk's addends with
// plain v_add_f32; while a proxy stays in the binade its significand has moved by exactly T0 (T1).  A proxy that leaves the binade
// (an addend from the sum's own binade or above, 2^23 steps, an infinity, a NaN) or a negative addend makes the pair invalid.  (The
// first version evaluated q, r, the half-way test and the tie rule in integer arithmetic: ~45 instructions per addend and candidate
// pair where this takes 4 additions; same pairs.)
// per (chunk, topic): the two candidate binades and their (T0, T1); pairs[c][z] = {T0_A, T1_A, T0_B, T1_B}, exps[c][z] = E_A | E_B << 16 [| PAIR_NOOP]
template <int NZ, int KIND>
__global__ __launch_bounds__(256) void k_ref_pair_build(const int *__restrict__ rowidx, const float *__restrict__ vals, i64 nnz,
                                                        const float *__restrict__ P, const float *__restrict__ sw, int kp, int PAIR_L,
                                                        i64 n_chunks, i64 n_pad, const double *__restrict__ prefix,
                                                        uint4 *__restrict__ pairs, unsigned *__restrict__ exps) {
    constexpr int SB = pair_batch(NZ);
    const int lane = threadIdx.x & 63;
    const i64 wid = ((i64)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((i64)gridDim.x * blockDim.x) >> 6;
    const i64 n_super = (n_chunks + PAIR_SC - 1) / PAIR_SC;
    for (i64 sc = wid; sc < n_super; sc += nw) {
        for (int c8 = 0; c8 < PAIR_SC; ++c8) {
            const i64 c = sc * PAIR_SC + c8;
            if (c >= n_chunks) break;                            // (uniform)
            int ea[NZ], eb[NZ];
            float a0[NZ], a1[NZ], b0[NZ], b1[NZ];                // the proxies: candidate A / B, starting parity 0 / 1
            unsigned sgn[NZ], nzb[NZ];                           // OR of the addends' bits (sign) / of their bits without the sign
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                const int z = lane + 64 * q;
                pair_candidates(prefix[(i64)(z < kp ? z : 0) * n_pad + c], ea[q], eb[q]);
                a0[q] = __uint_as_float((unsigned)ea[q] << 23); a1[q] = __uint_as_float(((unsigned)ea[q] << 23) | 1u);
                b0[q] = __uint_as_float((unsigned)eb[q] << 23); b1[q] = __uint_as_float(((unsigned)eb[q] << 23) | 1u);
                sgn[q] = nzb[q] = 0u;
            }
            const i64 row0 = c * PAIR_L;
            const int rows = (int)min((i64)PAIR_L, nnz - row0);
            const float *Pc = P + row0 * kp, *xc = vals + row0;
            const int *dc = rowidx + row0;
            for (int j0 = 0; j0 < rows; j0 += SB) {
                float t[SB][NZ];
                pair_addend_batch<NZ, KIND, SB>(Pc, xc, dc, sw, kp, lane, j0, rows, t);
#pragma unroll
                for (int u = 0; u < SB; ++u)
                    if (j0 + u < rows) {                         // (uniform)
#pragma unroll
                        for (int q = 0; q < NZ; ++q) {
                            const float v = t[u][q];
                            a0[q] = a0[q] + v; a1[q] = a1[q] + v; b0[q] = b0[q] + v; b1[q] = b1[q] + v;      // plsa.py:193, four times
                            sgn[q] |= __float_as_uint(v); nzb[q] |= __float_as_uint(v) << 1;
                        }
                    }
            }
#pragma unroll
            for (int q = 0; q < NZ; ++q) {
                const int z = lane + 64 * q;
                if (z < kp) {
                    // a chunk of + 0.0 addends leaves ANY sum as it is (padding topics, topics the E-step threshold emptied, stretches of
                    // zero responsibilities); a negative addend (-0.0 included: cheap, and only a negative weight makes one) rules both out
                    const bool all_zero = nzb[q] == 0u && (sgn[q] >> 31) == 0u, neg = (sgn[q] >> 31) != 0u;
                    const unsigned ua0 = __float_as_uint(a0[q]), ua1 = __float_as_uint(a1[q]);
                    const unsigned ub0 = __float_as_uint(b0[q]), ub1 = __float_as_uint(b1[q]);
                    const bool okA = ea[q] != 0 && !neg && (ua0 >> 23) == (unsigned)ea[q] && (ua1 >> 23) == (unsigned)ea[q];
                    const bool okB = eb[q] != 0 && !neg && (ub0 >> 23) == (unsigned)eb[q] && (ub1 >> 23) == (unsigned)eb[q];
                    uint4 o;
                    o.x = okA ? (ua0 & 0x7FFFFFu) : PAIR_INVALID; o.y = okA ? (ua1 & 0x7FFFFFu) - 1u : PAIR_INVALID;
                    o.z = okB ? (ub0 & 0x7FFFFFu) : PAIR_INVALID; o.w = okB ? (ub1 & 0x7FFFFFu) - 1u : PAIR_INVALID;
                    pairs[c * kp + z] = o;
                    exps[c * kp + z] = (okA ? (unsigned)ea[q] : PAIR_NO_BINADE) | ((okB ? (unsigned)eb[q] : PAIR_NO_BINADE) << 16) |
                                       (all_zero ? PAIR_NOOP : 0u);
                }
            }
        }
    }
}

// GROUPS of PAIR_R consecutive chunks: parity maps compose -- (p -> p + T_a[p]) then (q -> q + T_b[q]) is again a map from the starting
// parity to a total increment, for as long as every chunk of the group has a valid pair for the SAME binade (a chunk of zeros fits any).
// The walk then takes one step per group, PAIR_R times fewer, and falls back to the group's chunks where the group has no valid pair
// (a crossing inside it, chunks that guessed different binades).  Candidates: those of the group's first chunk that has any.
constexpr int PAIR_R = 16;
__global__ __launch_bounds__(256) void k_ref_pair_compose(const uint4 *__restrict__ pairs, const unsigned *__restrict__ exps, int kp,
                                                          i64 n_chunks, i64 n_groups, uint4 *__restrict__ pairs2,
                                                          unsigned *__restrict__ exps2) {
    const i64 total = n_groups * kp;
    for (i64 idx = (i64)blockIdx.x * 256 + threadIdx.x; idx < total; idx += (i64)gridDim.x * 256) {
        const i64 g = idx / kp;
        const int z = (int)(idx - g * kp);
        const i64 first = g * PAIR_R;
        const int cnt = (int)min((i64)PAIR_R, n_chunks - first);
        uint4 r[PAIR_R];
        unsigned x[PAIR_R];
#pragma unroll
        for (int i = 0; i < PAIR_R; ++i) {
            const i64 at = min(first + i, n_chunks - 1) * kp + z;
            r[i] = pairs[at];
            x[i] = i < cnt ? exps[at] : (PAIR_NOOP | PAIR_NO_BINADE | PAIR_NO_BINADE << 16);
        }
        unsigned cand[2] = {PAIR_NO_BINADE, PAIR_NO_BINADE};
        bool all_noop = true, have = false;
#pragma unroll
        for (int i = 0; i < PAIR_R; ++i) {
            const bool noop = (x[i] >> 31) != 0u;
            all_noop = all_noop && noop;
            const unsigned a = x[i] & 0xFFFFu, b = (x[i] >> 16) & 0x7FFFu;
            if (!have && !noop && (a != PAIR_NO_BINADE || b != PAIR_NO_BINADE)) { cand[0] = a; cand[1] = b; have = true; }
        }
        unsigned out[4];
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) {
            const unsigned e = cand[k2];
            bool ok = e != PAIR_NO_BINADE;
            unsigned inc0 = 0u, inc1 = 0u;
#pragma unroll
            for (int i = 0; i < PAIR_R; ++i) {
                if ((x[i] >> 31) != 0u) continue;                  // zeros: any sum stays what it is
                const bool isA = (x[i] & 0xFFFFu) == e, isB = ((x[i] >> 16) & 0x7FFFu) == e;
                const unsigned t0 = isA ? r[i].x : r[i].z, t1 = isA ? r[i].y : r[i].w;
                ok = ok && (isA || isB) && t0 != PAIR_INVALID;
                if (ok) {
                    inc0 += (inc0 & 1u) ? t1 : t0;
                    inc1 += ((inc1 + 1u) & 1u) ? t1 : t0;
                    if ((inc0 | inc1) >> 25) ok = false;
                }
            }
            out[2 * k2] = ok ? inc0 : PAIR_INVALID;
            out[2 * k2 + 1] = ok ? inc1 : PAIR_INVALID;
            if (!ok) cand[k2] = PAIR_NO_BINADE;
        }
        uint4 o; o.x = out[0]; o.y = out[1]; o.z = out[2]; o.w = out[3];
        pairs2[idx] = o;
        exps2[idx] = cand[0] | (cand[1] << 16) | (all_noop ? PAIR_NOOP : 0u);
    }
}

// The walk: one wave per 64 topics follows the chains, lane = topic, the sum kept as (biased exponent, 24-bit significand); per
// chunk ~30 integer instructions: pick the candidate whose binade IS the sum's, pick the total for the significand's parity, add,
// check M + T < 2^24.  Workgroup b owns topics 64 b .. 64 b + 63 (chains of different topics never meet: k = 1000 walks in sixteen
// workgroups side by side).  Like k_ref_norm_chain a workgroup is one walking wave and its suppliers -- eight waves: six stream the chunk records -- WALK_DEPTH tiles
// ahead in registers, then a double-buffered LDS tile -- wave 0 walks, the wave that shares wave 0's SIMD only keeps the barriers.
// A chunk that fails the check for any topic of the group is added addend by addend with real float32 additions (all lanes load the
// rows, only the lanes that need it add).  stats[0] += chunks that went that way, stats[1] += chunks, per workgroup.
// History (ms per call at config 1 / config 2 / the config-3 150 k sample, the serial chain: 18.9 / 64.1 / 100.3): the walking wave
// fetching its own records four chunks ahead 12.2 / 44.6 / 78.1 (264 ns per chunk = HBM latency / 4); records staged through LDS
// 7.3 / 25.2 / 46.2 (the slow way loaded one row per HBM latency: ~30 us per slow chunk, 0.3-0.5 % of the chunks); its loads 32 rows
// at a time 4.3 / 15.9 / 30.1 (every step began with s_waitcnt lgkmcnt(0) -- the slow path inside the loop is a join -- and the loop
// unrolled over the tile carried sixteen copies of it: ~100 KB of code); the form below, see the table in DESIGN.md section 4.
constexpr int WALK_THREADS = 512;                      // eight waves, two per SIMD: 256 registers each (sixteen waves leave 128, and the
                                                       // producers' tiles in flight then went through copies that waited for every load)
constexpr int WALK_PRODUCERS = 384;                    // waves 1-3 and 5-7
constexpr int WALK_DEPTH = 6;                          // tiles in flight in the producers' registers (even)
constexpr int WALK_TC = 16;                            // chunks per LDS tile: 16 KB of pairs + 4 KB of exponents, two tiles
constexpr int WALK_SLOTS = (WALK_TC * 64 + WALK_PRODUCERS - 1) / WALK_PRODUCERS;     // records per producer lane and tile (3)

template <int KIND, bool TWO>
__global__ __launch_bounds__(WALK_THREADS) void k_ref_pair_walk(const int *__restrict__ rowidx, const float *__restrict__ vals, i64 nnz,
                                                                 const float *__restrict__ P, const float *__restrict__ sw, int kp, int PAIR_L,
                                                                 i64 n_chunks, const uint4 *__restrict__ pairs_,
                                                                 const unsigned *__restrict__ exps, float *__restrict__ norm_pwz,
                                                                 unsigned long long *__restrict__ stats,
                                                                 i64 n_chunks1, const uint4 *__restrict__ pairs1_,
                                                                 const unsigned *__restrict__ exps1) {
    // TWO: the records streamed and walked are those of GROUPS of PAIR_R chunks (k_ref_pair_compose; n_chunks = groups); a group that
    // fails its check is walked chunk by chunk from the chunks' own records (pairs1_ / exps1, n_chunks1 chunks of PAIR_L addends),
    // and only a chunk that fails there is added addend by addend.
    // (a native vector type: arrays of HIP's uint4 struct are copied with memcpy and then stay in scratch memory)
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const u32x4 *__restrict__ pairs = reinterpret_cast<const u32x4 *>(pairs_);
    constexpr int TC = WALK_TC;
    __shared__ u32x4 trec[2][TC * 64];
    __shared__ unsigned texp[2][TC * 64];
    __shared__ u32x4 mid_rec[TWO ? PAIR_R * 64 : 1];               // the walking wave's own: a failed group's chunk records
    __shared__ unsigned mid_exp[TWO ? PAIR_R * 64 : 1];
    const u32x4 *__restrict__ pairs1 = reinterpret_cast<const u32x4 *>(pairs1_);
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = (wave & 3) != 0;
    const int ptid = (wave - 1 - (wave >> 2)) * 64 + (tid & 63);
    const int z0 = 64 * (int)blockIdx.x, width = min(64, kp - z0);   // this workgroup's topics
    const i64 n_tiles = (n_chunks + TC - 1) / TC;
    const i64 n_padded = (n_tiles + WALK_DEPTH - 1) / WALK_DEPTH * WALK_DEPTH;
    if (wave == 0) {
        unsigned es = 0u, m = 0u;                                // the sum: + 0.0
        const bool mine = tid < width;
        const int z = z0 + (mine ? tid : 0);
        unsigned long long slow = 0;
        for (i64 t = 0; t < n_padded; ++t) {
            __syncthreads();
            const u32x4 *rp = trec[t & 1] + tid;
            const unsigned *xp = texp[t & 1] + tid;
            u32x4 ra, rb, rc;
            unsigned xa, xb, xc;
            bool need;
            // one record into the sum of the lanes in `active`: true for the lanes it could not be applied to
            auto apply = [&](const u32x4 &r, unsigned x, bool active) {
                const bool useA = es == (x & 0xFFFFu), useB = es == ((x >> 16) & 0x7FFFu);
                const unsigned t0 = useA ? r.x : r.z, t1 = useA ? r.y : r.w;
                const unsigned mn = m + ((m & 1u) ? t1 : t0);
                const bool fast = active && (useA || useB) && mn < 0x1000000u;
                m = fast ? mn : m;
                return !fast && (x >> 31) == 0u && active;
            };
            auto fast_step = [&](const u32x4 &r, unsigned x) {
                need = apply(r, x, mine);
                return __any(need) != 0;
            };
            auto slow_chunk = [&](i64 c1, bool need) {           // rare: binade crossings, the first chunk, a drifted guess
                ++slow;
                const i64 row0 = c1 * PAIR_L;                    // (a chunk past the end is a no-op record: never here)
                const int rows = (int)min((i64)PAIR_L, nnz - row0);
                float sum = __uint_as_float(es ? (es << 23) | (m & 0x7FFFFFu) : m);
                const float *Pc = P + row0 * kp + z, *xc = vals + row0;
                const int *dc = rowidx + row0;
                if (rows == PAIR_L) {
                    // a whole chunk: the counts (and weights) of 64 rows in ONE load, lane u holding row u's, handed out with v_readlane;
                    // the values of P from a scalar row base + the lane's topic offset, 32 rows per batch in two register sets -- the
                    // next batch is requested BEFORE the current one is added, so the chunk costs one memory latency, not one per
                    // batch (5.7 -> ~3 us per slow chunk; left to itself the compiler forms each product as its load arrives and keeps
                    // ~12 loads in flight; with a scalar load and 64-bit row * kp arithmetic per addend it was 45 ns per addend)
                    float pa[32], pb[32];
                    auto load32 = [&](float (&pv)[32], int r0) {
#pragma unroll
                        for (int u = 0; u < 32; ++u) pv[u] = Pc[(r0 + u) * kp];
                    };
                    auto add32 = [&](const float (&pv)[32], float xl, float wl, int l0) {
#pragma unroll
                        for (int u = 0; u < 32; ++u) {
                            const float x = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(xl), l0 + u));
                            float a2 = KIND != PAIR_NEG_TERMS ? x * pv[u] : -pv[u];      // plsa.py:188 (: 322)
                            if (KIND == PAIR_WEIGHTED) a2 = a2 * __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(wl), l0 + u));   // plsa.py:294
                            sum = sum + a2;                                              // plsa.py:193
                        }
                    };
                    float xl = KIND != PAIR_NEG_TERMS ? xc[tid] : 1.0f;
                    float wl = KIND == PAIR_WEIGHTED ? sw[dc[tid]] : 1.0f;
                    load32(pa, 0);
#pragma unroll 1
                    for (int j0 = 0; j0 < PAIR_L; j0 += 64) {
                        const int jn = min(j0 + 64, PAIR_L - 64);              // (the last round requests its own rows again: no branch)
                        const float xn = KIND != PAIR_NEG_TERMS ? xc[jn + tid] : 1.0f;
                        const float wn = KIND == PAIR_WEIGHTED ? sw[dc[jn + tid]] : 1.0f;
                        load32(pb, j0 + 32);
                        asm volatile("" ::: "memory");
                        add32(pa, xl, wl, 0);
                        load32(pa, jn);
                        asm volatile("" ::: "memory");
                        add32(pb, xl, wl, 32);
                        xl = xn; wl = wn;
                    }
                } else {
                    // the last chunk of the corpus
                    for (int j = 0; j < rows; ++j) {
                        float a = KIND != PAIR_NEG_TERMS ? xc[j] * Pc[j * kp] : -Pc[j * kp];
                        if (KIND == PAIR_WEIGHTED) a = a * sw[dc[j]];
                        sum = sum + a;
                    }
                }
                const unsigned b = __float_as_uint(sum), e = b >> 23;
                if (need) { es = e; m = e ? (b & 0x7FFFFFu) | 0x800000u : b; }
            };
            auto slow_step = [&](int ch) {
                const i64 c = t * TC + ch;
                if (!TWO) { slow_chunk(c, need); return; }
                // the group's chunks, their records fetched in one go and parked in LDS (every lane reads back its own)
                const i64 first = c * PAIR_R;
                const int cnt = (int)min((i64)PAIR_R, n_chunks1 - first);
#pragma unroll
                for (int i = 0; i < PAIR_R; ++i) {
                    const i64 at = min(first + i, n_chunks1 - 1) * kp + z;
                    mid_rec[i * 64 + tid] = pairs1[at];
                    mid_exp[i * 64 + tid] = exps1[at];
                }
                const bool need2 = need;
#pragma unroll 1
                for (int i = 0; i < cnt; ++i) {
                    const bool need1 = apply(mid_rec[i * 64 + tid], mid_exp[i * 64 + tid], need2);
                    if (__any(need1)) slow_chunk(first + i, need1);
                }
            };
            // Runs of fast chunks in a loop whose body has NO join (the slow way sits outside it), the LDS reads two chunks ahead of
            // their use in three register sets; the empty asm statements keep the read-ahead ahead (the scheduler sinks a read to its
            // use otherwise).
            // (three copies of the step, one per register set: rotating the sets with moves would wait for the read just issued)
            int ch = 0;
            auto read = [&](u32x4 &r, unsigned &x, int c) { const int cc = min(c, TC - 1) * 64; r = rp[cc]; x = xp[cc]; };
            while (ch < TC) {
                bool hit = false;
                read(ra, xa, ch);
                read(rb, xb, ch + 1);
                for (;;) {
                    read(rc, xc, ch + 2);
                    asm volatile("" ::: "memory");
                    if (fast_step(ra, xa)) { hit = true; break; }
                    if (++ch >= TC) break;
                    read(ra, xa, ch + 2);
                    asm volatile("" ::: "memory");
                    if (fast_step(rb, xb)) { hit = true; break; }
                    if (++ch >= TC) break;
                    read(rb, xb, ch + 2);
                    asm volatile("" ::: "memory");
                    if (fast_step(rc, xc)) { hit = true; break; }
                    if (++ch >= TC) break;
                }
                if (!hit) break;
                slow_step(ch);
                ++ch;
            }
        }
        if (mine) norm_pwz[z] = __uint_as_float(es ? (es << 23) | (m & 0x7FFFFFu) : m);
        if (tid == 0) { atomicAdd(stats, slow); atomicAdd(stats + 1, (unsigned long long)n_chunks1); }     // (in chunks, whatever is walked)
    } else if (!producer) {
        for (i64 t = 0; t < n_padded; ++t) __syncthreads();
    } else {
        // a producer lane's record slots inside a tile (the same for every tile): validity, chunk, offset from the tile's first record
        // (records of one chunk: kp apart), LDS position
        bool sval[WALK_SLOTS];
        int soff[WALK_SLOTS], spos[WALK_SLOTS], sch[WALK_SLOTS];
#pragma unroll
        for (int s = 0; s < WALK_SLOTS; ++s) {
            const int f = ptid + WALK_PRODUCERS * s;
            sval[s] = f < TC * width;
            sch[s] = sval[s] ? f / width : 0;
            const int col = sval[s] ? f - sch[s] * width : 0;
            soff[s] = sch[s] * kp + z0 + col;
            spos[s] = sch[s] * 64 + col;
        }
        u32x4 rr[WALK_DEPTH][WALK_SLOTS];
        unsigned rx[WALK_DEPTH][WALK_SLOTS];
        auto chunks_of = [&](i64 t) { return (int)max((i64)0, min((i64)TC, n_chunks - t * TC)); };
        auto fetch = [&](i64 t, u32x4 (&dr)[WALK_SLOTS], unsigned (&dx)[WALK_SLOTS]) {
            const i64 rec0 = min(t, n_tiles - 1) * TC * kp;      // (uniform; a tile past the end re-reads the last one)
            const int chunks = t < n_tiles ? chunks_of(t) : 0;
#pragma unroll
            for (int s = 0; s < WALK_SLOTS; ++s) {
                const bool ok = sval[s] && sch[s] < chunks;
                dr[s] = pairs[rec0 + (ok ? soff[s] : z0)];
                dx[s] = exps[rec0 + (ok ? soff[s] : z0)];
            }
        };
#pragma unroll
        for (int dd = 0; dd < WALK_DEPTH; ++dd) fetch(dd, rr[dd], rx[dd]);
        for (i64 t0 = 0; t0 < n_padded; t0 += WALK_DEPTH) {
#pragma unroll
            for (int dd = 0; dd < WALK_DEPTH; ++dd) {
                const i64 t = t0 + dd;
                const int chunks = chunks_of(t);
#pragma unroll
                for (int s = 0; s < WALK_SLOTS; ++s)
                    if (sval[s]) {
                        trec[dd & 1][spos[s]] = rr[dd][s];
                        texp[dd & 1][spos[s]] = sch[s] < chunks ? rx[dd][s] : (PAIR_NOOP | PAIR_NO_BINADE | PAIR_NO_BINADE << 16);   // past the end: no-ops
                    }
                __syncthreads();
                fetch(t + WALK_DEPTH, rr[dd], rx[dd]);
            }
        }
    }
}

// the walk's sum of the negated terms -> the likelihood (a serial chain starting from + 0.0 never ends on - 0.0)
__global__ void k_ref_ll_from_walk(const float *__restrict__ neg_sum, double *__restrict__ out) {
    const float s = neg_sum[0];
    out[0] = s == 0.0f ? 0.0 : (double)(-s);
}

// ------------------------------------------------------------------------------------------------
// PLSA_REFERENCE_LL: the log-likelihood as the reference's SOURCE states it, plsa.py:372-384 -- p_w_given_d one float32
// sum over the topics in order, result one float32 running sum over the non-zeros in order (plsa.py:322).  (What a numba
// user sees is something else: the compiled prange reduction is vectorised and lands within 1e-7 of the float64 sum at
// config 1, where this chain is 3.4e-3 away -- DESIGN.md section 2.)  The logarithm: float64 log of the float32 dot,
// rounded to float32 -- within an ulp of any libm's logf (NumPy's, glibc's and numba's differ from each other in the
// last place as well).
// The running sum itself: k_ref_ll_chain below (serial, 3.7 ns per term), or -- from 4096 non-zeros -- the pair kernels above
// over the negated terms (PAIR_NEG_TERMS; 11 / 37 / 56 / 377 ms -> 3.5 / 7.2 / 10.8 / 70 ms per evaluation at config 1 / config 2 /
// the config-3 150 k sample / config 3 whole; test_sequential_likelihood_from_parity_pairs: the same float32 either way).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_ref_ll_terms(const int *__restrict__ rowidx, const int *__restrict__ colidx,
                                                      const float *__restrict__ vals, i64 nnz,
                                                      const float *__restrict__ U, const float *__restrict__ Vt,
                                                      const float *__restrict__ sw, int kp, float *__restrict__ terms) {
    for (i64 nz = (i64)blockIdx.x * 256 + threadIdx.x; nz < nnz; nz += (i64)gridDim.x * 256) {
        const int d = rowidx[nz];
        const float4 *u = reinterpret_cast<const float4 *>(U + (i64)d * kp);
        const float4 *v = reinterpret_cast<const float4 *>(Vt + (i64)colidx[nz] * kp);
        float dot = 0.0f;
        for (int c = 0; c < (kp >> 2); ++c) {       // plsa.py:380-381
            const float4 a = v[c], b = u[c];
            dot += a.x * b.x; dot += a.y * b.y; dot += a.z * b.z; dot += a.w * b.w;
        }
        const float lg = (float)log((double)dot);
        float term = vals[nz] * lg;                  // plsa.py:383: x * log(p) * sample_weight[d], left to right
        term = term * (sw ? sw[d] : 1.0f);
        terms[nz] = term;
    }
}

// one wave: terms are loaded 64 at a time (PF loads in flight), every lane adds them in order (the same chain in all lanes).
// A tile of 64 terms goes through LDS (one write per lane, 16 broadcast ds_read_b128); the reads of tile p + 1 are in flight while
// tile p is added.  Terms past the end are +0.0 (x + 0.0 == x): no tail case.
__global__ __launch_bounds__(64) void k_ref_ll_chain(const float *__restrict__ terms, i64 nnz, double *__restrict__ out) {
    constexpr int PF = 16;
    __shared__ float4 sx[2][16];
    const int lane = threadIdx.x;
    float nxt[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) {
        const i64 i = (i64)p * 64 + lane;
        nxt[p] = i < nnz ? terms[i] : 0.0f;
    }
    float s = 0.0f;
    auto stage = [&](float mine, int buf, float4 (&v)[16]) {
        reinterpret_cast<float *>(sx[buf])[lane] = mine;
        wave_lds_fence();
#pragma unroll
        for (int c = 0; c < 16; ++c) v[c] = sx[buf][c];
    };
    auto add = [&](const float4 (&v)[16]) {
#pragma unroll
        for (int c = 0; c < 16; ++c) { s += v[c].x; s += v[c].y; s += v[c].z; s += v[c].w; }      // plsa.py:383
    };
    for (i64 base = 0; base < nnz; base += 64 * PF) {
        float cur[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) cur[p] = nxt[p];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const i64 i = base + 64 * PF + (i64)p * 64 + lane;
            nxt[p] = i < nnz ? terms[i] : 0.0f;
        }
        float4 va[16], vb[16];
        stage(cur[0], 0, va);
#pragma unroll
        for (int p = 0; p < PF; p += 2) {
            stage(cur[p + 1], 1, vb);
            add(va);
            wave_lds_fence();
            if (p + 2 < PF) stage(cur[p + 2], 0, va);
            add(vb);
            wave_lds_fence();
        }
    }
    if (lane == 0) out[0] = (double)s;
}

}  // namespace ref
}  // namespace plsa

#pragma clang fp contract(fast)   // hipcc's default for the rest of the translation unit
