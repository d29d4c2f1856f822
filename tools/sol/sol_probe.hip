// sol_probe.hip -- speed-of-light probes for the fused EM passes (round 3).
//
// Stand-alone harness: the corpus comes from the product library (plsa_generate_synthetic through the C ABI),
// everything else (CSC copy, column items, visiting orders, factors) is rebuilt here on the host so that
// kernel variants can be timed side by side with the shipped kernels (plsa_kernels.hpp is included as is).
//
//   build:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 tools/sol/sol_probe.hip \
//                 -Ienstop_amd/csrc -Iinclude -Lenstop_amd -lplsa_hip -Wl,-rpath,'$ORIGIN/../../enstop_amd' -o tools/sol/sol_probe
//   run:    tools/sol/sol_probe [config=3] [reps=10] [tests=all]      -> one JSON object per line
//
// What it answers: which ceiling bounds k_row_pass<fused> / k_col_pass<fused> -- VALU issue, the L2 request
// path, or the rate at which L2 misses are served (fabric) -- by timing the same traversals with the
// arithmetic stripped (gather-only) and with the thresholding removed, next to micro-benchmarks of the
// VALU issue rates and of random 256-byte row gathers against tables of L2 / Infinity-Cache / HBM size.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <string>
#include <vector>
#include "plsa_hip.h"
#include "plsa_kernels.hpp"

using plsa::i64;
using plsa::Shape;

#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)
#define PC(x) do { int r_ = (x); if (r_) { fprintf(stderr, "%s failed: %s\n", #x, plsa_last_error(ctx)); exit(3); } } while (0)

static plsa_ctx *ctx = nullptr;
static int g_reps = 10;
static hipStream_t g_stream;

template <class F>
static double time_ms(F &&launch, int reps = -1) {
    if (reps < 0) reps = g_reps;
    hipEvent_t a, b;
    HC(hipEventCreate(&a)); HC(hipEventCreate(&b));
    launch();   // warm-up
    HC(hipStreamSynchronize(g_stream));
    HC(hipEventRecord(a, g_stream));
    for (int i = 0; i < reps; ++i) launch();
    HC(hipEventRecord(b, g_stream));
    HC(hipStreamSynchronize(g_stream));
    HC(hipGetLastError());
    float ms = 0.f;
    HC(hipEventElapsedTime(&ms, a, b));
    HC(hipEventDestroy(a)); HC(hipEventDestroy(b));
    return ms / reps;
}

template <class T>
static T *dev(const std::vector<T> &h) {
    T *p = nullptr;
    HC(hipMalloc(&p, std::max<size_t>(h.size(), 4) * sizeof(T)));
    HC(hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
    return p;
}
template <class T>
static T *dev_alloc(size_t n) {
    T *p = nullptr;
    HC(hipMalloc(&p, std::max<size_t>(n, 4) * sizeof(T)));
    HC(hipMemset(p, 0, std::max<size_t>(n, 4) * sizeof(T)));
    return p;
}

// ---------------------------------------------------------------------------------------------------------
// 1. VALU issue rates: N independent chains per lane, ITER x 16 instructions, every SIMD filled (8 waves)
// ---------------------------------------------------------------------------------------------------------
enum { V_FMA, V_PKFMA, V_RCP, V_DPPADD, V_CNDMASK, V_MUL, V_BPERM, V_LOG };
template <int OP>
__global__ __launch_bounds__(256) void k_valu(float *out, int iters, float seed) {
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = seed + threadIdx.x * 1e-3f + i;
    const float m = 1.0000001f, c = 1e-9f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) p[i] = f2{a[2 * i], a[2 * i + 1]};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            if (OP == V_FMA) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else if (OP == V_MUL) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            } else if (OP == V_PKFMA) {
                const f2 mm = {m, m}, cc = {c, c};
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
#pragma unroll
                for (int i = 0; i < 4; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[i]) : "v"(mm), "v"(cc));
            } else if (OP == V_RCP) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(a[i]));
            } else if (OP == V_LOG) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_log_f32 %0, %0" : "+v"(a[i]));
            } else if (OP == V_DPPADD) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i]));
            } else if (OP == V_CNDMASK) {
#pragma unroll
                for (int i = 0; i < 8; ++i) asm volatile("v_cmp_gt_f32 vcc, %0, %1\n v_cndmask_b32 %0, 0, %0, vcc" : "+v"(a[i]) : "v"(c) : "vcc");
            } else if (OP == V_BPERM) {
                const int addr = ((threadIdx.x + 1) & 63) * 4;
#pragma unroll
                for (int i = 0; i < 8; ++i) a[i] = __int_as_float(__builtin_amdgcn_ds_bpermute(addr, __float_as_int(a[i])));
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += a[i];
#pragma unroll
    for (int i = 0; i < 4; ++i) s += p[i].x + p[i].y;
    if (s == 12345.678f) out[0] = s;
}

template <int OP>
static void valu_case(const char *name, int instr_per_iter, int cus) {
    float *out = dev_alloc<float>(4);
    const int iters = 4096;
    const int blocks = cus * 8;   // 8 x 256 threads per CU = 8 waves per SIMD
    double ms = time_ms([&] { hipLaunchKernelGGL((k_valu<OP>), dim3(blocks), dim3(256), 0, g_stream, out, iters, 1.0f); }, 5);
    const double winstr = (double)blocks * 4 * iters * instr_per_iter;       // wave-instructions
    const double per_simd_per_us = winstr / (cus * 4.0) / (ms * 1e3);
    printf("{\"test\": \"valu\", \"op\": \"%s\", \"ms\": %.4f, \"wave_instr_per_simd_per_us\": %.1f, \"cycles_per_wave_instr_at_2.4GHz\": %.3f}\n",
           name, ms, per_simd_per_us, 2400.0 / per_simd_per_us);
    fflush(stdout);
    HC(hipFree(out));
}

// ---------------------------------------------------------------------------------------------------------
// 2. random 256-byte row gathers (16 lanes x 16 B per row) against a table of a given size; UNR rows in
//    flight per group.  idx: precomputed pseudo-random row ids.  "rate" = bytes gathered / time.
// ---------------------------------------------------------------------------------------------------------
template <int UNR>
__global__ __launch_bounds__(256) void k_gather_rows(const float *__restrict__ table, const int *__restrict__ idx,
                                                     i64 n_idx, float *__restrict__ out) {
    const int li = threadIdx.x & 15;
    const i64 g = ((i64)blockIdx.x * 256 + threadIdx.x) >> 4;
    const i64 ng = ((i64)gridDim.x * 256) >> 4;
    float4 acc = plsa::zero4();
    for (i64 j = g * 16; j + 16 <= n_idx; j += ng * 16) {
        const int my = idx[j + li];
#pragma unroll
        for (int s0 = 0; s0 < 16; s0 += UNR) {
            float4 v[UNR];
#pragma unroll
            for (int q = 0; q < UNR; ++q) v[q] = plsa::ld4(table + (i64)__shfl(my, s0 + q, 16) * 64 + li * 4);
#pragma unroll
            for (int q = 0; q < UNR; ++q) { acc.x += v[q].x; acc.y += v[q].y; acc.z += v[q].z; acc.w += v[q].w; }
        }
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

__global__ void k_stream_read(const float4 *__restrict__ p, i64 n4, float *out) {
    float4 acc = plsa::zero4();
    const i64 stride = (i64)gridDim.x * 256 * 4;
    i64 i = (i64)blockIdx.x * 256 * 4 + threadIdx.x;
    for (; i + 3 * 256 < n4; i += stride) {
        float4 a = p[i], b = p[i + 256], c = p[i + 512], d = p[i + 768];
        acc.x += a.x + b.x + c.x + d.x; acc.y += a.y + b.y + c.y + d.y;
    }
    if (acc.x + acc.y == 12345.678f) out[0] = acc.x;
}

static void gather_cases(int cus) {
    const i64 n_idx = (i64)64 << 20;   // 64 M row gathers = 16 GB of rows
    std::vector<int> h(n_idx);
    float *out = dev_alloc<float>(4);
    const size_t sizes_mb[] = {2, 25, 64, 256, 2048};
    for (size_t mb : sizes_mb) {
        const i64 rows = (i64)(mb << 20) / 256;
        uint64_t s = 88172645463325252ull;
        for (i64 i = 0; i < n_idx; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (int)(s % (uint64_t)rows); }
        int *idx = dev(h);
        float *table = dev_alloc<float>((size_t)rows * 64);
        for (int unr : {4, 8, 16}) {
            const int grid = cus * 128;
            double ms = 0;
            if (unr == 4) ms = time_ms([&] { hipLaunchKernelGGL((k_gather_rows<4>), dim3(grid), dim3(256), 0, g_stream, table, idx, n_idx, out); }, 3);
            if (unr == 8) ms = time_ms([&] { hipLaunchKernelGGL((k_gather_rows<8>), dim3(grid), dim3(256), 0, g_stream, table, idx, n_idx, out); }, 3);
            if (unr == 16) ms = time_ms([&] { hipLaunchKernelGGL((k_gather_rows<16>), dim3(grid), dim3(256), 0, g_stream, table, idx, n_idx, out); }, 3);
            printf("{\"test\": \"random_row_gather\", \"table_mb\": %zu, \"rows_in_flight\": %d, \"ms\": %.3f, \"row_bytes_tb_s\": %.3f, \"rows_per_ns\": %.3f}\n",
                   mb, unr, ms, n_idx * 256.0 / ms / 1e9, n_idx / ms / 1e6);
            fflush(stdout);
        }
        double ms = time_ms([&] { hipLaunchKernelGGL(k_stream_read, dim3(cus * 16), dim3(256), 0, g_stream, (const float4 *)table, rows * 16, out); }, 20);
        printf("{\"test\": \"stream_read\", \"table_mb\": %zu, \"ms\": %.4f, \"tb_s\": %.3f}\n", mb, ms, (double)(mb << 20) / ms / 1e9);
        fflush(stdout);
        HC(hipFree(idx)); HC(hipFree(table));
    }
    HC(hipFree(out));
}

// ---------------------------------------------------------------------------------------------------------
// 3. the fused passes with parts removed.  MODE 0: as shipped (responsibilities with thresholding);
//    1: no thresholding (keep = product); 2: gather-only (acc += x * row: the memory floor of the traversal)
// ---------------------------------------------------------------------------------------------------------
template <int MODE>
__device__ __forceinline__ void nz_update(const float4 &own, const float4 &row, float x, float thresh, float4 &acc) {
    if (MODE == 2) {
        acc.x += x * row.x; acc.y += x * row.y; acc.z += x * row.z; acc.w += x * row.w;
        return;
    }
    float4 v;
    v.x = row.x * own.x; v.y = row.y * own.y; v.z = row.z * own.z; v.w = row.w * own.w;
    if (MODE == 0) {
        v.x = v.x > thresh ? v.x : 0.f; v.y = v.y > thresh ? v.y : 0.f;
        v.z = v.z > thresh ? v.z : 0.f; v.w = v.w > thresh ? v.w : 0.f;
    }
    const float norm = plsa::group_sum<16>(plsa::hsum(v));
    const float s = x * plsa::inv_norm(norm);
    acc.x += s * v.x; acc.y += s * v.y; acc.z += s * v.z; acc.w += s * v.w;
}

template <int MODE, int UNR>
__global__ __launch_bounds__(256) void k_row_variant(const int *__restrict__ indptr, const int *__restrict__ colidx,
                                                     const float *__restrict__ vals, int n,
                                                     const int *__restrict__ row_order, const float *__restrict__ U,
                                                     const float *__restrict__ Vt, float *__restrict__ U_new, float thresh) {
    constexpr int LPN = 16, GPB = 16;
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    for (i64 r = (i64)blockIdx.x * GPB + gid; r < n; r += (i64)gridDim.x * GPB) {
        const int d = row_order[r];
        const int j0 = indptr[d], j1 = indptr[d + 1];
        const float4 u = plsa::ld4(U + (i64)d * 64 + li * 4);
        float4 acc = plsa::zero4();
        int w_n = (j0 + li < j1) ? colidx[j0 + li] : 0;
        float x_n = (j0 + li < j1) ? vals[j0 + li] : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int w_l = w_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            w_n = jn < j1 ? colidx[jn] : 0;
            x_n = jn < j1 ? vals[jn] : 0.f;
            const int cnt = min(LPN, j1 - jb);
            for (int s0 = 0; s0 < cnt; s0 += UNR) {
                float4 a[UNR];
                float x[UNR];
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    const int w = __shfl(w_l, s0 + q, LPN);
                    x[q] = __shfl(x_l, s0 + q, LPN);
                    a[q] = plsa::ld4(Vt + (i64)w * 64 + li * 4);
                }
#pragma unroll
                for (int q = 0; q < UNR; ++q) nz_update<MODE>(u, a[q], x[q], thresh, acc);
            }
        }
        const float rown = plsa::group_sum<LPN>(plsa::hsum(acc));
        float4 o = acc;
        if (rown > 0.f) { o.x /= rown; o.y /= rown; o.z /= rown; o.w /= rown; }
        plsa::st4(U_new + (i64)d * 64 + li * 4, o);
    }
}

template <int MODE, int UNR>
__global__ __launch_bounds__(256) void k_col_variant(const int *__restrict__ item_order, const int *__restrict__ item_col,
                                                     const int *__restrict__ item_start, const int *__restrict__ item_end,
                                                     i64 n_items, const int *__restrict__ csc_row,
                                                     const float *__restrict__ csc_val, const float *__restrict__ U,
                                                     const float *__restrict__ Vt, float *__restrict__ partial, float thresh) {
    constexpr int LPN = 16, GPB = 16;
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    const int xcd = (int)(blockIdx.x & 7);
    const i64 nq = (gridDim.x + 7 - xcd) / 8;
    const i64 q = blockIdx.x >> 3;
    const i64 per = (n_items + 7) / 8;
    const i64 lo = xcd * per, hi = min(n_items, lo + per);
    for (i64 io = lo + q * GPB + gid; io < hi; io += nq * GPB) {
        const int it = item_order[io];
        const int w = item_col[it];
        const int j0 = item_start[it], j1 = item_end[it];
        const float4 vt = plsa::ld4(Vt + (i64)w * 64 + li * 4);
        float4 acc = plsa::zero4();
        int d_n = (j0 + li < j1) ? csc_row[j0 + li] : 0;
        float x_n = (j0 + li < j1) ? csc_val[j0 + li] : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int d_l = d_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            d_n = jn < j1 ? csc_row[jn] : 0;
            x_n = jn < j1 ? csc_val[jn] : 0.f;
            const int cnt = min(LPN, j1 - jb);
            int s0 = 0;
            for (; s0 + UNR <= cnt; s0 += UNR) {
                float4 a[UNR];
                float x[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    x[u] = __shfl(x_l, s0 + u, LPN);
                    a[u] = plsa::ld4(U + (i64)__shfl(d_l, s0 + u, LPN) * 64 + li * 4);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) nz_update<MODE>(vt, a[u], x[u], thresh, acc);
            }
            for (; s0 < cnt; s0 += 2) {
                float4 a[2];
                float x[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    x[u] = __shfl(x_l, s0 + u, LPN);
                    a[u] = plsa::ld4(U + (i64)__shfl(d_l, s0 + u, LPN) * 64 + li * 4);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) nz_update<MODE>(vt, a[u], x[u], thresh, acc);
            }
        }
        plsa::st4(partial + (i64)it * 64 + li * 4, acc);
    }
}

// per-workgroup timeline of a column-pass launch: end time (100 MHz wall clock) and XCC id of every block
__device__ __forceinline__ unsigned xcc_id() {
    unsigned v;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v));
    return v & 0xf;
}
template <int MODE, int UNR, int ASSIGN>
__global__ __launch_bounds__(256) void k_col_timeline(const int *__restrict__ item_order, const int *__restrict__ item_col,
                                                      const int *__restrict__ item_start, const int *__restrict__ item_end,
                                                      i64 n_items, const int *__restrict__ csc_row,
                                                      const float *__restrict__ csc_val, const float *__restrict__ U,
                                                      const float *__restrict__ Vt, float *__restrict__ partial, float thresh,
                                                      unsigned long long *__restrict__ t_end, unsigned *__restrict__ xcc,
                                                      unsigned long long *__restrict__ t_start) {
    constexpr int LPN = 16, GPB = 16;
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    if (threadIdx.x == 0) t_start[blockIdx.x] = wall_clock64();
    // ASSIGN 0: contiguous eighth of the visiting list per XCD (shipped); 1: plain grid-stride over the whole list
    const int xcd = ASSIGN == 0 ? (int)(blockIdx.x & 7) : 0;
    const i64 nq = ASSIGN == 0 ? (gridDim.x + 7 - xcd) / 8 : gridDim.x;
    const i64 q = ASSIGN == 0 ? (blockIdx.x >> 3) : blockIdx.x;
    const i64 per = ASSIGN == 0 ? (n_items + 7) / 8 : n_items;
    const i64 lo = xcd * per, hi = min(n_items, lo + per);
    for (i64 io = lo + q * GPB + gid; io < hi; io += nq * GPB) {
        const int it = item_order[io];
        const int w = item_col[it];
        const int j0 = item_start[it], j1 = item_end[it];
        const float4 vt = plsa::ld4(Vt + (i64)w * 64 + li * 4);
        float4 acc = plsa::zero4();
        int d_n = (j0 + li < j1) ? csc_row[j0 + li] : 0;
        float x_n = (j0 + li < j1) ? csc_val[j0 + li] : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int d_l = d_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            d_n = jn < j1 ? csc_row[jn] : 0;
            x_n = jn < j1 ? csc_val[jn] : 0.f;
            const int cnt = min(LPN, j1 - jb);
            for (int s0 = 0; s0 < cnt; s0 += UNR) {
                float4 a[UNR];
                float x[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    x[u] = __shfl(x_l, s0 + u, LPN);
                    a[u] = plsa::ld4(U + (i64)__shfl(d_l, s0 + u, LPN) * 64 + li * 4);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) nz_update<MODE>(vt, a[u], x[u], thresh, acc);
            }
        }
        plsa::st4(partial + (i64)it * 64 + li * 4, acc);
    }
    __syncthreads();
    if (threadIdx.x == 0) { t_end[blockIdx.x] = wall_clock64(); xcc[blockIdx.x] = xcc_id(); }
}

// ---------------------------------------------------------------------------------------------------------
// 4. column pass with dynamic, locality-preserving scheduling: the visiting list (ascending first document) is
//    cut into one contiguous range per XCD; every 16-lane group claims the next item of ITS XCD's range from an
//    atomic queue word (so an XCD's resident groups always work on a contiguous window of the list), and when
//    the range is exhausted it steals from the tail of the range with the most work left.
//    queue word = (head << 32) | (tail + QOFF)
// ---------------------------------------------------------------------------------------------------------
constexpr unsigned QOFF = 1u << 30;
template <int MODE, int UNR>
__global__ __launch_bounds__(256) void k_col_dyn(const int4 *__restrict__ items, unsigned long long *__restrict__ queue,
                                                 const int *__restrict__ csc_row, const float *__restrict__ csc_val,
                                                 const float *__restrict__ U, const float *__restrict__ Vt,
                                                 float *__restrict__ partial, float thresh, int steal, int by_block_id) {
    constexpr int LPN = 16;
    const int li = threadIdx.x % LPN;
    const unsigned me = by_block_id ? (blockIdx.x & 7) : (xcc_id() & 7);
    bool own_empty = false;
    for (;;) {
        int io = -1;
        if (li == 0) {
            if (!own_empty) {
                const unsigned long long old = atomicAdd(&queue[me], 1ull << 32);
                const int head = (int)(old >> 32), tail = (int)((unsigned)old - QOFF);
                if (head < tail) io = head; else own_empty = true;
            }
            if (io < 0 && steal) {
                for (int tries = 0; tries < 16 && io < 0; ++tries) {
                    int best = -1, bestrem = 0;
                    for (int x = 0; x < 8; ++x) {
                        const unsigned long long w = __hip_atomic_load(&queue[x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const int rem = (int)((unsigned)w - QOFF) - (int)(w >> 32);
                        if (rem > bestrem) { bestrem = rem; best = x; }
                    }
                    if (best < 0) break;
                    const unsigned long long o2 = atomicAdd(&queue[best], ~0ull);    // tail - 1
                    const int h2 = (int)(o2 >> 32), t2 = (int)((unsigned)o2 - QOFF);
                    if (h2 < t2) io = t2 - 1;
                }
            }
        }
        io = __shfl(io, 0, LPN);
        if (io < 0) break;
        const int4 rec = items[io];
        const int w = rec.x, j0 = rec.y, j1 = rec.z, it = rec.w;
        const float4 vt = plsa::ld4(Vt + (i64)w * 64 + li * 4);
        float4 acc = plsa::zero4();
        int d_n = (j0 + li < j1) ? csc_row[j0 + li] : 0;
        float x_n = (j0 + li < j1) ? csc_val[j0 + li] : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int d_l = d_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            d_n = jn < j1 ? csc_row[jn] : 0;
            x_n = jn < j1 ? csc_val[jn] : 0.f;
            const int cnt = min(LPN, j1 - jb);
            int s0 = 0;
            for (; s0 + UNR <= cnt; s0 += UNR) {
                float4 a[UNR];
                float x[UNR];
#pragma unroll
                for (int u = 0; u < UNR; ++u) {
                    x[u] = __shfl(x_l, s0 + u, LPN);
                    a[u] = plsa::ld4(U + (i64)__shfl(d_l, s0 + u, LPN) * 64 + li * 4);
                }
#pragma unroll
                for (int u = 0; u < UNR; ++u) nz_update<MODE>(vt, a[u], x[u], thresh, acc);
            }
            for (; s0 < cnt; s0 += 2) {
                float4 a[2];
                float x[2];
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    x[u] = __shfl(x_l, s0 + u, LPN);
                    a[u] = plsa::ld4(U + (i64)__shfl(d_l, s0 + u, LPN) * 64 + li * 4);
                }
#pragma unroll
                for (int u = 0; u < 2; ++u) nz_update<MODE>(vt, a[u], x[u], thresh, acc);
            }
        }
        plsa::st4(partial + (i64)it * 64 + li * 4, acc);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 3b. document pass with the R most frequent words' P(w|z) rows staged in LDS (round 4).  `colidx` is TAGGED: a hot word
//     is stored as ~slot (negative), every other word as its id.  Persistent grid (BPC workgroups per CU, grid-stride over
//     the documents): the hot rows are staged once per workgroup.  T threads per workgroup share one LDS image.
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int UNR, int T>
__global__ __launch_bounds__(T) void k_row_hot(const int *__restrict__ indptr, const int *__restrict__ colidx_tag,
                                               const float *__restrict__ vals, int n, const int *__restrict__ row_order,
                                               const float *__restrict__ U, const float *__restrict__ Vt,
                                               const int *__restrict__ hot_words, int n_hot, float *__restrict__ U_new,
                                               float thresh) {
    constexpr int LPN = 16, GPB = T / 16;
    extern __shared__ float4 hot[];          // [n_hot][16] float4 = n_hot rows of 256 B
    for (int i = threadIdx.x; i < n_hot * 16; i += T)
        hot[i] = plsa::ld4(Vt + (i64)hot_words[i >> 4] * 64 + (i & 15) * 4);
    __syncthreads();
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    for (i64 r = (i64)blockIdx.x * GPB + gid; r < n; r += (i64)gridDim.x * GPB) {
        const int d = row_order[r];
        const int j0 = indptr[d], j1 = indptr[d + 1];
        const float4 u = plsa::ld4(U + (i64)d * 64 + li * 4);
        float4 acc = plsa::zero4();
        int w_n = (j0 + li < j1) ? colidx_tag[j0 + li] : 0;
        float x_n = (j0 + li < j1) ? vals[j0 + li] : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int w_l = w_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            w_n = jn < j1 ? colidx_tag[jn] : 0;
            x_n = jn < j1 ? vals[jn] : 0.f;
            const int cnt = min(LPN, j1 - jb);
            for (int s0 = 0; s0 < cnt; s0 += UNR) {
                float4 a[UNR];
                float x[UNR];
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    const int w = __shfl(w_l, s0 + q, LPN);
                    x[q] = __shfl(x_l, s0 + q, LPN);
                    if (w < 0) a[q] = hot[(~w) * 16 + li];
                    else a[q] = plsa::ld4(Vt + (i64)w * 64 + li * 4);
                }
#pragma unroll
                for (int q = 0; q < UNR; ++q) nz_update<MODE>(u, a[q], x[q], thresh, acc);
            }
        }
        const float rown = plsa::group_sum<LPN>(plsa::hsum(acc));
        float4 o = acc;
        if (rown > 0.f) { o.x /= rown; o.y /= rown; o.z /= rown; o.w /= rown; }
        plsa::st4(U_new + (i64)d * 64 + li * 4, o);
    }
}

// ---------------------------------------------------------------------------------------------------------
// 5. column pass, static schedule with MEASURED XCD boundaries.  The visiting list is cut into chunks of 16 items
//    (one per group of a workgroup); XCD x (= blockIdx & 7) walks the chunks [lo[x], lo[x+1]).  Per chunk the
//    workgroup also writes the float64 sum of its 16 accumulators (what norm_pwz is made of), so the results do
//    not depend on the boundaries.  TIMED: thread 0 records the block's end time (100 MHz wall clock).
// ---------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float4 ld4_sel(const float *p, bool nt) { return nt ? plsa::ld4_nt(p) : plsa::ld4(p); }
// TAG only makes the kernel NAME distinct (rocprofv3 --pmc rows of the hit / miss experiment); W = min waves per SIMD
template <int MODE, int UNR, bool TIMED, int TAG = 0, int W = 1>
__global__ __launch_bounds__(256, W) void k_col_chunks(const int4 *__restrict__ items, i64 n_items, const int *__restrict__ lo,
                                                    const int *__restrict__ csc_row, const float *__restrict__ csc_val,
                                                    const float *__restrict__ U, const float *__restrict__ Vt,
                                                    float *__restrict__ partial, double *__restrict__ chunk_sums, float thresh,
                                                    unsigned long long *__restrict__ t_end) {
    constexpr int LPN = 16, GPB = 16;
    __shared__ double sred[GPB * 64];
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    const int x = blockIdx.x & 7;
    const int nq = (gridDim.x + 7 - x) / 8;
    for (int ch = lo[x] + (int)(blockIdx.x >> 3); ch < lo[x + 1]; ch += nq) {
        const i64 io = (i64)ch * GPB + gid;
        float4 acc = plsa::zero4();
        if (io < n_items) {
            const int4 rec = items[io];
            const int w = rec.x, j0 = rec.y, j1 = rec.z;
            const bool nt = rec.w != 0;
            const float4 vt = plsa::ld4(Vt + (i64)w * 64 + li * 4);
            int d_n = (j0 + li < j1) ? csc_row[j0 + li] : 0;
            float x_n = (j0 + li < j1) ? csc_val[j0 + li] : 0.f;
            for (int jb = j0; jb < j1; jb += LPN) {
                const int d_l = d_n;
                const float x_l = x_n;
                const int jn = jb + LPN + li;
                d_n = jn < j1 ? csc_row[jn] : 0;
                x_n = jn < j1 ? csc_val[jn] : 0.f;
                const int cnt = min(LPN, j1 - jb);
                int s0 = 0;
                for (; s0 + UNR <= cnt; s0 += UNR) {
                    float4 a[UNR];
                    float xx[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) {
                        xx[u] = __shfl(x_l, s0 + u, LPN);
                        a[u] = ld4_sel(U + (i64)__shfl(d_l, s0 + u, LPN) * 64 + li * 4, nt);
                    }
#pragma unroll
                    for (int u = 0; u < UNR; ++u) nz_update<MODE>(vt, a[u], xx[u], thresh, acc);
                }
                for (; s0 < cnt; s0 += 2) {
                    float4 a[2];
                    float xx[2];
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        xx[u] = __shfl(x_l, s0 + u, LPN);
                        a[u] = ld4_sel(U + (i64)__shfl(d_l, s0 + u, LPN) * 64 + li * 4, nt);
                    }
#pragma unroll
                    for (int u = 0; u < 2; ++u) nz_update<MODE>(vt, a[u], xx[u], thresh, acc);
                }
            }
            plsa::st4(partial + io * 64 + li * 4, acc);
        }
        double *p = sred + gid * 64 + li * 4;
        p[0] = (double)acc.x; p[1] = (double)acc.y; p[2] = (double)acc.z; p[3] = (double)acc.w;
        __syncthreads();
        if (threadIdx.x < 64) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < GPB; ++g) t += sred[g * 64 + threadIdx.x];
            chunk_sums[(i64)ch * 64 + threadIdx.x] = t;
        }
        __syncthreads();
    }
    if (TIMED && threadIdx.x == 0) t_end[blockIdx.x] = wall_clock64();
}

// ---------------------------------------------------------------------------------------------------------
// 5b. column pass, software-pipelined: the P(z|d) rows of the NEXT 8 entries are requested before the current 8 are
//     reduced (round 4: both passes sit at ~80 % of the VALU issue ceiling AND ~80 % of their miss-service bound --
//     does decoupling the gathers from the arithmetic of the same wave overlap them better?).  Same chunk schedule as
//     k_col_chunks; padded entries read document 0 with count 0.
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int W>
__global__ __launch_bounds__(256, W) void k_col_chunks_pipe(const int4 *__restrict__ items, i64 n_items, const int *__restrict__ lo,
                                                            const int *__restrict__ csc_row, const float *__restrict__ csc_val,
                                                            const float *__restrict__ U, const float *__restrict__ Vt,
                                                            float *__restrict__ partial, double *__restrict__ chunk_sums, float thresh) {
    constexpr int LPN = 16, GPB = 16, UNR = 8;
    __shared__ double sred[GPB * 64];
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    const int x = blockIdx.x & 7;
    const int nq = (gridDim.x + 7 - x) / 8;
    for (int ch = lo[x] + (int)(blockIdx.x >> 3); ch < lo[x + 1]; ch += nq) {
        const i64 io = (i64)ch * GPB + gid;
        float4 acc = plsa::zero4();
        if (io < n_items) {
            const int4 rec = items[io];
            const int w = rec.x, j0 = rec.y, j1 = rec.z;
            const float4 vt = plsa::ld4(Vt + (i64)w * 64 + li * 4);
            // index batches of 16 entries, two sub-batches of 8 each; (d_c, x_c): the index batch being consumed
            int d_c = (j0 + li < j1) ? csc_row[j0 + li] : 0;
            float x_c = (j0 + li < j1) ? csc_val[j0 + li] : 0.f;
            float4 a_n[UNR];
            float xs_n[UNR];
#pragma unroll
            for (int u = 0; u < UNR; ++u) {                 // prologue: first sub-batch
                xs_n[u] = __shfl(x_c, u, LPN);
                a_n[u] = plsa::ld4(U + (i64)__shfl(d_c, u, LPN) * 64 + li * 4);
            }
            for (int jb = j0; jb < j1; jb += LPN) {
                const int jn = jb + LPN + li;
                const int d_nx = jn < j1 ? csc_row[jn] : 0;          // next index batch (requested a batch ahead)
                const float x_nx = jn < j1 ? csc_val[jn] : 0.f;
#pragma unroll
                for (int half = 0; half < 2; ++half) {
                    float4 a_c[UNR];
                    float xs_c[UNR];
#pragma unroll
                    for (int u = 0; u < UNR; ++u) { a_c[u] = a_n[u]; xs_c[u] = xs_n[u]; }
                    // request the next sub-batch: second half of this index batch, or first half of the next one
                    const int d_src = half == 0 ? d_c : d_nx;
                    const float x_src = half == 0 ? x_c : x_nx;
                    const int base = half == 0 ? UNR : 0;
                    const bool more = half == 0 ? (jb + UNR < j1) : (jb + LPN < j1);
                    if (more) {
#pragma unroll
                        for (int u = 0; u < UNR; ++u) {
                            xs_n[u] = __shfl(x_src, base + u, LPN);
                            a_n[u] = plsa::ld4(U + (i64)__shfl(d_src, base + u, LPN) * 64 + li * 4);
                        }
                    }
                    if (half == 0 || jb + UNR < j1) {
#pragma unroll
                        for (int u = 0; u < UNR; ++u) nz_update<MODE>(vt, a_c[u], xs_c[u], thresh, acc);
                    }
                }
                d_c = d_nx; x_c = x_nx;
            }
            plsa::st4(partial + io * 64 + li * 4, acc);
        }
        double *p = sred + gid * 64 + li * 4;
        p[0] = (double)acc.x; p[1] = (double)acc.y; p[2] = (double)acc.z; p[3] = (double)acc.w;
        __syncthreads();
        if (threadIdx.x < 64) {
            double t = 0.0;
#pragma unroll
            for (int g = 0; g < GPB; ++g) t += sred[g * 64 + threadIdx.x];
            chunk_sums[(i64)ch * 64 + threadIdx.x] = t;
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------------
// 6. document pass split in TIME by word class: phase 0 walks only a document's entries of frequent words (their
//    P(w|z) rows fit the L2 and are not evicted by the stream of rare rows), phase 1 the rest.  The un-normalised
//    accumulator travels through memory between the phases.  (Measured: slower, r03_row_pass_split_by_word_class_rejected.jsonl)
// ---------------------------------------------------------------------------------------------------------
template <int MODE, int UNR, int PHASE>
__global__ __launch_bounds__(256) void k_row_split(const int *__restrict__ jbeg, const int *__restrict__ jend,
                                                   const int *__restrict__ colidx, const float *__restrict__ vals, int n,
                                                   const int *__restrict__ row_order, const float *__restrict__ U,
                                                   const float *__restrict__ Vt, float *__restrict__ acc_buf,
                                                   float *__restrict__ U_new, float thresh) {
    constexpr int LPN = 16, GPB = 16;
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    for (i64 r = (i64)blockIdx.x * GPB + gid; r < n; r += (i64)gridDim.x * GPB) {
        const int d = row_order[r];
        const int j0 = jbeg[d], j1 = jend[d];
        const float4 u = plsa::ld4(U + (i64)d * 64 + li * 4);
        float4 acc = PHASE == 0 ? plsa::zero4() : plsa::ld4(acc_buf + (i64)d * 64 + li * 4);
        int w_n = (j0 + li < j1) ? colidx[j0 + li] : 0;
        float x_n = (j0 + li < j1) ? vals[j0 + li] : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int w_l = w_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            w_n = jn < j1 ? colidx[jn] : 0;
            x_n = jn < j1 ? vals[jn] : 0.f;
            const int cnt = min(LPN, j1 - jb);
            for (int s0 = 0; s0 < cnt; s0 += UNR) {
                float4 a[UNR];
                float x[UNR];
#pragma unroll
                for (int q = 0; q < UNR; ++q) {
                    const int w = __shfl(w_l, s0 + q, LPN);
                    x[q] = __shfl(x_l, s0 + q, LPN);
                    a[q] = plsa::ld4(Vt + (i64)w * 64 + li * 4);
                }
#pragma unroll
                for (int q = 0; q < UNR; ++q) nz_update<MODE>(u, a[q], x[q], thresh, acc);
            }
        }
        if (PHASE == 0) {
            plsa::st4(acc_buf + (i64)d * 64 + li * 4, acc);
        } else {
            const float rown = plsa::group_sum<LPN>(plsa::hsum(acc));
            float4 o = acc;
            if (rown > 0.f) { o.x /= rown; o.y /= rown; o.z /= rown; o.w /= rown; }
            plsa::st4(U_new + (i64)d * 64 + li * 4, o);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------
// 7. leaner instruction stream for the document pass: the (word, count) of entry q is broadcast with DPP
//    row_newbcast instead of ds_bpermute (no LDS-pipe traffic), the P(w|z) row is addressed by a 32-bit byte offset
//    against the uniform table base.  (Measured: -2.3 % in isolation, nothing inside the iteration:
//    r03_row_pass_dpp_broadcast_32bit_offsets_rejected.txt)
// ---------------------------------------------------------------------------------------------------------
template <int Q>
__device__ __forceinline__ int bcast16_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + Q, 0xF, 0xF, false); }
template <int Q>
__device__ __forceinline__ float bcast16_f(float v) { return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x150 + Q, 0xF, 0xF, false)); }

template <int MODE, int S0>
__device__ __forceinline__ void lean_batch4(int w_l, float x_l, int cnt, const char *__restrict__ Vt_bytes, unsigned lane_off,
                                            const float4 &u, float thresh, float4 &acc) {
    if (S0 < cnt) {       // group-uniform
        const unsigned o0 = ((unsigned)bcast16_i<S0 + 0>(w_l) << 8) + lane_off;
        const unsigned o1 = ((unsigned)bcast16_i<S0 + 1>(w_l) << 8) + lane_off;
        const unsigned o2 = ((unsigned)bcast16_i<S0 + 2>(w_l) << 8) + lane_off;
        const unsigned o3 = ((unsigned)bcast16_i<S0 + 3>(w_l) << 8) + lane_off;
        const float4 a0 = *reinterpret_cast<const float4 *>(Vt_bytes + o0);
        const float4 a1 = *reinterpret_cast<const float4 *>(Vt_bytes + o1);
        const float4 a2 = *reinterpret_cast<const float4 *>(Vt_bytes + o2);
        const float4 a3 = *reinterpret_cast<const float4 *>(Vt_bytes + o3);
        nz_update<MODE>(u, a0, bcast16_f<S0 + 0>(x_l), thresh, acc);
        nz_update<MODE>(u, a1, bcast16_f<S0 + 1>(x_l), thresh, acc);
        nz_update<MODE>(u, a2, bcast16_f<S0 + 2>(x_l), thresh, acc);
        nz_update<MODE>(u, a3, bcast16_f<S0 + 3>(x_l), thresh, acc);
    }
}

template <int MODE>
__global__ __launch_bounds__(256) void k_row_lean(const int *__restrict__ indptr, const int *__restrict__ colidx,
                                                  const float *__restrict__ vals, int n,
                                                  const int *__restrict__ row_order, const float *__restrict__ U,
                                                  const float *__restrict__ Vt, float *__restrict__ U_new, float thresh) {
    constexpr int LPN = 16, GPB = 16;
    const int li = threadIdx.x % LPN, gid = threadIdx.x / LPN;
    const char *Vt_bytes = reinterpret_cast<const char *>(Vt);
    const unsigned lane_off = li * 16;
    for (i64 r = (i64)blockIdx.x * GPB + gid; r < n; r += (i64)gridDim.x * GPB) {
        const int d = row_order[r];
        const int j0 = indptr[d], j1 = indptr[d + 1];
        const float4 u = plsa::ld4(U + (i64)d * 64 + li * 4);
        float4 acc = plsa::zero4();
        int w_n = (j0 + li < j1) ? colidx[j0 + li] : 0;
        float x_n = (j0 + li < j1) ? vals[j0 + li] : 0.f;
        for (int jb = j0; jb < j1; jb += LPN) {
            const int w_l = w_n;
            const float x_l = x_n;
            const int jn = jb + LPN + li;
            w_n = jn < j1 ? colidx[jn] : 0;
            x_n = jn < j1 ? vals[jn] : 0.f;
            const int cnt = min(LPN, j1 - jb);
            lean_batch4<MODE, 0>(w_l, x_l, cnt, Vt_bytes, lane_off, u, thresh, acc);
            lean_batch4<MODE, 4>(w_l, x_l, cnt, Vt_bytes, lane_off, u, thresh, acc);
            lean_batch4<MODE, 8>(w_l, x_l, cnt, Vt_bytes, lane_off, u, thresh, acc);
            lean_batch4<MODE, 12>(w_l, x_l, cnt, Vt_bytes, lane_off, u, thresh, acc);
        }
        const float rown = plsa::group_sum<LPN>(plsa::hsum(acc));
        float4 o = acc;
        if (rown > 0.f) { o.x /= rown; o.y /= rown; o.z /= rown; o.w /= rown; }
        plsa::st4(U_new + (i64)d * 64 + li * 4, o);
    }
}

// ---------------------------------------------------------------------------------------------------------
int main(int argc, char **argv) {
    const int config = argc > 1 ? atoi(argv[1]) : 3;
    g_reps = argc > 2 ? atoi(argv[2]) : 10;
    const std::string tests = argc > 3 ? argv[3] : "all";
    auto want = [&](const char *t) { return tests == "all" || tests.find(t) != std::string::npos; };
    PC(plsa_create(0, &ctx));
    HC(hipStreamCreate(&g_stream));
    hipDeviceProp_t prop;
    HC(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    printf("{\"test\": \"device\", \"name\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.name, cus, prop.clockRate / 1000);

    if (want("valu")) {
        valu_case<V_FMA>("v_fma_f32", 16, cus);
        valu_case<V_MUL>("v_mul_f32", 16, cus);
        valu_case<V_PKFMA>("v_pk_fma_f32", 16, cus);
        valu_case<V_RCP>("v_rcp_f32", 16, cus);
        valu_case<V_LOG>("v_log_f32", 16, cus);
        valu_case<V_DPPADD>("v_add_f32_dpp", 16, cus);
        valu_case<V_CNDMASK>("v_cmp+v_cndmask", 32, cus);
        valu_case<V_BPERM>("ds_bpermute_b32", 16, cus);
    }
    if (want("gather")) gather_cases(cus);
    if (!want("row") && !want("col") && !want("timeline") && !want("dyn") && !want("balance") && !want("rowx") && !want("ntcol") && !want("order") && !want("rowsplit") && !want("rowlean") && !want("hitmiss") && !want("mix") && !want("headsplit") && !want("rowhot") && !want("rowcold") && !want("rowhitmiss") && !want("pipe")) return 0;

    // ---- corpus ------------------------------------------------------------------------------------------
    i64 n = 1000000, m = 100000, nnz_t = 100000000;
    if (config == 2) { n = 100000; m = 50000; nnz_t = 10000000; }
    int64_t nnz64 = 0;
    PC(plsa_generate_synthetic(ctx, n, m, nnz_t, 1.07, 0, &nnz64));
    const i64 nnz = (i64)nnz64;
    std::vector<int> indptr(n + 1), col(nnz);
    std::vector<float> val(nnz);
    PC(plsa_download_active_csr(ctx, indptr.data(), col.data(), val.data()));
    plsa_destroy(ctx);
    ctx = nullptr;
    // rows by descending length (stable)
    std::vector<int> row_order(n);
    std::iota(row_order.begin(), row_order.end(), 0);
    std::stable_sort(row_order.begin(), row_order.end(), [&](int a, int b) { return indptr[a + 1] - indptr[a] > indptr[b + 1] - indptr[b]; });
    // CSC (entries of a column in document order), items of <= 256 entries, visiting order = first document
    std::vector<int> colptr(m + 1, 0), csc_row(nnz);
    std::vector<float> csc_val(nnz);
    for (i64 j = 0; j < nnz; ++j) colptr[col[j] + 1]++;
    for (i64 c = 0; c < m; ++c) colptr[c + 1] += colptr[c];
    {
        std::vector<int> fill(colptr.begin(), colptr.end() - 1);
        for (i64 d = 0; d < n; ++d)
            for (int j = indptr[d]; j < indptr[d + 1]; ++j) { const int p = fill[col[j]]++; csc_row[p] = (int)d; csc_val[p] = val[j]; }
    }
    const int seg = 256;
    std::vector<int> item_col, item_start, item_end;
    for (i64 c = 0; c < m; ++c)
        for (int st = colptr[c]; st < colptr[c + 1]; st += seg) { item_col.push_back((int)c); item_start.push_back(st); item_end.push_back(std::min(st + seg, colptr[c + 1])); }
    const i64 n_items = (i64)item_col.size();
    std::vector<int> item_order(n_items);
    std::iota(item_order.begin(), item_order.end(), 0);
    std::stable_sort(item_order.begin(), item_order.end(), [&](int a, int b) { return csc_row[item_start[a]] < csc_row[item_start[b]]; });
    // factors: positive pseudo-random, rows of U and topics of V normalised
    std::vector<float> U((size_t)n * 64), Vt((size_t)m * 64);
    {
        uint64_t s = 0x9E3779B97F4A7C15ull;
        auto rnd = [&] { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (float)((s >> 40) + 1) * (1.0f / 16777217.0f); };
        for (i64 d = 0; d < n; ++d) { float t = 0; for (int z = 0; z < 64; ++z) t += (U[d * 64 + z] = rnd()); for (int z = 0; z < 64; ++z) U[d * 64 + z] /= t; }
        std::vector<double> cs(64, 0.0);
        for (i64 w = 0; w < m; ++w) for (int z = 0; z < 64; ++z) cs[z] += (Vt[w * 64 + z] = rnd());
        for (i64 w = 0; w < m; ++w) for (int z = 0; z < 64; ++z) Vt[w * 64 + z] = (float)(Vt[w * 64 + z] / cs[z]);
    }
    int *d_indptr = dev(indptr), *d_col = dev(col), *d_order = dev(row_order);
    float *d_val = dev(val);
    int *d_cscrow = dev(csc_row), *d_icol = dev(item_col), *d_ist = dev(item_start), *d_iend = dev(item_end), *d_iord = dev(item_order);
    float *d_cscval = dev(csc_val);
    float *d_U = dev(U), *d_Vt = dev(Vt);
    float *d_Un = dev_alloc<float>((size_t)n * 64), *d_part = dev_alloc<float>((size_t)n_items * 64);
    double *d_colsum = dev_alloc<double>((size_t)cus * 128 * 64 + 64);
    double *d_ll = dev_alloc<double>((size_t)cus * 128 + 64);
    const float thresh = 1e-32f;
    const int grid_cap = cus * 128;
    const int grid_row = (int)std::min<i64>((n + 15) / 16, grid_cap);
    const int grid_col = (int)std::min<i64>((n_items + 15) / 16, grid_cap);
    printf("{\"test\": \"corpus\", \"n\": %lld, \"m\": %lld, \"nnz\": %lld, \"n_items\": %lld, \"grid_row\": %d, \"grid_col\": %d}\n",
           (long long)n, (long long)m, (long long)nnz, (long long)n_items, grid_row, grid_col);
    const double row_gather_gb = nnz * 256.0 / 1e9;
    using S = Shape<16, 1, true>;
    auto report = [&](const char *name, double ms) {
        printf("{\"test\": \"pass\", \"kernel\": \"%s\", \"ms\": %.4f, \"gathered_rows_tb_s\": %.2f}\n", name, ms, row_gather_gb / ms);
        fflush(stdout);
    };
    if (want("row")) {
        report("k_row_pass<fused> (shipped)", time_ms([&] {
            hipLaunchKernelGGL((plsa::k_row_pass<S, false, false>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n,
                               d_order, d_U, d_Vt, (const float *)nullptr, d_Un, (const float *)nullptr, (float *)nullptr, 64, thresh, d_ll,
                               (const int *)nullptr, (const int *)nullptr, 64, (i64)0, (float *)nullptr); }));
        report("k_row_pass<fused,LL> (shipped)", time_ms([&] {
            hipLaunchKernelGGL((plsa::k_row_pass<S, false, true>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n,
                               d_order, d_U, d_Vt, (const float *)nullptr, d_Un, (const float *)nullptr, (float *)nullptr, 64, thresh, d_ll,
                               (const int *)nullptr, (const int *)nullptr, 64, (i64)0, (float *)nullptr); }));
#define ROWV(MODE, UNR, NAME) report(NAME, time_ms([&] { hipLaunchKernelGGL((k_row_variant<MODE, UNR>), dim3(grid_row), dim3(256), 0, g_stream, \
        d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); }))
        ROWV(0, 4, "row variant: full, 4 rows in flight");
        ROWV(1, 4, "row variant: no threshold, 4 in flight");
        ROWV(2, 4, "row variant: gather-only, 4 in flight");
        ROWV(0, 8, "row variant: full, 8 in flight");
        ROWV(1, 8, "row variant: no threshold, 8 in flight");
        ROWV(2, 8, "row variant: gather-only, 8 in flight");
        ROWV(1, 16, "row variant: no threshold, 16 in flight");
        ROWV(2, 16, "row variant: gather-only, 16 in flight");
    }
    if (want("col")) {
#define COLV(MODE, UNR, NAME) report(NAME, time_ms([&] { hipLaunchKernelGGL((k_col_variant<MODE, UNR>), dim3(grid_col), dim3(256), 0, g_stream, \
        d_iord, d_icol, d_ist, d_iend, n_items, d_cscrow, d_cscval, d_U, d_Vt, d_part, thresh); }))
        COLV(0, 8, "col variant: full, 8 rows in flight");
        COLV(1, 8, "col variant: no threshold, 8 in flight");
        COLV(2, 8, "col variant: gather-only, 8 in flight");
        COLV(1, 16, "col variant: no threshold, 16 in flight");
        COLV(2, 16, "col variant: gather-only, 16 in flight");
        COLV(2, 4, "col variant: gather-only, 4 in flight");
    }
    if (want("timeline")) {
        unsigned long long *d_te = dev_alloc<unsigned long long>(grid_col), *d_ts = dev_alloc<unsigned long long>(grid_col);
        unsigned *d_x = dev_alloc<unsigned>(grid_col);
        std::vector<unsigned long long> te(grid_col), ts(grid_col);
        std::vector<unsigned> xc(grid_col);
        for (int assign = 0; assign < 2; ++assign) {
            for (int rep = 0; rep < 2; ++rep) {
                if (assign == 0)
                    hipLaunchKernelGGL((k_col_timeline<0, 8, 0>), dim3(grid_col), dim3(256), 0, g_stream, d_iord, d_icol, d_ist, d_iend, n_items,
                                       d_cscrow, d_cscval, d_U, d_Vt, d_part, thresh, d_te, d_x, d_ts);
                else
                    hipLaunchKernelGGL((k_col_timeline<0, 8, 1>), dim3(grid_col), dim3(256), 0, g_stream, d_iord, d_icol, d_ist, d_iend, n_items,
                                       d_cscrow, d_cscval, d_U, d_Vt, d_part, thresh, d_te, d_x, d_ts);
                HC(hipStreamSynchronize(g_stream));
            }
            HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid_col, hipMemcpyDeviceToHost));
            HC(hipMemcpy(ts.data(), d_ts, sizeof(unsigned long long) * grid_col, hipMemcpyDeviceToHost));
            HC(hipMemcpy(xc.data(), d_x, sizeof(unsigned) * grid_col, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, t1 = 0;
            for (int b = 0; b < grid_col; ++b) { t0 = std::min(t0, ts[b]); t1 = std::max(t1, te[b]); }
            double last[16] = {0}, busy[16] = {0};
            int cnt[16] = {0}, mism = 0;
            for (int b = 0; b < grid_col; ++b) {
                const int x = xc[b] & 15;
                last[x] = std::max(last[x], (double)(te[b] - t0) / 100.0);     // us at 100 MHz
                busy[x] += (double)(te[b] - ts[b]) / 100.0;
                cnt[x]++;
                if (x != (b & 7)) mism++;
            }
            printf("{\"test\": \"col_timeline\", \"assignment\": \"%s\", \"kernel_us\": %.1f, \"blocks_not_on_xcd_b_mod_8\": %d, \"per_xcd\": [",
                   assign == 0 ? "contiguous eighth per XCD (shipped)" : "grid-stride over the whole list", (double)(t1 - t0) / 100.0, mism);
            for (int x = 0; x < 8; ++x) printf("%s{\"xcd\": %d, \"blocks\": %d, \"last_block_ends_us\": %.1f, \"block_time_sum_ms\": %.1f}", x ? ", " : "", x, cnt[x], last[x], busy[x] / 1e3);
            printf("]}\n");
            fflush(stdout);
        }
    }
    if (want("dyn")) {
        // item construction: seg entries per item; optionally columns with >= min_per_band entries per band of `band`
        // documents on average are ALSO cut at the band boundaries (so that all items of a band are visited together)
        struct Cfg { int seg, band, min_per_band; };
        const Cfg cfgs[] = {{256, 0, 0}, {128, 0, 0}, {256, 2048, 16}, {256, 4096, 16}, {256, 8192, 16}, {256, 4096, 64}, {256, 8192, 64},
                            {256, 16384, 32}, {256, 4096, 4}, {512, 0, 0}};
        unsigned long long *d_queue = dev_alloc<unsigned long long>(8);
        for (const Cfg &cf : cfgs) {
            std::vector<int4> recs;
            for (i64 c = 0; c < m; ++c) {
                const int c0 = colptr[c], c1 = colptr[c + 1];
                const bool banded = cf.band > 0 && (double)(c1 - c0) * cf.band / (double)n >= cf.min_per_band;
                int st = c0;
                while (st < c1) {
                    int en = std::min(st + cf.seg, c1);
                    if (banded) {
                        const int b = csc_row[st] / cf.band;
                        // first entry of the next band (entries are in document order)
                        const int lim = (int)(std::lower_bound(csc_row.begin() + st, csc_row.begin() + en, (b + 1) * cf.band) - csc_row.begin());
                        en = std::max(st + 1, lim);
                    }
                    recs.push_back(make_int4((int)c, st, en, 0));
                    st = en;
                }
            }
            const i64 ni = (i64)recs.size();
            std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) { return csc_row[a.y] < csc_row[b.y]; });
            for (i64 i = 0; i < ni; ++i) recs[i].w = (int)i;      // partial slot = visiting position (the reduce step would follow a per-column list)
            int4 *d_items = dev(recs);
            float *d_p2 = dev_alloc<float>((size_t)ni * 64);
            // contiguous range per XCD: equal entry counts
            std::vector<unsigned long long> qinit(8);
            {
                i64 tot = 0, acc_e = 0; for (auto &r : recs) tot += r.z - r.y;
                int lo = 0, x = 0;
                for (i64 i = 0; i < ni && x < 8; ++i) {
                    acc_e += recs[i].z - recs[i].y;
                    if (acc_e * 8 >= tot * (x + 1) || i == ni - 1) {
                        const int hi = (x == 7) ? (int)ni : (int)(i + 1);
                        qinit[x] = ((unsigned long long)lo << 32) | (unsigned long long)((unsigned)hi + QOFF);
                        lo = hi; ++x;
                    }
                }
                for (; x < 8; ++x) qinit[x] = ((unsigned long long)lo << 32) | (unsigned long long)((unsigned)lo + QOFF);
            }
            unsigned long long *d_qinit = dev(qinit);
            for (int steal = 1; steal >= 0; --steal) {
                for (int mode : {0, 2}) {
                    const int grid = cus * 8;
                    auto go = [&] {
                        HC(hipMemcpyAsync(d_queue, d_qinit, 64, hipMemcpyDeviceToDevice, g_stream));
                        if (mode == 0) hipLaunchKernelGGL((k_col_dyn<0, 8>), dim3(grid), dim3(256), 0, g_stream, d_items, d_queue, d_cscrow, d_cscval, d_U, d_Vt, d_p2, thresh, steal, 0);
                        else hipLaunchKernelGGL((k_col_dyn<2, 8>), dim3(grid), dim3(256), 0, g_stream, d_items, d_queue, d_cscrow, d_cscval, d_U, d_Vt, d_p2, thresh, steal, 0);
                    };
                    const double ms = time_ms(go);
                    printf("{\"test\": \"col_dyn\", \"seg\": %d, \"band\": %d, \"min_per_band\": %d, \"n_items\": %lld, \"steal\": %d, \"mode\": \"%s\", \"ms\": %.4f}\n",
                           cf.seg, cf.band, cf.min_per_band, (long long)ni, steal, mode == 0 ? "full" : "gather-only", ms);
                    fflush(stdout);
                }
            }
            HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_qinit));
        }
    }
    if (want("balance")) {
        for (int seg : {192, 160, 128, 96, 64}) {
            std::vector<int4> recs;
            for (i64 c = 0; c < m; ++c)
                for (int st = colptr[c]; st < colptr[c + 1]; st += seg) recs.push_back(make_int4((int)c, st, std::min(st + seg, colptr[c + 1]), 0));
            const i64 ni = (i64)recs.size();
            std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) { return csc_row[a.y] < csc_row[b.y]; });
            const int n_chunks = (int)((ni + 15) / 16);
            int4 *d_items = dev(recs);
            float *d_p2 = dev_alloc<float>((size_t)ni * 64);
            double *d_cs = dev_alloc<double>((size_t)n_chunks * 64);
            std::vector<int> lo(9);
            for (int x = 0; x <= 8; ++x) lo[x] = (int)((i64)n_chunks * x / 8);
            int *d_lo = dev(lo);
            const int grid = std::min(n_chunks + 8, cus * 128) / 8 * 8;
            unsigned long long *d_te = dev_alloc<unsigned long long>(grid);
            std::vector<unsigned long long> te(grid);
            auto run_timed = [&](double T[8]) {
                HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
                HC(hipMemsetAsync(d_te, 0, sizeof(unsigned long long) * grid, g_stream));
                hipLaunchKernelGGL((k_col_chunks<0, 8, true>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
                HC(hipStreamSynchronize(g_stream));
                HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull;
                unsigned long long last[8] = {0};
                for (int b = 0; b < grid; ++b) if (te[b]) { t0 = std::min(t0, te[b]); last[b & 7] = std::max(last[b & 7], te[b]); }
                for (int x = 0; x < 8; ++x) T[x] = (double)(last[x] - t0) / 100.0;     // us after the FIRST block end (relative is enough)
            };
            for (int iter = 0; iter < 4; ++iter) {
                HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
                const double ms = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                const double msg = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                double T[8];
                run_timed(T); run_timed(T);
                printf("{\"test\": \"col_balance\", \"seg\": %d, \"n_items\": %lld, \"iter\": %d, \"ms\": %.4f, \"ms_gather_only\": %.4f, \"lo\": [%d,%d,%d,%d,%d,%d,%d,%d,%d], \"xcd_end_us\": [%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f,%.0f]}\n",
                       seg, (long long)ni, iter, ms, msg, lo[0], lo[1], lo[2], lo[3], lo[4], lo[5], lo[6], lo[7], lo[8], T[0], T[1], T[2], T[3], T[4], T[5], T[6], T[7]);
                fflush(stdout);
                // rebalance: range sizes proportional to size / time (damped), renormalised
                double size[8], tot = 0, mean = 0;
                for (int x = 0; x < 8; ++x) mean += T[x] / 8;
                for (int x = 0; x < 8; ++x) { size[x] = (lo[x + 1] - lo[x]) * (1.0 + 0.8 * (mean / std::max(T[x], 1.0) - 1.0)); tot += size[x]; }
                double accs = 0;
                for (int x = 0; x < 8; ++x) { accs += size[x]; lo[x + 1] = (int)(accs / tot * n_chunks + 0.5); }
                lo[8] = n_chunks;
            }
            HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_cs)); HC(hipFree(d_lo)); HC(hipFree(d_te));
        }
    }
    if (want("ntcol")) {
        for (int seg : {64, 128}) {
            for (int nt_below : {0, 1000, 10000, 100000}) {      // columns with fewer entries than this gather non-temporally
                std::vector<int4> recs;
                for (i64 c = 0; c < m; ++c)
                    for (int st = colptr[c]; st < colptr[c + 1]; st += seg)
                        recs.push_back(make_int4((int)c, st, std::min(st + seg, colptr[c + 1]), (colptr[c + 1] - colptr[c]) < nt_below ? 1 : 0));
                const i64 ni = (i64)recs.size();
                std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) { return csc_row[a.y] < csc_row[b.y]; });
                const int n_chunks = (int)((ni + 15) / 16);
                int4 *d_items = dev(recs);
                float *d_p2 = dev_alloc<float>((size_t)ni * 64);
                double *d_cs = dev_alloc<double>((size_t)n_chunks * 64);
                std::vector<int> lo(9);
                for (int x = 0; x <= 8; ++x) lo[x] = (int)((i64)n_chunks * x / 8);
                int *d_lo = dev(lo);
                const int grid = std::min(n_chunks + 8, cus * 128) / 8 * 8;
                unsigned long long *d_te = dev_alloc<unsigned long long>(grid);
                std::vector<unsigned long long> te(grid);
                double ms = 0, msg = 0;
                for (int iter = 0; iter < 4; ++iter) {
                    HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
                    ms = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                    msg = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                    HC(hipMemsetAsync(d_te, 0, sizeof(unsigned long long) * grid, g_stream));
                    hipLaunchKernelGGL((k_col_chunks<0, 8, true>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
                    HC(hipStreamSynchronize(g_stream));
                    HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ull, last[8] = {0};
                    for (int b = 0; b < grid; ++b) if (te[b]) { t0 = std::min(t0, te[b]); last[b & 7] = std::max(last[b & 7], te[b]); }
                    double T[8], mean = 0, size[8], tot = 0, accs = 0;
                    for (int x = 0; x < 8; ++x) { T[x] = std::max(1.0, (double)(last[x] - t0) / 100.0 + 100.0); mean += T[x] / 8; }
                    for (int x = 0; x < 8; ++x) { size[x] = (lo[x + 1] - lo[x]) * (1.0 + 0.8 * (mean / T[x] - 1.0)); tot += size[x]; }
                    for (int x = 0; x < 8; ++x) { accs += size[x]; lo[x + 1] = (int)(accs / tot * n_chunks + 0.5); }
                    lo[8] = n_chunks;
                }
                printf("{\"test\": \"col_nt\", \"seg\": %d, \"nt_for_columns_below\": %d, \"ms\": %.4f, \"ms_gather_only\": %.4f}\n", seg, nt_below, ms, msg);
                fflush(stdout);
                HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_cs)); HC(hipFree(d_lo)); HC(hipFree(d_te));
            }
        }
    }
    if (want("order")) {
        // visiting orders at 64-entry items, measured boundaries (4 rounds): does making a chunk's 16 items alike help?
        //   band 0: ascending first document (shipped).   band B: (first document / B), then column length descending
        const int seg = 64;
        for (int band : {0, 1024, 2048, 4096}) {
            std::vector<int4> recs;
            for (i64 c = 0; c < m; ++c)
                for (int st = colptr[c]; st < colptr[c + 1]; st += seg) recs.push_back(make_int4((int)c, st, std::min(st + seg, colptr[c + 1]), 0));
            const i64 ni = (i64)recs.size();
            if (band == 0)
                std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) { return csc_row[a.y] < csc_row[b.y]; });
            else
                std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) {
                    const int B = band > 0 ? band : -band;      // negative: ascending column length inside a band
                    const int ba = csc_row[a.y] / B, bb = csc_row[b.y] / B;
                    if (ba != bb) return ba < bb;
                    const int la = colptr[a.x + 1] - colptr[a.x], lb = colptr[b.x + 1] - colptr[b.x];
                    return band > 0 ? la > lb : la < lb; });
            const int n_chunks = (int)((ni + 15) / 16);
            int4 *d_items = dev(recs);
            float *d_p2 = dev_alloc<float>((size_t)ni * 64);
            double *d_cs = dev_alloc<double>((size_t)n_chunks * 64);
            std::vector<int> lo(9);
            for (int x = 0; x <= 8; ++x) lo[x] = (int)((i64)n_chunks * x / 8);
            int *d_lo = dev(lo);
            int grid = (n_chunks / 8 + 8) * 8;      // one chunk per workgroup: 8 x the longest stretch (recomputed every round)
            unsigned long long *d_te = dev_alloc<unsigned long long>(grid);
            std::vector<unsigned long long> te(grid);
            double ms = 0, msg = 0;
            for (int iter = 0; iter < 4; ++iter) {
                { int longest = 1; for (int x = 0; x < 8; ++x) longest = std::max(longest, lo[x + 1] - lo[x]); grid = 8 * longest; }
                if ((int)te.size() < grid) { te.resize(grid); HC(hipFree(d_te)); d_te = dev_alloc<unsigned long long>(grid); }
                HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
                ms = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                msg = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                HC(hipMemsetAsync(d_te, 0, sizeof(unsigned long long) * grid, g_stream));
                hipLaunchKernelGGL((k_col_chunks<0, 8, true>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
                HC(hipStreamSynchronize(g_stream));
                HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, last[8] = {0};
                for (int b = 0; b < grid; ++b) if (te[b]) { t0 = std::min(t0, te[b]); last[b & 7] = std::max(last[b & 7], te[b]); }
                double T[8], mean = 0, size[8], tot = 0, accs = 0;
                for (int x = 0; x < 8; ++x) { T[x] = std::max(1.0, (double)(last[x] - t0) / 100.0 + 100.0); mean += T[x] / 8; }
                for (int x = 0; x < 8; ++x) { size[x] = (lo[x + 1] - lo[x]) * (1.0 + 0.8 * (mean / T[x] - 1.0)); tot += size[x]; }
                for (int x = 0; x < 8; ++x) { accs += size[x]; lo[x + 1] = (int)(accs / tot * n_chunks + 0.5); }
                lo[8] = n_chunks;
                printf("{\"test\": \"col_order\", \"seg\": %d, \"band\": %d, \"round\": %d, \"ms\": %.4f, \"ms_gather_only\": %.4f}\n", seg, band, iter, ms, msg);
                fflush(stdout);
            }
            HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_cs)); HC(hipFree(d_lo)); HC(hipFree(d_te));
        }
    }
    if (want("hitmiss")) {
        // Round 4, VERDICT r03 item 2b: the SHIPPED column schedule (64-entry items, band-major order of 2048 documents,
        // head words first, one chunk per workgroup, measured XCD boundaries) run against three P(z|d) row maps:
        //   real      the corpus' own document ids                         (61 % of the gathers hit the 4 MB L2s)
        //   all-hit   document id & 8191: a 2 MB table, L2-resident on every XCD
        //   all-miss  a pseudo-random row of the same 256 MB table per ENTRY (no reuse distance shorter than the table)
        // full arithmetic (MODE 0) and gather-only (MODE 2), at several (rows in flight, waves per SIMD) points.  If a CU
        // overlapped hits with misses perfectly the real pass would take max(0.61 t_hit, 0.39 t_miss); if its request
        // slots are one shared pool (Little's law) it takes 0.61 t_hit + 0.39 t_miss.
        const int seg = 64, band = 2048;
        std::vector<int4> recs;
        for (i64 c = 0; c < m; ++c)
            for (int st = colptr[c]; st < colptr[c + 1]; st += seg) recs.push_back(make_int4((int)c, st, std::min(st + seg, colptr[c + 1]), 0));
        const i64 ni = (i64)recs.size();
        std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) {
            const int ba = csc_row[a.y] / band, bb = csc_row[b.y] / band;
            if (ba != bb) return ba < bb;
            return colptr[a.x + 1] - colptr[a.x] > colptr[b.x + 1] - colptr[b.x]; });
        const int n_chunks = (int)((ni + 15) / 16);
        int4 *d_items = dev(recs);
        float *d_p2 = dev_alloc<float>((size_t)ni * 64);
        double *d_cs = dev_alloc<double>((size_t)n_chunks * 64);
        std::vector<int> lo(9);
        for (int x = 0; x <= 8; ++x) lo[x] = (int)((i64)n_chunks * x / 8);
        int *d_lo = dev(lo);
        int grid = (n_chunks / 8 + 8) * 8;
        unsigned long long *d_te = dev_alloc<unsigned long long>((size_t)n_chunks * 8 + 16);
        std::vector<unsigned long long> te((size_t)n_chunks * 8 + 16);
        for (int iter = 0; iter < 5; ++iter) {       // measured boundaries, as plsa_hip.hip::ensure_balance
            { int longest = 1; for (int x = 0; x < 8; ++x) longest = std::max(longest, lo[x + 1] - lo[x]); grid = 8 * longest; }
            HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
            HC(hipMemsetAsync(d_te, 0, sizeof(unsigned long long) * grid, g_stream));
            hipLaunchKernelGGL((k_col_chunks<0, 8, true>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
            HC(hipStreamSynchronize(g_stream));
            HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, last[8] = {0};
            for (int b = 0; b < grid; ++b) if (te[b]) { t0 = std::min(t0, te[b]); last[b & 7] = std::max(last[b & 7], te[b]); }
            double T[8], mean = 0, size[8], tot = 0, accs = 0, lo_t = 1e300, hi_t = 0;
            for (int x = 0; x < 8; ++x) { T[x] = std::max(1.0, (double)(last[x] - t0) / 100.0 + 100.0); mean += T[x] / 8; lo_t = std::min(lo_t, T[x]); hi_t = std::max(hi_t, T[x]); }
            if (iter == 4 || (hi_t - lo_t) / mean < 0.02) break;
            for (int x = 0; x < 8; ++x) { size[x] = (lo[x + 1] - lo[x]) * (1.0 + 0.8 * (mean / T[x] - 1.0)); tot += size[x]; }
            for (int x = 0; x < 8; ++x) { accs += size[x]; lo[x + 1] = (int)(accs / tot * n_chunks + 0.5); }
            lo[8] = n_chunks;
        }
        { int longest = 1; for (int x = 0; x < 8; ++x) longest = std::max(longest, lo[x + 1] - lo[x]); grid = 8 * longest; }
        HC(hipMemcpy(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice));
        std::vector<int> row_hit(nnz), row_miss(nnz);
        for (i64 j = 0; j < nnz; ++j) {
            row_hit[j] = csc_row[j] & 8191;
            row_miss[j] = (int)(((uint64_t)csc_row[j] * 2654435761ull + (uint64_t)j * 0x9E3779B97F4A7C15ull) % (uint64_t)n);
        }
        int *d_rhit = dev(row_hit), *d_rmiss = dev(row_miss);
        const int *maps[3] = {d_cscrow, d_rhit, d_rmiss};
        const char *map_name[3] = {"real", "all-hit (2 MB table)", "all-miss (random row of the 256 MB table per entry)"};
#define HM_CASE(UNRV, WV, TAGBASE)                                                                                          \
        for (int mp = 0; mp < 3; ++mp) {                                                                                    \
            const int *rows_ = maps[mp];                                                                                    \
            double full_ = 0, gath_ = 0;                                                                                    \
            if (mp == 0) { full_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, UNRV, false, TAGBASE + 0, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, rows_, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); }); \
                           gath_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, UNRV, false, TAGBASE + 0, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, rows_, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); }); } \
            if (mp == 1) { full_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, UNRV, false, TAGBASE + 1, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, rows_, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); }); \
                           gath_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, UNRV, false, TAGBASE + 1, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, rows_, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); }); } \
            if (mp == 2) { full_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, UNRV, false, TAGBASE + 2, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, rows_, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); }); \
                           gath_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, UNRV, false, TAGBASE + 2, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, rows_, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); }); } \
            printf("{\"test\": \"col_hitmiss\", \"rows_in_flight\": %d, \"min_waves_per_simd\": %d, \"tag\": %d, \"row_map\": \"%s\", \"ms_full\": %.4f, \"ms_gather_only\": %.4f, \"rows_per_ns_gather_only\": %.1f}\n", \
                   UNRV, WV, TAGBASE + mp, map_name[mp], full_, gath_, nnz / gath_ / 1e6);                                  \
            fflush(stdout);                                                                                                 \
        }
        HM_CASE(8, 1, 10)
        HM_CASE(8, 6, 20)
        HM_CASE(8, 8, 30)
        HM_CASE(4, 8, 40)
        HM_CASE(6, 6, 50)
        HM_CASE(16, 1, 60)
#undef HM_CASE
        HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_cs)); HC(hipFree(d_lo)); HC(hipFree(d_te)); HC(hipFree(d_rhit)); HC(hipFree(d_rmiss));
    }
    if (want("mix")) {
        // Round 4, VERDICT r03 item 2c: does mixing word classes INSIDE a chunk (= inside a workgroup / wave) let L2 hits
        // overlap L2 misses better than the shipped homogeneous chunks?  All items are 64 entries long (the rare words'
        // single short items are <3 % of the items), so a mixed chunk wastes no lanes.  Orders inside a band of 2048 documents:
        //   0  head words first (shipped)
        //   1  halves: position j of the first half next to position j of the second half, 8 + 8 per chunk
        //   2  riffle: most frequent, least frequent, 2nd most frequent, 2nd least frequent, ...
        //   3  quarters: 4 + 4 + 4 + 4 per chunk
        const int seg = 64, band = 2048;
        for (int mode = 0; mode < 4; ++mode) {
            std::vector<int4> recs;
            for (i64 c = 0; c < m; ++c)
                for (int st = colptr[c]; st < colptr[c + 1]; st += seg) recs.push_back(make_int4((int)c, st, std::min(st + seg, colptr[c + 1]), 0));
            const i64 ni = (i64)recs.size();
            std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) {
                const int ba = csc_row[a.y] / band, bb = csc_row[b.y] / band;
                if (ba != bb) return ba < bb;
                return colptr[a.x + 1] - colptr[a.x] > colptr[b.x + 1] - colptr[b.x]; });
            if (mode > 0) {
                std::vector<int4> out(recs.size());
                i64 b0 = 0;
                while (b0 < ni) {
                    i64 b1 = b0;
                    const int bd = csc_row[recs[b0].y] / band;
                    while (b1 < ni && csc_row[recs[b1].y] / band == bd) ++b1;
                    const i64 L = b1 - b0;
                    const int parts = mode == 1 ? 2 : (mode == 3 ? 4 : 0);
                    if (mode == 2) {
                        for (i64 j = 0; j < L; ++j) out[b0 + j] = recs[b0 + ((j & 1) ? L - 1 - j / 2 : j / 2)];
                    } else {
                        // `parts` strands of the sorted list, taken 16 / parts at a time from each in turn
                        const int take = 16 / parts;
                        std::vector<i64> pos(parts), end(parts);
                        for (int q = 0; q < parts; ++q) { pos[q] = b0 + L * q / parts; end[q] = b0 + L * (q + 1) / parts; }
                        i64 o = b0;
                        while (o < b1)
                            for (int q = 0; q < parts; ++q)
                                for (int t = 0; t < take && pos[q] < end[q]; ++t) out[o++] = recs[pos[q]++];
                    }
                    b0 = b1;
                }
                recs.swap(out);
            }
            const int n_chunks = (int)((ni + 15) / 16);
            int4 *d_items = dev(recs);
            float *d_p2 = dev_alloc<float>((size_t)ni * 64);
            double *d_cs = dev_alloc<double>((size_t)n_chunks * 64);
            std::vector<int> lo(9);
            for (int x = 0; x <= 8; ++x) lo[x] = (int)((i64)n_chunks * x / 8);
            int *d_lo = dev(lo);
            int grid = 8;
            unsigned long long *d_te = dev_alloc<unsigned long long>((size_t)n_chunks * 8 + 16);
            std::vector<unsigned long long> te((size_t)n_chunks * 8 + 16);
            double ms = 0, msg = 0;
            for (int iter = 0; iter < 5; ++iter) {
                { int longest = 1; for (int x = 0; x < 8; ++x) longest = std::max(longest, lo[x + 1] - lo[x]); grid = 8 * longest; }
                HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
                ms = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                msg = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                HC(hipMemsetAsync(d_te, 0, sizeof(unsigned long long) * grid, g_stream));
                hipLaunchKernelGGL((k_col_chunks<0, 8, true>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
                HC(hipStreamSynchronize(g_stream));
                HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
                unsigned long long t0 = ~0ull, last[8] = {0};
                for (int b = 0; b < grid; ++b) if (te[b]) { t0 = std::min(t0, te[b]); last[b & 7] = std::max(last[b & 7], te[b]); }
                double T[8], mean = 0, size[8], tot = 0, accs = 0;
                for (int x = 0; x < 8; ++x) { T[x] = std::max(1.0, (double)(last[x] - t0) / 100.0 + 100.0); mean += T[x] / 8; }
                for (int x = 0; x < 8; ++x) { size[x] = (lo[x + 1] - lo[x]) * (1.0 + 0.8 * (mean / T[x] - 1.0)); tot += size[x]; }
                for (int x = 0; x < 8; ++x) { accs += size[x]; lo[x + 1] = (int)(accs / tot * n_chunks + 0.5); }
                lo[8] = n_chunks;
                printf("{\"test\": \"col_mix\", \"order_in_band\": %d, \"round\": %d, \"ms\": %.4f, \"ms_gather_only\": %.4f}\n", mode, iter, ms, msg);
                fflush(stdout);
            }
            HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_cs)); HC(hipFree(d_lo)); HC(hipFree(d_te));
        }
    }
    if (want("headsplit")) {
        // Round 4: how much of the column pass is the Zipf-HEAD words (their gathers are the L2 hits)?  Shipped schedule,
        // items of the R most frequent words removed / kept alone.  If t(all) ~ t(without head) the head's gathers are already
        // hidden behind the rare words' misses and serving them from LDS tiles instead of the L2 cannot pay.
        const int seg = 64, band = 2048;
        std::vector<int> by_len(m);
        std::iota(by_len.begin(), by_len.end(), 0);
        std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b) { return colptr[a + 1] - colptr[a] > colptr[b + 1] - colptr[b]; });
        std::vector<int> rank_of(m);
        for (i64 r = 0; r < m; ++r) rank_of[by_len[r]] = (int)r;
        for (int R : {0, 128, 1024}) {
            for (int keep_head = 0; keep_head < (R ? 2 : 1); ++keep_head) {
                std::vector<int4> recs;
                i64 entries = 0;
                for (i64 c = 0; c < m; ++c) {
                    const bool head = rank_of[c] < R;
                    if (R && (head != (keep_head == 1))) continue;
                    for (int st = colptr[c]; st < colptr[c + 1]; st += seg) recs.push_back(make_int4((int)c, st, std::min(st + seg, colptr[c + 1]), 0));
                    entries += colptr[c + 1] - colptr[c];
                }
                const i64 ni = (i64)recs.size();
                std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) {
                    const int ba = csc_row[a.y] / band, bb = csc_row[b.y] / band;
                    if (ba != bb) return ba < bb;
                    return colptr[a.x + 1] - colptr[a.x] > colptr[b.x + 1] - colptr[b.x]; });
                const int n_chunks = (int)((ni + 15) / 16);
                int4 *d_items = dev(recs);
                float *d_p2 = dev_alloc<float>((size_t)ni * 64);
                double *d_cs = dev_alloc<double>((size_t)n_chunks * 64);
                std::vector<int> lo(9);
                for (int x = 0; x <= 8; ++x) lo[x] = (int)((i64)n_chunks * x / 8);
                int *d_lo = dev(lo);
                int grid = 8;
                unsigned long long *d_te = dev_alloc<unsigned long long>((size_t)n_chunks * 8 + 16);
                std::vector<unsigned long long> te((size_t)n_chunks * 8 + 16);
                double ms = 0, msg = 0;
                for (int iter = 0; iter < 5; ++iter) {
                    { int longest = 1; for (int x = 0; x < 8; ++x) longest = std::max(longest, lo[x + 1] - lo[x]); grid = 8 * longest; }
                    HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
                    ms = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                    msg = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
                    HC(hipMemsetAsync(d_te, 0, sizeof(unsigned long long) * grid, g_stream));
                    hipLaunchKernelGGL((k_col_chunks<0, 8, true>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
                    HC(hipStreamSynchronize(g_stream));
                    HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
                    unsigned long long t0 = ~0ull, last[8] = {0};
                    for (int b = 0; b < grid; ++b) if (te[b]) { t0 = std::min(t0, te[b]); last[b & 7] = std::max(last[b & 7], te[b]); }
                    double T[8], mean = 0, size[8], tot = 0, accs = 0;
                    for (int x = 0; x < 8; ++x) { T[x] = std::max(1.0, (double)(last[x] - t0) / 100.0 + 100.0); mean += T[x] / 8; }
                    for (int x = 0; x < 8; ++x) { size[x] = (lo[x + 1] - lo[x]) * (1.0 + 0.8 * (mean / T[x] - 1.0)); tot += size[x]; }
                    for (int x = 0; x < 8; ++x) { accs += size[x]; lo[x + 1] = (int)(accs / tot * n_chunks + 0.5); }
                    lo[8] = n_chunks;
                }
                printf("{\"test\": \"col_headsplit\", \"head_words\": %d, \"part\": \"%s\", \"entries\": %lld, \"share_of_nnz\": %.3f, \"items\": %lld, \"ms\": %.4f, \"ms_gather_only\": %.4f}\n",
                       R, R == 0 ? "all words" : (keep_head ? "head words only" : "all but the head words"), (long long)entries, (double)entries / nnz, (long long)ni, ms, msg);
                fflush(stdout);
                HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_cs)); HC(hipFree(d_lo)); HC(hipFree(d_te));
            }
        }
    }
    if (want("rowhot")) {
        // the R most frequent words (by document frequency) -> LDS slots; tagged column ids
        std::vector<int> by_len(m);
        std::iota(by_len.begin(), by_len.end(), 0);
        std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b) { return colptr[a + 1] - colptr[a] > colptr[b + 1] - colptr[b]; });
        const double base = time_ms([&] { hipLaunchKernelGGL((plsa::k_row_pass<S, false, false>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, (const float *)nullptr, d_Un, (const float *)nullptr, (float *)nullptr, 64, thresh, d_ll, (const int *)nullptr, (const int *)nullptr, 0, (i64)0, (float *)nullptr); });
        printf("{\"test\": \"row_hot\", \"variant\": \"shipped k_row_pass<fused>\", \"ms\": %.4f}\n", base);
        fflush(stdout);
        for (int R : {0, 64, 128, 256, 512}) {
            std::vector<int> slot(m, -1), hot_words(std::max(R, 1), 0);
            i64 hot_entries = 0;
            for (int r = 0; r < R; ++r) { slot[by_len[r]] = r; hot_words[r] = by_len[r]; hot_entries += colptr[by_len[r] + 1] - colptr[by_len[r]]; }
            std::vector<int> tag(nnz);
            for (i64 j = 0; j < nnz; ++j) tag[j] = slot[col[j]] >= 0 ? ~slot[col[j]] : col[j];
            int *d_tag = dev(tag), *d_hot = dev(hot_words);
            const size_t smem = (size_t)R * 256;
#define RH_CASE(TT, BPC)                                                                                                     \
            if (smem * (BPC) <= 160 * 1024 && (TT) * (BPC) <= 2048) {                                                        \
                const int grid = cus * (BPC);                                                                                \
                HC(hipFuncSetAttribute((const void *)k_row_hot<0, 4, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                HC(hipFuncSetAttribute((const void *)k_row_hot<2, 4, TT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)); \
                const double f_ = time_ms([&] { hipLaunchKernelGGL((k_row_hot<0, 4, TT>), dim3(grid), dim3(TT), smem, g_stream, d_indptr, d_tag, d_val, (int)n, d_order, d_U, d_Vt, d_hot, R, d_Un, thresh); }); \
                const double g_ = time_ms([&] { hipLaunchKernelGGL((k_row_hot<2, 4, TT>), dim3(grid), dim3(TT), smem, g_stream, d_indptr, d_tag, d_val, (int)n, d_order, d_U, d_Vt, d_hot, R, d_Un, thresh); }); \
                printf("{\"test\": \"row_hot\", \"hot_words\": %d, \"hot_share_of_nnz\": %.3f, \"threads\": %d, \"workgroups_per_cu\": %d, \"lds_kb\": %.0f, \"ms\": %.4f, \"ms_gather_only\": %.4f}\n", \
                       R, (double)hot_entries / nnz, TT, BPC, smem / 1024.0, f_, g_);                                        \
                fflush(stdout);                                                                                              \
            }
            RH_CASE(256, 8) RH_CASE(256, 4) RH_CASE(512, 4) RH_CASE(512, 2) RH_CASE(1024, 2) RH_CASE(1024, 1)
#undef RH_CASE
            HC(hipFree(d_tag)); HC(hipFree(d_hot));
        }
    }
    if (want("rowcold")) {
        // document pass over the entries of all but the R most frequent words / of those words alone (the counterpart of
        // "headsplit" for the column pass): what would remain of the pass if the head region were computed elsewhere
        std::vector<int> by_len(m);
        std::iota(by_len.begin(), by_len.end(), 0);
        std::stable_sort(by_len.begin(), by_len.end(), [&](int a, int b) { return colptr[a + 1] - colptr[a] > colptr[b + 1] - colptr[b]; });
        for (int R : {0, 128, 1024}) {
            std::vector<char> head(m, 0);
            for (int r = 0; r < R; ++r) head[by_len[r]] = 1;
            for (int keep_head = 0; keep_head < (R ? 2 : 1); ++keep_head) {
                std::vector<int> ip(n + 1, 0), cc;
                std::vector<float> vv;
                cc.reserve(nnz); vv.reserve(nnz);
                for (i64 d = 0; d < n; ++d) {
                    for (int j = indptr[d]; j < indptr[d + 1]; ++j)
                        if (!R || (head[col[j]] != 0) == (keep_head == 1)) { cc.push_back(col[j]); vv.push_back(val[j]); }
                    ip[d + 1] = (int)cc.size();
                }
                std::vector<int> ord(n);
                std::iota(ord.begin(), ord.end(), 0);
                std::stable_sort(ord.begin(), ord.end(), [&](int a, int b) { return ip[a + 1] - ip[a] > ip[b + 1] - ip[b]; });
                int *d_ip = dev(ip), *d_cc = dev(cc), *d_ord = dev(ord);
                float *d_vv = dev(vv);
                const double shipped = time_ms([&] { hipLaunchKernelGGL((plsa::k_row_pass<S, false, false>), dim3(grid_row), dim3(256), 0, g_stream, d_ip, d_cc, d_vv, (int)n, d_ord, d_U, d_Vt, (const float *)nullptr, d_Un, (const float *)nullptr, (float *)nullptr, 64, thresh, d_ll, (const int *)nullptr, (const int *)nullptr, 0, (i64)0, (float *)nullptr); });
                const double full = time_ms([&] { hipLaunchKernelGGL((k_row_variant<0, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_ip, d_cc, d_vv, (int)n, d_ord, d_U, d_Vt, d_Un, thresh); });
                const double gath = time_ms([&] { hipLaunchKernelGGL((k_row_variant<2, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_ip, d_cc, d_vv, (int)n, d_ord, d_U, d_Vt, d_Un, thresh); });
                printf("{\"test\": \"row_cold\", \"head_words\": %d, \"part\": \"%s\", \"entries\": %lld, \"share_of_nnz\": %.3f, \"ms_shipped_kernel\": %.4f, \"ms_variant\": %.4f, \"ms_gather_only\": %.4f}\n",
                       R, R == 0 ? "all words" : (keep_head ? "head words only" : "all but the head words"), (long long)cc.size(), (double)cc.size() / nnz, shipped, full, gath);
                fflush(stdout);
                HC(hipFree(d_ip)); HC(hipFree(d_cc)); HC(hipFree(d_ord)); HC(hipFree(d_vv));
            }
        }
    }
    if (want("rowhitmiss")) {
        // Round 4: the document pass (shipped kernel, 8 lanes x 2 chunks at k = 64) against three P(w|z) row maps:
        // real word ids / word id & 8191 (2 MB table: every gather an L2 hit) / a pseudo-random word per ENTRY (the
        // 25.6 MB table misses the 4 MB L2s and is served by the Infinity Cache).  rocprofv3 --pmc passes tell the
        // three launches apart by the TAG template argument of the gather-only variant and by launch order for the
        // shipped kernel (real, all-hit, all-miss, in that order, after the warm-up of each).
        std::vector<int> col_hit(nnz), col_miss(nnz);
        for (i64 j = 0; j < nnz; ++j) {
            col_hit[j] = col[j] & 8191;
            col_miss[j] = (int)(((uint64_t)col[j] * 2654435761ull + (uint64_t)j * 0x9E3779B97F4A7C15ull) % (uint64_t)m);
        }
        int *d_chit = dev(col_hit), *d_cmiss = dev(col_miss);
        const int *maps[3] = {d_col, d_chit, d_cmiss};
        const char *map_name[3] = {"real", "all-hit (2 MB table)", "all-miss in L2 (random row of the 25.6 MB table per entry)"};
        using S82 = Shape<8, 2, true>;
        const int grid82 = (int)std::min<i64>((n + 31) / 32, grid_cap);
        for (int mp = 0; mp < 3; ++mp) {
            const int *cc = maps[mp];
            const double full = time_ms([&] { hipLaunchKernelGGL((plsa::k_row_pass<S82, false, false, false>), dim3(grid82), dim3(256), 0, g_stream, d_indptr, cc, d_val, (int)n, d_order, d_U, d_Vt, (const float *)nullptr, d_Un, (const float *)nullptr, (float *)nullptr, 64, thresh, d_ll, (const int *)nullptr, (const int *)nullptr, 0, (i64)0, (float *)nullptr); });
            const double gath = time_ms([&] { hipLaunchKernelGGL((k_row_variant<2, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, cc, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
            printf("{\"test\": \"row_hitmiss\", \"row_map\": \"%s\", \"ms_shipped_kernel\": %.4f, \"ms_gather_only_probe_variant\": %.4f, \"rows_per_ns_shipped\": %.1f}\n",
                   map_name[mp], full, gath, nnz / full / 1e6);
            fflush(stdout);
        }
        HC(hipFree(d_chit)); HC(hipFree(d_cmiss));
    }
    if (want("pipe")) {
        // shipped schedule (64-entry items, band-major, head words first, one chunk per workgroup, measured boundaries):
        // k_col_chunks against the software-pipelined k_col_chunks_pipe; partial sums compared
        const int seg = 64, band = 2048;
        std::vector<int4> recs;
        for (i64 c = 0; c < m; ++c)
            for (int st = colptr[c]; st < colptr[c + 1]; st += seg) recs.push_back(make_int4((int)c, st, std::min(st + seg, colptr[c + 1]), 0));
        const i64 ni = (i64)recs.size();
        std::stable_sort(recs.begin(), recs.end(), [&](const int4 &a, const int4 &b) {
            const int ba = csc_row[a.y] / band, bb = csc_row[b.y] / band;
            if (ba != bb) return ba < bb;
            return colptr[a.x + 1] - colptr[a.x] > colptr[b.x + 1] - colptr[b.x]; });
        const int n_chunks = (int)((ni + 15) / 16);
        int4 *d_items = dev(recs);
        float *d_p2 = dev_alloc<float>((size_t)ni * 64), *d_p3 = dev_alloc<float>((size_t)ni * 64);
        double *d_cs = dev_alloc<double>((size_t)n_chunks * 64);
        std::vector<int> lo(9);
        for (int x = 0; x <= 8; ++x) lo[x] = (int)((i64)n_chunks * x / 8);
        int *d_lo = dev(lo);
        int grid = 8;
        unsigned long long *d_te = dev_alloc<unsigned long long>((size_t)n_chunks * 8 + 16);
        std::vector<unsigned long long> te((size_t)n_chunks * 8 + 16);
        for (int iter = 0; iter < 5; ++iter) {
            { int longest = 1; for (int x = 0; x < 8; ++x) longest = std::max(longest, lo[x + 1] - lo[x]); grid = 8 * longest; }
            HC(hipMemcpyAsync(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice, g_stream));
            HC(hipMemsetAsync(d_te, 0, sizeof(unsigned long long) * grid, g_stream));
            hipLaunchKernelGGL((k_col_chunks<0, 8, true>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
            HC(hipStreamSynchronize(g_stream));
            HC(hipMemcpy(te.data(), d_te, sizeof(unsigned long long) * grid, hipMemcpyDeviceToHost));
            unsigned long long t0 = ~0ull, last[8] = {0};
            for (int b = 0; b < grid; ++b) if (te[b]) { t0 = std::min(t0, te[b]); last[b & 7] = std::max(last[b & 7], te[b]); }
            double T[8], mean = 0, size[8], tot = 0, accs = 0;
            for (int x = 0; x < 8; ++x) { T[x] = std::max(1.0, (double)(last[x] - t0) / 100.0 + 100.0); mean += T[x] / 8; }
            if (iter == 4) break;
            for (int x = 0; x < 8; ++x) { size[x] = (lo[x + 1] - lo[x]) * (1.0 + 0.8 * (mean / T[x] - 1.0)); tot += size[x]; }
            for (int x = 0; x < 8; ++x) { accs += size[x]; lo[x + 1] = (int)(accs / tot * n_chunks + 0.5); }
            lo[8] = n_chunks;
        }
        { int longest = 1; for (int x = 0; x < 8; ++x) longest = std::max(longest, lo[x + 1] - lo[x]); grid = 8 * longest; }
        HC(hipMemcpy(d_lo, lo.data(), sizeof(int) * 9, hipMemcpyHostToDevice));
        for (int rep = 0; rep < 2; ++rep) {
            const double base_f = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<0, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
            const double base_g = time_ms([&] { hipLaunchKernelGGL((k_col_chunks<2, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te); });
            hipLaunchKernelGGL((k_col_chunks<0, 8, false>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p2, d_cs, thresh, d_te);
            printf("{\"test\": \"col_pipe\", \"variant\": \"k_col_chunks (8 rows per batch, no pipelining)\", \"ms\": %.4f, \"ms_gather_only\": %.4f}\n", base_f, base_g);
#define PIPE_CASE(WV)                                                                                                        \
            { const double f_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks_pipe<0, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p3, d_cs, thresh); }); \
              const double g_ = time_ms([&] { hipLaunchKernelGGL((k_col_chunks_pipe<2, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p3, d_cs, thresh); }); \
              hipLaunchKernelGGL((k_col_chunks_pipe<0, WV>), dim3(grid), dim3(256), 0, g_stream, d_items, ni, d_lo, d_cscrow, d_cscval, d_U, d_Vt, d_p3, d_cs, thresh); \
              HC(hipStreamSynchronize(g_stream));                                                                           \
              std::vector<float> h2(1 << 20), h3(1 << 20);                                                                   \
              HC(hipMemcpy(h2.data(), d_p2, sizeof(float) * h2.size(), hipMemcpyDeviceToHost));                             \
              HC(hipMemcpy(h3.data(), d_p3, sizeof(float) * h3.size(), hipMemcpyDeviceToHost));                             \
              double md = 0; for (size_t i = 0; i < h2.size(); ++i) md = std::max(md, (double)std::fabs(h2[i] - h3[i]));     \
              printf("{\"test\": \"col_pipe\", \"variant\": \"software-pipelined (next 8 rows requested before the current 8 are reduced)\", \"min_waves_per_simd\": %d, \"ms\": %.4f, \"ms_gather_only\": %.4f, \"max_abs_diff_of_partials\": %.3g}\n", WV, f_, g_, md); \
              fflush(stdout); }
            PIPE_CASE(1) PIPE_CASE(4) PIPE_CASE(5)
#undef PIPE_CASE
        }
        HC(hipFree(d_items)); HC(hipFree(d_p2)); HC(hipFree(d_p3)); HC(hipFree(d_cs)); HC(hipFree(d_lo)); HC(hipFree(d_te));
    }
    if (want("rowx")) {
        // does the order of a document's entries matter?  as stored (by word id = random w.r.t. frequency) vs sorted by
        // the word's document frequency, descending / ascending (hot words first / last)
        for (int mode = 0; mode < 3; ++mode) {
            std::vector<int> col2(col);
            std::vector<float> val2(val);
            if (mode > 0) {
                std::vector<std::pair<int, int>> tmp;
                for (i64 d = 0; d < n; ++d) {
                    const int j0 = indptr[d], j1 = indptr[d + 1];
                    tmp.resize(j1 - j0);
                    for (int j = j0; j < j1; ++j) tmp[j - j0] = {colptr[col[j] + 1] - colptr[col[j]], j};
                    if (mode == 1) std::stable_sort(tmp.begin(), tmp.end(), [](auto &a, auto &b) { return a.first > b.first; });
                    else std::stable_sort(tmp.begin(), tmp.end(), [](auto &a, auto &b) { return a.first < b.first; });
                    for (int j = j0; j < j1; ++j) { col2[j] = col[tmp[j - j0].second]; val2[j] = val[tmp[j - j0].second]; }
                }
            }
            int *d_col2 = dev(col2);
            float *d_val2 = dev(val2);
            const char *nm = mode == 0 ? "by word id" : (mode == 1 ? "frequent words first" : "rare words first");
            for (int rep = 0; rep < 2; ++rep) {
                const double a = time_ms([&] { hipLaunchKernelGGL((k_row_variant<0, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col2, d_val2, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
                const double b = time_ms([&] { hipLaunchKernelGGL((k_row_variant<2, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col2, d_val2, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
                const double c8 = time_ms([&] { hipLaunchKernelGGL((k_row_variant<0, 8>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col2, d_val2, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
                const double s1 = time_ms([&] {
                    hipLaunchKernelGGL((plsa::k_row_pass<S, false, false>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col2, d_val2, (int)n,
                               d_order, d_U, d_Vt, (const float *)nullptr, d_Un, (const float *)nullptr, (float *)nullptr, 64, thresh, d_ll,
                               (const int *)nullptr, (const int *)nullptr, 64, (i64)0, (float *)nullptr); });
                printf("{\"test\": \"row_entry_order\", \"order\": \"%s\", \"rep\": %d, \"full_unr4_ms\": %.4f, \"gather_only_unr4_ms\": %.4f, \"full_unr8_ms\": %.4f, \"shipped_ms\": %.4f}\n", nm, rep, a, b, c8, s1);
                fflush(stdout);
            }
            HC(hipFree(d_col2)); HC(hipFree(d_val2));
        }
    }
    if (want("rowsplit")) {
        // entries of a document: frequent words (column length >= T) first, then the rest; split position per document
        for (int T : {100000000, 2000, 1000, 500, 200}) {       // first: everything "rare" -> phase 1 alone = the unsplit pass
            std::vector<int> col2(col), split(n);
            std::vector<float> val2(val);
            std::vector<std::pair<int, int>> tmp;
            i64 hot = 0;
            for (i64 d = 0; d < n; ++d) {
                const int j0 = indptr[d], j1 = indptr[d + 1];
                tmp.resize(j1 - j0);
                for (int j = j0; j < j1; ++j) tmp[j - j0] = {(colptr[col[j] + 1] - colptr[col[j]]) >= T ? 0 : 1, j};
                std::stable_sort(tmp.begin(), tmp.end(), [](auto &a, auto &b) { return a.first < b.first; });
                int sp = j0;
                for (int j = j0; j < j1; ++j) { col2[j] = col[tmp[j - j0].second]; val2[j] = val[tmp[j - j0].second]; if (tmp[j - j0].first == 0) sp = j + 1; }
                split[d] = sp; hot += sp - j0;
            }
            int *d_col2 = dev(col2), *d_split = dev(split);
            float *d_val2 = dev(val2);
            float *d_acc = dev_alloc<float>((size_t)n * 64);
            for (int mode : {0, 2}) {
                double a = 0, b = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) {
                        a = time_ms([&] { hipLaunchKernelGGL((k_row_split<0, 4, 0>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_split, d_col2, d_val2, (int)n, d_order, d_U, d_Vt, d_acc, d_Un, thresh); });
                        b = time_ms([&] { hipLaunchKernelGGL((k_row_split<0, 4, 1>), dim3(grid_row), dim3(256), 0, g_stream, d_split, d_indptr + 1, d_col2, d_val2, (int)n, d_order, d_U, d_Vt, d_acc, d_Un, thresh); });
                    } else {
                        a = time_ms([&] { hipLaunchKernelGGL((k_row_split<2, 4, 0>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_split, d_col2, d_val2, (int)n, d_order, d_U, d_Vt, d_acc, d_Un, thresh); });
                        b = time_ms([&] { hipLaunchKernelGGL((k_row_split<2, 4, 1>), dim3(grid_row), dim3(256), 0, g_stream, d_split, d_indptr + 1, d_col2, d_val2, (int)n, d_order, d_U, d_Vt, d_acc, d_Un, thresh); });
                    }
                }
                printf("{\"test\": \"row_split\", \"frequent_if_column_len_ge\": %d, \"frequent_entries_frac\": %.3f, \"mode\": \"%s\", \"phase0_ms\": %.4f, \"phase1_ms\": %.4f, \"sum_ms\": %.4f}\n",
                       T, (double)hot / nnz, mode == 0 ? "full" : "gather-only", a, b, a + b);
                fflush(stdout);
            }
            HC(hipFree(d_col2)); HC(hipFree(d_split)); HC(hipFree(d_val2)); HC(hipFree(d_acc));
        }
    }
    if (want("rowlean")) {
        for (int rep = 0; rep < 3; ++rep) {
            const double a = time_ms([&] { hipLaunchKernelGGL((k_row_variant<0, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
            const double b = time_ms([&] { hipLaunchKernelGGL((k_row_variant<1, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
            const double c0 = time_ms([&] { hipLaunchKernelGGL((k_row_lean<0>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
            const double c1 = time_ms([&] { hipLaunchKernelGGL((k_row_lean<1>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
            const double c2 = time_ms([&] { hipLaunchKernelGGL((k_row_lean<2>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
            const double g = time_ms([&] { hipLaunchKernelGGL((k_row_variant<2, 4>), dim3(grid_row), dim3(256), 0, g_stream, d_indptr, d_col, d_val, (int)n, d_order, d_U, d_Vt, d_Un, thresh); });
            printf("{\"test\": \"row_lean\", \"rep\": %d, \"bpermute_full_ms\": %.4f, \"bpermute_nothresh_ms\": %.4f, \"dpp32_full_ms\": %.4f, \"dpp32_nothresh_ms\": %.4f, \"dpp32_gather_only_ms\": %.4f, \"bpermute_gather_only_ms\": %.4f}\n", rep, a, b, c0, c1, c2, g);
            fflush(stdout);
        }
    }
    return 0;
}
