// estep_probe.hip -- what bounds the flat (one group per non-zero) materialising E-step on a small corpus?  (round 5)
//
// Stand-alone: a random corpus of config 1's shape (documents of ~156 sorted entries, Zipf words), random factors,
// the shipped kernels of plsa_kernels.hpp next to stripped variants of the same traversal:
//   store_only   the tile's P rows are written (constants), nothing is gathered
//   load_only    ids + gathers + arithmetic, one float per lane and tile written
//   packed_pf2   packed slots, the NEXT tile's factor rows are gathered before this tile's rows are stored
// Every full variant is compared bit for bit with k_e_step.
//
//   build:  hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 tools/sol/estep_probe.hip -Ienstop_amd/csrc -o tools/sol/estep_probe
//   run:    tools/sol/estep_probe [k=20] [n=18846] [m=173762] [per_doc=156] [grid_mult=8]
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <vector>
#include "plsa_kernels.hpp"

using plsa::i64;
#define HC(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s (%s:%d)\n", #x, hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

template <class F>
static double time_us(F &&launch, int reps = 20) {
    hipEvent_t a, b;
    HC(hipEventCreate(&a)); HC(hipEventCreate(&b));
    launch(); launch();
    HC(hipDeviceSynchronize());
    double best = 1e30;
    for (int r = 0; r < 5; ++r) {
        HC(hipEventRecord(a, 0));
        for (int i = 0; i < reps; ++i) launch();
        HC(hipEventRecord(b, 0));
        HC(hipDeviceSynchronize());
        float ms = 0.f;
        HC(hipEventElapsedTime(&ms, a, b));
        best = std::min(best, (double)ms * 1e3 / reps);
    }
    HC(hipGetLastError());
    return best;
}

namespace plsa {
// ------------------------------------------------------------------------------------------------
// k_e_step_packed: the flat E-step for topic counts whose C = kp/4 chunks do not fill a power-of-two
// lane group (k = 20: 5 chunks in an 8-lane group, 3 of 8 lanes idle in every gather, product and
// store of k_e_step; k = 10: 3 in 4).  A wave still takes a tile of 64 consecutive non-zeros, but
// the tile's 64*C (entry, chunk) slots are dealt to the lanes in ROW-MAJOR order, slot s = lane + 64 i
// <-> entry s / C, chunk s % C: no idle lane, and slot s of the tile is float4 number s of the tile's
// P rows, so every store instruction of the wave writes 1024 contiguous, line-aligned bytes (k = 20:
// k_e_step writes 8 rows of 80 bytes with holes in the lane mask).  The C partial sums of an entry
// meet through LDS; the norm is added in the SAME order as group_sum's butterfly over the LPN-lane
// group (tree_sum: the pad lanes' exact zeros drop out), so P is bit-identical to k_e_step's.
// Thresholds below TINY_THRESH keep k_e_step (no rescue here).
// ------------------------------------------------------------------------------------------------
template <int C, int LO, int LEN>
__device__ __forceinline__ float tree_sum(const float (&p)[C]) {
    if constexpr (LEN == 1) {
        return p[LO];
    } else {
        constexpr int H = LEN / 2;
        if constexpr (LO + H >= C) return tree_sum<C, LO, H>(p);
        else return tree_sum<C, LO, H>(p) + tree_sum<C, LO + H, H>(p);
    }
}

template <int C, int LPN>
__global__ __launch_bounds__(256, PLSA_WAVES) void k_e_step_packed(const int *__restrict__ rowidx,
                                                       const int *__restrict__ colidx, i64 nnz,
                                                       const float *__restrict__ U,
                                                       const float *__restrict__ Vt, float *__restrict__ P,
                                                       float thresh) {
    constexpr int KP = 4 * C;
    __shared__ float part[4][64 * C];
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    float *mypart = part[wave];
    const i64 tiles = (nnz + 63) >> 6;
    // the (doc, word) ids of the NEXT tile are requested before this tile's stores are issued: vmcnt retires in
    // order, so ids requested behind the stores could only be waited for together with the stores' acknowledgements
    // (three memory latencies per tile in a row -- ids, gathers, store acks -- instead of the longer of two)
    i64 t = (i64)blockIdx.x * 4 + wave;
    int d_n = 0, w_n = 0;
    if (t < tiles) {
        const i64 mine = (t << 6) + lane;
        d_n = mine < nnz ? __builtin_nontemporal_load(rowidx + mine) : 0;
        w_n = mine < nnz ? __builtin_nontemporal_load(colidx + mine) : 0;
    }
    for (; t < tiles; t += (i64)gridDim.x * 4) {
        const i64 base = t << 6;
        const int d_l = d_n, w_l = w_n;
        float4 u[C][1], vt[C][1], keep[C][1];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const unsigned s = (unsigned)lane + 64u * i;
            const unsigned e = s / (unsigned)C, c = s - e * (unsigned)C;
            const int d = __shfl(d_l, (int)e, 64);
            const int w = __shfl(w_l, (int)e, 64);
            u[i][0] = ld4(U + (i64)d * KP + 4 * c);
            vt[i][0] = ld4(Vt + (i64)w * KP + 4 * c);
        }
        {
            const i64 tn = t + (i64)gridDim.x * 4;
            const i64 mine = (tn << 6) + lane;
            const bool in = tn < tiles && mine < nnz;
            d_n = in ? __builtin_nontemporal_load(rowidx + mine) : 0;
            w_n = in ? __builtin_nontemporal_load(colidx + mine) : 0;
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            float unth;
            mypart[lane + 64 * i] = products<1, false>(u[i], vt[i], thresh, keep[i], unth);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float p[C];
#pragma unroll
        for (int j = 0; j < C; ++j) p[j] = mypart[lane * C + j];     // lane = entry of the tile
        const float inv = inv_norm(tree_sum<C, 0, LPN>(p));
        float *prow = P + base * KP;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const unsigned s = (unsigned)lane + 64u * i;
            const float iv = __shfl(inv, (int)(s / (unsigned)C), 64);
            float4 q;
            q.x = keep[i][0].x * iv; q.y = keep[i][0].y * iv; q.z = keep[i][0].z * iv; q.w = keep[i][0].w * iv;
            st4_nt(prow + 4 * s, q);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
}  // namespace plsa

namespace probe {
using namespace plsa;

// the tile's stores only (same addresses and lane mask as k_e_step_packed)
template <int C>
__global__ __launch_bounds__(256) void k_store_only(i64 nnz, float *__restrict__ P) {
    constexpr int KP = 4 * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 tiles = (nnz + 63) >> 6;
    for (i64 t = (i64)blockIdx.x * 4 + wave; t < tiles; t += (i64)gridDim.x * 4) {
        float *prow = P + (t << 6) * KP;
#pragma unroll
        for (int i = 0; i < C; ++i) st4_nt(prow + 4 * (lane + 64 * i), make_float4(1.f, 2.f, 3.f, (float)t));
    }
}

// ids + gathers + arithmetic of k_e_step_packed, one float per lane and tile stored
template <int C, int LPN>
__global__ __launch_bounds__(256) void k_load_only(const int *__restrict__ rowidx, const int *__restrict__ colidx, i64 nnz,
                                                   const float *__restrict__ U, const float *__restrict__ Vt,
                                                   float *__restrict__ sink, float thresh) {
    constexpr int KP = 4 * C;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const i64 tiles = (nnz + 63) >> 6;
    float acc = 0.f;
    for (i64 t = (i64)blockIdx.x * 4 + wave; t < tiles; t += (i64)gridDim.x * 4) {
        const i64 mine = (t << 6) + lane;
        const int d_l = mine < nnz ? __builtin_nontemporal_load(rowidx + mine) : 0;
        const int w_l = mine < nnz ? __builtin_nontemporal_load(colidx + mine) : 0;
        float4 u[C][1], vt[C][1], keep[C][1];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const unsigned s = (unsigned)lane + 64u * i;
            const unsigned e = s / (unsigned)C, c = s - e * (unsigned)C;
            u[i][0] = ld4(U + (i64)__shfl(d_l, (int)e, 64) * KP + 4 * c);
            vt[i][0] = ld4(Vt + (i64)__shfl(w_l, (int)e, 64) * KP + 4 * c);
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            float unth;
            acc += products<1, false>(u[i], vt[i], thresh, keep[i], unth);
        }
    }
    sink[(i64)blockIdx.x * 256 + threadIdx.x] = acc;
}

// packed slots, software-pipelined one tile deep: the NEXT tile's ids AND factor rows are in flight while this
// tile is reduced and stored (its loads are older than this tile's stores in the in-order vmcnt queue)
template <int C, int LPN>
__global__ __launch_bounds__(256) void k_e_step_packed_pf2(const int *__restrict__ rowidx, const int *__restrict__ colidx,
                                                           i64 nnz, const float *__restrict__ U,
                                                           const float *__restrict__ Vt, float *__restrict__ P, float thresh) {
    constexpr int KP = 4 * C;
    __shared__ float part[4][64 * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mypart = part[wave];
    const i64 tiles = (nnz + 63) >> 6;
    const i64 stride = (i64)gridDim.x * 4;
    i64 t = (i64)blockIdx.x * 4 + wave;
    if (t >= tiles) return;
    auto ids = [&](i64 tt, int &d, int &w) {
        const i64 mine = (tt << 6) + lane;
        const bool in = tt < tiles && mine < nnz;
        d = in ? __builtin_nontemporal_load(rowidx + mine) : 0;
        w = in ? __builtin_nontemporal_load(colidx + mine) : 0;
    };
    auto gather = [&](int d_l, int w_l, float4 (&u)[C][1], float4 (&vt)[C][1]) {
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const unsigned s = (unsigned)lane + 64u * i;
            const unsigned e = s / (unsigned)C, c = s - e * (unsigned)C;
            u[i][0] = ld4(U + (i64)__shfl(d_l, (int)e, 64) * KP + 4 * c);
            vt[i][0] = ld4(Vt + (i64)__shfl(w_l, (int)e, 64) * KP + 4 * c);
        }
    };
    int d0, w0, d1, w1;
    ids(t, d0, w0);
    ids(t + stride, d1, w1);
    float4 u[C][1], vt[C][1];
    gather(d0, w0, u, vt);
    for (; t < tiles; t += stride) {
        float4 keep[C][1];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            float unth;
            mypart[lane + 64 * i] = products<1, false>(u[i], vt[i], thresh, keep[i], unth);
        }
        // next tile: rows requested now, ids of the tile after it too
        gather(d1, w1, u, vt);
        ids(t + 2 * stride, d1, w1);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float p[C];
#pragma unroll
        for (int j = 0; j < C; ++j) p[j] = mypart[lane * C + j];
        const float inv = inv_norm(tree_sum<C, 0, LPN>(p));
        float *prow = P + (t << 6) * KP;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const unsigned s = (unsigned)lane + 64u * i;
            const float iv = __shfl(inv, (int)(s / (unsigned)C), 64);
            float4 q;
            q.x = keep[i][0].x * iv; q.y = keep[i][0].y * iv; q.z = keep[i][0].z * iv; q.w = keep[i][0].w * iv;
            st4_nt(prow + 4 * s, q);
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// k_e_step_packed with one side of its memory traffic removed:  MODE 1: the stores wrap inside a 1 MB window of P
// (they stay in the L2s: full store instruction stream, almost no write traffic to the fabric);  MODE 2: every gather
// reads row 0 (L1 hits: full load instruction stream, no read traffic), stores as shipped
template <int C, int LPN, int MODE>
__global__ __launch_bounds__(256) void k_e_step_packed_mode(const int *__restrict__ rowidx, const int *__restrict__ colidx,
                                                            i64 nnz, const float *__restrict__ U,
                                                            const float *__restrict__ Vt, float *__restrict__ P, float thresh) {
    constexpr int KP = 4 * C;
    __shared__ float part[4][64 * C];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float *mypart = part[wave];
    const i64 tiles = (nnz + 63) >> 6;
    for (i64 t = (i64)blockIdx.x * 4 + wave; t < tiles; t += (i64)gridDim.x * 4) {
        const i64 mine = (t << 6) + lane;
        int d_l = mine < nnz ? __builtin_nontemporal_load(rowidx + mine) : 0;
        int w_l = mine < nnz ? __builtin_nontemporal_load(colidx + mine) : 0;
        if (MODE == 2) { d_l &= 1; w_l &= 1; }
        float4 u[C][1], vt[C][1], keep[C][1];
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const unsigned s = (unsigned)lane + 64u * i;
            const unsigned e = s / (unsigned)C, c = s - e * (unsigned)C;
            u[i][0] = ld4(U + (i64)__shfl(d_l, (int)e, 64) * KP + 4 * c);
            vt[i][0] = ld4(Vt + (i64)__shfl(w_l, (int)e, 64) * KP + 4 * c);
        }
#pragma unroll
        for (int i = 0; i < C; ++i) {
            float unth;
            mypart[lane + 64 * i] = products<1, false>(u[i], vt[i], thresh, keep[i], unth);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        float p[C];
#pragma unroll
        for (int j = 0; j < C; ++j) p[j] = mypart[lane * C + j];
        const float inv = inv_norm(tree_sum<C, 0, LPN>(p));
        float *prow = P + (MODE == 1 ? (t & 127) << 6 : t << 6) * KP;    // 128 tiles x 5 KB = 640 KB window
#pragma unroll
        for (int i = 0; i < C; ++i) {
            const unsigned s = (unsigned)lane + 64u * i;
            const float iv = __shfl(inv, (int)(s / (unsigned)C), 64);
            float4 q;
            q.x = keep[i][0].x * iv; q.y = keep[i][0].y * iv; q.z = keep[i][0].z * iv; q.w = keep[i][0].w * iv;
            if (MODE == 1) st4(prow + 4 * s, q); else st4_nt(prow + 4 * s, q);
        }
        __builtin_amdgcn_wave_barrier();
    }
}
}  // namespace probe

int main(int argc, char **argv) {
    const int k = argc > 1 ? atoi(argv[1]) : 20;
    const int n = argc > 2 ? atoi(argv[2]) : 18846;
    const int m = argc > 3 ? atoi(argv[3]) : 173762;
    const int per_doc = argc > 4 ? atoi(argv[4]) : 156;
    const int mult = argc > 5 ? atoi(argv[5]) : 8;
    const int kp = (k + 3) / 4 * 4;
    if (kp != 20) { fprintf(stderr, "this probe instantiates k = 17..20 only\n"); return 1; }
    std::mt19937_64 rng(1);
    std::vector<int> rowidx, colidx;
    std::vector<double> cdf(m);
    double z = 0;
    for (int i = 0; i < m; ++i) { z += 1.0 / std::pow(i + 1.0, 1.07); cdf[i] = z; }
    std::vector<int> perm(m);
    for (int i = 0; i < m; ++i) perm[i] = i;
    std::shuffle(perm.begin(), perm.end(), rng);
    std::uniform_real_distribution<double> un(0.0, 1.0);
    for (int d = 0; d < n; ++d) {
        std::vector<int> ws;
        for (int j = 0; j < per_doc; ++j) {
            const int r = (int)(std::lower_bound(cdf.begin(), cdf.end(), un(rng) * z) - cdf.begin());
            ws.push_back(perm[std::min(r, m - 1)]);
        }
        std::sort(ws.begin(), ws.end());
        ws.erase(std::unique(ws.begin(), ws.end()), ws.end());
        for (int w : ws) { rowidx.push_back(d); colidx.push_back(w); }
    }
    const i64 nnz = (i64)rowidx.size();
    std::vector<float> U((size_t)n * kp), Vt((size_t)m * kp);
    for (auto &x : U) x = (float)un(rng) / k;
    for (auto &x : Vt) x = (float)un(rng) / m;
    int *d_row, *d_col;
    float *d_U, *d_Vt, *d_P, *d_Q, *d_sink;
    HC(hipMalloc(&d_row, nnz * 4 + 256)); HC(hipMalloc(&d_col, nnz * 4 + 256));
    HC(hipMalloc(&d_U, U.size() * 4)); HC(hipMalloc(&d_Vt, Vt.size() * 4));
    const size_t pbytes = (size_t)(nnz + 64) * kp * 4;
    HC(hipMalloc(&d_P, pbytes)); HC(hipMalloc(&d_Q, pbytes)); HC(hipMalloc(&d_sink, 4096 * 256 * 4));
    HC(hipMemcpy(d_row, rowidx.data(), nnz * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(d_col, colidx.data(), nnz * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(d_U, U.data(), U.size() * 4, hipMemcpyHostToDevice));
    HC(hipMemcpy(d_Vt, Vt.data(), Vt.size() * 4, hipMemcpyHostToDevice));
    hipDeviceProp_t prop;
    HC(hipGetDeviceProperties(&prop, 0));
    const i64 tiles = (nnz + 63) / 64;
    const int grid = (int)std::min<i64>((tiles + 3) / 4, (i64)prop.multiProcessorCount * mult);
    const float thresh = 1e-32f;
    const double gb = (4.0 * (n + 1) + 4.0 * nnz + 4.0 * k * nnz + 4.0 * k * (n + m)) / 1e9;
    printf("{\"nnz\": %lld, \"k\": %d, \"grid\": %d, \"algorithmic_GB\": %.4f}\n", (long long)nnz, k, grid, gb);
    auto report = [&](const char *name, double us, bool full) {
        printf("{\"variant\": \"%s\", \"us\": %.2f, \"entries_per_ns\": %.1f%s", name, us, nnz / us / 1e3, full ? "" : "}\n");
        if (full) printf(", \"frac_of_8TBs\": %.3f}\n", gb / (us * 1e-6) / 8000.0);
    };
    using S = plsa::Shape<8, 1, false>;
    HC(hipMemset(d_P, 0, pbytes));
    double us = time_us([&] { hipLaunchKernelGGL((plsa::k_e_step<S, false>), dim3(grid), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_P, kp, thresh); });
    report("k_e_step", us, true);
    auto check = [&](const char *name) {
        std::vector<float> a((size_t)nnz * kp), b((size_t)nnz * kp);
        HC(hipMemcpy(a.data(), d_P, a.size() * 4, hipMemcpyDeviceToHost));
        HC(hipMemcpy(b.data(), d_Q, b.size() * 4, hipMemcpyDeviceToHost));
        printf("{\"check\": \"%s\", \"bit_identical\": %s}\n", name, memcmp(a.data(), b.data(), a.size() * 4) == 0 ? "true" : "false");
    };
    HC(hipMemset(d_Q, 0, pbytes));
    us = time_us([&] { hipLaunchKernelGGL((plsa::k_e_step_packed<5, 8>), dim3(grid), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_Q, thresh); });
    report("k_e_step_packed", us, true); check("k_e_step_packed");
    HC(hipMemset(d_Q, 0, pbytes));
    us = time_us([&] { hipLaunchKernelGGL((probe::k_e_step_packed_pf2<5, 8>), dim3(grid), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_Q, thresh); });
    report("k_e_step_packed_pf2", us, true); check("k_e_step_packed_pf2");
    for (int g2 : {grid / 2, grid, grid * 2}) {
        us = time_us([&] { hipLaunchKernelGGL((probe::k_e_step_packed_pf2<5, 8>), dim3(g2), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_Q, thresh); });
        char nm[64]; snprintf(nm, sizeof nm, "k_e_step_packed_pf2 grid %d", g2);
        report(nm, us, true);
    }
    us = time_us([&] { hipLaunchKernelGGL((probe::k_e_step_packed_mode<5, 8, 0>), dim3(grid), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_Q, thresh); });
    report("packed, no id prefetch", us, true);
    us = time_us([&] { hipLaunchKernelGGL((probe::k_e_step_packed_mode<5, 8, 1>), dim3(grid), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_Q, thresh); });
    report("packed, stores wrapped into a 640 KB window (no write traffic)", us, false);
    us = time_us([&] { hipLaunchKernelGGL((probe::k_e_step_packed_mode<5, 8, 2>), dim3(grid), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_Q, thresh); });
    report("packed, every gather reads rows 0 / 1 (no read traffic)", us, false);
    us = time_us([&] { hipLaunchKernelGGL((probe::k_store_only<5>), dim3(grid), dim3(256), 0, 0, nnz, d_Q); });
    report("store_only", us, false);
    us = time_us([&] { hipLaunchKernelGGL((probe::k_load_only<5, 8>), dim3(grid), dim3(256), 0, 0, d_row, d_col, nnz, d_U, d_Vt, d_sink, thresh); });
    report("load_only", us, false);
    us = time_us([&] { hipLaunchKernelGGL((probe::k_store_only<5>), dim3(1), dim3(256), 0, 0, (i64)64, d_Q); });
    report("empty_launch", us, false);
    return 0;
}
