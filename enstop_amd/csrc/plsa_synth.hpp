// plsa_synth.hpp -- synthetic bag-of-words corpus generated directly in HBM (bench.py and the
// large-size tests; the reference has no generator -- its only corpus is 20-Newsgroups, which is
// not available offline).  Model: document d holds T_d tokens, T_d ~ lognormal(sigma = 0.6); each
// token is an independent Zipf(s) draw over a fixed pseudo-random permutation of the vocabulary;
// the CSR stores the multiplicity of every distinct (doc, word) pair as a float32 count.  The
// matrix is a pure function of (n, m, tokens-per-doc mean, s, seed): counter-based hashing, no
// generator state, identical on every device.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace plsa {

__host__ __device__ __forceinline__ uint64_t mix64(uint64_t x) {  // splitmix64 finaliser
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
__device__ __forceinline__ double u01(uint64_t h) { return ((h >> 11) + 0.5) * (1.0 / 9007199254740992.0); }

// tokens per document: clamp(round(exp(mu + sigma * N(0,1))), 1, cap)
__global__ void k_synth_doc_tokens(int n, double mu, double sigma, int cap, uint64_t seed,
                                   int *__restrict__ tokens) {
    const long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (d > n) return;
    if (d == n) { tokens[d] = 0; return; }
    const uint64_t h1 = mix64(seed ^ mix64(2 * (uint64_t)d + 1));
    const uint64_t h2 = mix64(seed ^ mix64(2 * (uint64_t)d + 2) ^ 0xD1B54A32D192ED03ull);
    const double z = sqrt(-2.0 * log(u01(h1))) * cos(6.283185307179586 * u01(h2));
    double t = rint(exp(mu + sigma * z));
    if (t < 1.0) t = 1.0;
    if (t > (double)cap) t = (double)cap;
    tokens[d] = (int)t;
}

// one wave per document: Zipf draws by inverse CDF (binary search), word id through the affine
// permutation (a * rank + b) mod m, key = doc << 32 | word
__global__ void k_synth_draw(int n, int m, const long long *__restrict__ tok_ptr,
                             const double *__restrict__ cdf, uint64_t perm_a, uint64_t perm_b,
                             uint64_t seed, unsigned long long *__restrict__ keys) {
    const int lane = threadIdx.x & 63;
    const long long wid = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const long long nw = ((long long)gridDim.x * blockDim.x) >> 6;
    for (long long d = wid; d < n; d += nw) {
        const long long t0 = tok_ptr[d], t1 = tok_ptr[d + 1];
        for (long long t = t0 + lane; t < t1; t += 64) {
            const double u = u01(mix64(seed ^ mix64((uint64_t)t * 0x9E3779B97F4A7C15ull + 0x5851F42D4C957F2Dull)));
            int lo = 0, hi = m - 1;                     // first rank with cdf[rank] >= u
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (cdf[mid] < u) lo = mid + 1; else hi = mid;
            }
            const uint64_t word = (perm_a * (uint64_t)lo + perm_b) % (uint64_t)m;
            keys[t] = ((unsigned long long)d << 32) | (unsigned long long)word;
        }
    }
}

__global__ void k_synth_heads(const unsigned long long *__restrict__ keys, long long T,
                              int *__restrict__ flag) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < T;
         j += (long long)gridDim.x * blockDim.x)
        flag[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1 : 0;
}

__global__ void k_synth_emit(const unsigned long long *__restrict__ keys, long long T,
                             const int *__restrict__ flag, const int *__restrict__ pos, int n,
                             int nnz, int *__restrict__ indptr, int *__restrict__ col,
                             float *__restrict__ val) {
    for (long long j = (long long)blockIdx.x * blockDim.x + threadIdx.x; j < T;
         j += (long long)gridDim.x * blockDim.x) {
        if (j == 0) indptr[n] = nnz;
        if (!flag[j]) continue;
        const unsigned long long key = keys[j];
        const int p = pos[j];
        int c = 1;
        while (j + c < T && keys[j + c] == key) ++c;
        col[p] = (int)(key & 0xFFFFFFFFull);
        val[p] = (float)c;
        const unsigned long long d = key >> 32;
        if (j == 0 || (keys[j - 1] >> 32) != d) indptr[d] = p;
    }
}

}  // namespace plsa
