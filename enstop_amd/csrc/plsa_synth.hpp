// plsa_synth.hpp -- synthetic bag-of-words corpus generated directly in HBM (bench.py and the
// large-size tests; the reference has no generator -- its only corpus is 20-Newsgroups, which is
// not available offline).  Model: document d holds T_d tokens, T_d ~ lognormal(sigma = 0.6); each
// token is an independent Zipf(s) draw over a fixed pseudo-random permutation of the vocabulary;
// the CSR stores the multiplicity of every distinct (doc, word) pair as a float32 count.  The
// matrix is a pure function of (n, m, tokens-per-doc mean, s, seed): counter-based hashing, no
// generator state, identical on every device.
//
// Round 5 -- TOPICAL variant (k_synth_draw_topics): the corpus above draws every token independently, so no two
// words co-occur more often than chance and nothing about it resembles text (20-Newsgroups, the reference's
// only corpus, notebooks/EnsTop with 20-Newsgroups.ipynb:49).  Here document d first draws a topic mixture
// theta_d ~ Dirichlet(alpha) over k0 latent topics (gamma variates by Marsaglia-Tsang from the counter-based
// hash), every topic owns its OWN Zipf(s) ranking of the vocabulary (an affine permutation per topic), and a
// token is: with probability `background` a draw from the shared ranking (function words), otherwise topic
// t ~ theta_d and a word from topic t's ranking -- the generative model pLSA itself assumes.
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

// one gamma(shape, 1) variate, shape >= 1 (Marsaglia & Tsang 2000), from the hash stream (key, 0), (key, 1), ...
__device__ __forceinline__ double gamma_mt(double shape, uint64_t key) {
    const double dd = shape - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * dd);
    for (uint64_t j = 0; j < 64; ++j) {
        const uint64_t h1 = mix64(key ^ mix64(3 * j + 1)), h2 = mix64(key ^ mix64(3 * j + 2)),
                       h3 = mix64(key ^ mix64(3 * j + 3));
        const double x = sqrt(-2.0 * log(u01(h1))) * cos(6.283185307179586 * u01(h2));
        double v = 1.0 + cc * x;
        if (v <= 0.0) continue;
        v = v * v * v;
        if (log(u01(h3)) < 0.5 * x * x + dd - dd * v + dd * log(v)) return dd * v;
    }
    return dd;      // (never reached in practice: acceptance > 95 % per round)
}

// un-normalised weight of latent topic t in document d: gamma(alpha + 1) * u^(1/alpha) (= a gamma(alpha) variate)
__device__ __forceinline__ double doc_topic_weight(uint64_t seed, long long d, int t, double alpha) {
    const uint64_t key = mix64(seed ^ 0x7091C5ull ^ mix64((uint64_t)d * 1024ull + (uint64_t)t));
    return gamma_mt(alpha + 1.0, key) * pow(u01(mix64(key ^ 0xA5A5A5A5ull)), 1.0 / alpha);
}

// ground truth of the topical corpus: the latent topic with the largest share of each document's mixture
__global__ void k_synth_dominant_topic(int n, int k0, double alpha, uint64_t seed, int *__restrict__ out) {
    const long long d = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (d >= n) return;
    double best = -1.0;
    int arg = 0;
    for (int t = 0; t < k0; ++t) {
        const double g = doc_topic_weight(seed, d, t, alpha);
        if (g > best) { best = g; arg = t; }
    }
    out[d] = arg;
}

// one wave per document, four documents per workgroup.  Topic mixture: g_t = gamma(alpha + 1) * u^(1/alpha)
// (= gamma(alpha) for any alpha > 0), normalised -> cumulative theta in LDS; tokens as described in the header.
// perm[2*t], perm[2*t+1]: the affine ranking (a * rank + b) mod m of topic t; t == k0 is the shared ranking.
__global__ void k_synth_draw_topics(int n, int m, const long long *__restrict__ tok_ptr,
                                    const double *__restrict__ cdf, const uint64_t *__restrict__ perm, int k0,
                                    double alpha, double background, uint64_t seed,
                                    unsigned long long *__restrict__ keys) {
    __shared__ double tcdf[4][256];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    for (long long base = (long long)blockIdx.x * 4; base < n; base += (long long)gridDim.x * 4) {
        const long long d = base + w;
        if (d < n) {
            for (int t = lane; t < k0; t += 64) tcdf[w][t] = doc_topic_weight(seed, d, t, alpha);
        }
        __syncthreads();
        if (d < n && lane == 0) {
            double tot = 0.0;
            for (int t = 0; t < k0; ++t) { tot += tcdf[w][t]; tcdf[w][t] = tot; }
            // a document whose every gamma underflowed (alpha tiny) falls back to one hashed topic
            if (!(tot > 0.0)) {
                const int t1 = (int)(mix64(seed ^ (uint64_t)d) % (uint64_t)k0);
                for (int t = 0; t < k0; ++t) tcdf[w][t] = t >= t1 ? 1.0 : 0.0;
                tot = 1.0;
            }
            const double inv = 1.0 / tot;
            for (int t = 0; t < k0; ++t) tcdf[w][t] *= inv;
            tcdf[w][k0 - 1] = 1.0;
        }
        __syncthreads();
        if (d < n) {
            const long long t0 = tok_ptr[d], t1 = tok_ptr[d + 1];
            for (long long t = t0 + lane; t < t1; t += 64) {
                const uint64_t ht = mix64((uint64_t)t * 0x9E3779B97F4A7C15ull + 0x5851F42D4C957F2Dull);
                const double u = u01(mix64(seed ^ ht));
                const double ub = u01(mix64(seed ^ ht ^ 0xB5297A4D3F84D5B5ull));
                int topic = k0;                                   // shared ranking
                if (ub >= background) {
                    const double ut = u01(mix64(seed ^ ht ^ 0x68E31DA4C2B2AE35ull));
                    int lo = 0, hi = k0 - 1;                      // first topic with tcdf >= ut
                    while (lo < hi) {
                        const int mid = (lo + hi) >> 1;
                        if (tcdf[w][mid] < ut) lo = mid + 1; else hi = mid;
                    }
                    topic = lo;
                }
                int lo = 0, hi = m - 1;                           // first rank with cdf[rank] >= u
                while (lo < hi) {
                    const int mid = (lo + hi) >> 1;
                    if (cdf[mid] < u) lo = mid + 1; else hi = mid;
                }
                const uint64_t word = (perm[2 * topic] * (uint64_t)lo + perm[2 * topic + 1]) % (uint64_t)m;
                keys[t] = ((unsigned long long)d << 32) | (unsigned long long)word;
            }
        }
        __syncthreads();                                           // tcdf is rewritten by the next four documents
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
