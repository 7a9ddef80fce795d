// Counter-based Pauli-error sampler shared by gf2_mc.hip and gf2_sparse.hip (DESIGN.md "Sampler").
#pragma once

#include "gf2_internal.h"

#define GF2_GOLDEN 0x9E3779B97F4A7C15ull
#define GF2_STREAM_MULT 0xD1B54A32D192ED03ull

__host__ __device__ static inline u64 mix64(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Sampler (DESIGN.md "Sampler"), per 64-qubit word w of sample i:
//   ks = mix64(seed + G (i + 1)),  d = mix64(ks + M (w + 1))                        one draw per word
//   K  = #{k < nb : (d >> 32) >= cdf[k]} erroneous qubits (inverse binomial CDF, an integer table made on the host)
//   for k = 0 .. K - 1:  v = mix64(d + G (k + 1));  the high half of v picks a position by Floyd's algorithm (K distinct
//   positions out of nb), the low half c its kind: c < t_1 -> X, t_1 <= c < t_2 -> Y, else Z.
// A word without errors costs one mix64, each erroneous qubit one more.
struct SamplerTables {
    u64 t_1, t_2;              // thresholds in [0, 2^32]: X only below t_1, Y below t_2
    u64 cdf_full[65];          // K = #{k < 64 : u >= cdf_full[k]} for whole words
    u64 cdf_last[65];          // the same for the last word of nb_last valid qubits
    int nb_last;
};

// The draw of word w that fixes its number of errors and seeds its further draws; ks = sample_key(seed, sample).
__host__ __device__ static inline u64 sample_key(u64 seed, u64 sample) { return mix64(seed + GF2_GOLDEN * (sample + 1)); }
__host__ __device__ static inline u64 word_draw(u64 ks, u64 w) { return mix64(ks + GF2_STREAM_MULT * (w + 1)); }

// Errors of a word with `count` erroneous qubits out of nb (Floyd's sampling, one mix64 per qubit).
__device__ static inline void place_errors(u64 d, int count, int nb, u64 t_1, u64 t_2, u64* ex, u64* ez) {
    u64 chosen = 0, has_x = 0, has_z = 0;
    for (int k = 0; k < count; ++k) {
        const u64 v = mix64(d + GF2_GOLDEN * (u64)(k + 1));
        const int j = nb - count + k;
        const int t = (int)(((v >> 32) * (u64)(j + 1)) >> 32);
        const int pos = ((chosen >> t) & 1ull) ? j : t;
        const u64 c = v & 0xFFFFFFFFull;
        chosen |= 1ull << pos;
        if (c < t_2) has_x |= 1ull << pos;
        if (c >= t_1) has_z |= 1ull << pos;
    }
    *ex = has_x;
    *ez = has_z;
}

// cdf: the table for this word (in LDS); nb: valid qubits of this word (1..64).
__device__ static inline int error_count(u64 d, int nb, const u64* cdf) {
    const u64 u = d >> 32;
    int count = 0;
    while (count < nb && u >= cdf[count]) count += 1;
    return count;
}

__device__ static inline void sample_word(u64 seed, u64 sample, u64 w, int nb, const u64* cdf, u64 t_1, u64 t_2,
                                          u64* ex, u64* ez) {
    const u64 d = word_draw(sample_key(seed, sample), w);
    place_errors(d, error_count(d, nb, cdf), nb, t_1, t_2, ex, ez);
}

// Copies the two CDF tables of the kernel argument into LDS (per-lane table indices need addressable memory).
__device__ __forceinline__ void stage_cdf(const SamplerTables& tb, u64* cdf_lds) {
    for (int i = threadIdx.x; i < 130; i += blockDim.x) cdf_lds[i] = i < 65 ? tb.cdf_full[i] : tb.cdf_last[i - 65];
    __syncthreads();
}

// Inverse binomial CDF as integers: cdf[k] = round(2^32 * P(Bin(nb, q) <= k)) (to nearest, clamped to 2^32), q = T / 2^32, in
// IEEE doubles with this exact operation order (oracle/gf2_oracle.c and oracle/cpu_ref.py repeat it).
// K = #{k < nb : u >= cdf[k]}.  Rounding to nearest matters in the tail: a sum that ends one ulp short of 1.0 must still give
// 2^32 (never reached by u <= 2^32 - 1), not 2^32 - 1, or u = 2^32 - 1 would make all nb qubits of the word err.
static inline void binomial_cdf_table(uint64_t t_any, int nb, u64* cdf) {
    for (int k = 0; k < 65; ++k) cdf[k] = 4294967296ull;
    if (nb <= 0) return;
    if (t_any >= 4294967296ull) {
        for (int k = 0; k < nb; ++k) cdf[k] = 0;               // every qubit errs
        return;
    }
    const double q = (double)t_any / 4294967296.0, om = 1.0 - q;
    double pmf = 1.0;
    for (int i = 0; i < nb; ++i) pmf *= om;
    double cum = 0.0;
    for (int k = 0; k < nb; ++k) {
        cum += pmf;
        double c = __builtin_floor(cum * 4294967296.0 + 0.5);
        if (c > 4294967296.0) c = 4294967296.0;
        cdf[k] = (u64)c;
        pmf = pmf * (double)(nb - k) / (double)(k + 1) * q / om;
    }
}

static inline int make_thresholds(double p_x, double p_y, double p_z, int64_t n, SamplerTables* th) {
    if (!(p_x >= 0.0) || !(p_y >= 0.0) || !(p_z >= 0.0) || p_x + p_y + p_z > 1.0 + 1e-12)
        GF2_FAIL(GF2_E_ARG, "probabilities must be non-negative and sum to at most 1");
    const double p_t = p_x + p_y + p_z, p_xy = p_x + p_y;
    const uint64_t t_any = gf2_quantise(p_t);
    th->t_1 = p_t > 0.0 ? gf2_quantise(p_x / p_t) : 0;
    th->t_2 = p_t > 0.0 ? gf2_quantise(p_xy / p_t) : 0;
    th->nb_last = n > 0 ? (int)(n - ((n - 1) / 64) * 64) : 0;
    binomial_cdf_table(t_any, 64, th->cdf_full);
    binomial_cdf_table(t_any, th->nb_last, th->cdf_last);
    return GF2_OK;
}

