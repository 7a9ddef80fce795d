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

// Sampler (DESIGN.md "Sampler"): per 64-qubit word of a sample, three independent streams of 32-bit uniforms
// (high then low half of successive splitmix64 draws).
//   stream 0: the first uniform picks the number K of erroneous qubits in the word by inverse binomial CDF (an integer
//             table made on the host); K distinct positions follow by Floyd's algorithm, one uniform each.
//   stream 1: one uniform per erroneous qubit, ascending: the error has an X component iff uniform < t_x.
//   stream 2: one uniform per qubit with an X component, ascending: it is a Y iff uniform < t_y.
struct SamplerTables {
    u64 t_x, t_y;              // thresholds in [0, 2^32]
    u64 cdf_full[65];          // K = #{k < 64 : u >= cdf_full[k]} for whole words
    u64 cdf_last[65];          // the same for the last word of nb_last valid qubits
    int nb_last;
};

struct UniformStream {
    u64 base, cur;
    unsigned int k;
    bool half;
    __device__ __forceinline__ explicit UniformStream(u64 b) : base(b), cur(0), k(0), half(false) {}
    __device__ __forceinline__ u64 next() {                   // 32-bit uniform: high half, then low half of each draw
        if (!half) {
            cur = mix64(base + GF2_GOLDEN * (u64)(k + 1));
            k += 1;
            half = true;
            return cur >> 32;
        }
        half = false;
        return cur & 0xFFFFFFFFull;
    }
};

// cdf: the table for this word (in LDS); nb: valid qubits of this word (1..64).
__device__ static inline void sample_word(u64 seed, u64 sample, u64 w, int nb, const u64* cdf, u64 t_x, u64 t_y,
                                          u64* ex, u64* ez) {
    const u64 ks = mix64(seed + GF2_GOLDEN * (sample + 1));
    UniformStream s0(mix64(ks ^ (GF2_STREAM_MULT * (4 * w + 1))));
    const u64 u = s0.next();
    int count = 0;
    while (count < nb && u >= cdf[count]) count += 1;
    u64 any_err = 0;
    for (int idx = 0; idx < count; ++idx) {                    // Floyd: `count` distinct positions out of nb
        const int i = nb - count + idx;
        const int t = (int)((s0.next() * (u64)(i + 1)) >> 32);
        any_err |= 1ull << (((any_err >> t) & 1ull) ? i : t);
    }
    u64 has_x = 0, is_y = 0;
    if (any_err) {
        UniformStream s1(mix64(ks ^ (GF2_STREAM_MULT * (4 * w + 2))));
        for (u64 x = any_err; x; x &= x - 1)
            if (s1.next() < t_x) has_x |= x & (0ull - x);
    }
    if (has_x) {
        UniformStream s2(mix64(ks ^ (GF2_STREAM_MULT * (4 * w + 3))));
        for (u64 x = has_x; x; x &= x - 1)
            if (s2.next() < t_y) is_y |= x & (0ull - x);
    }
    *ex = has_x;
    *ez = (any_err & ~has_x) | is_y;
}

// Copies the two CDF tables of the kernel argument into LDS (per-lane table indices need addressable memory).
__device__ __forceinline__ void stage_cdf(const SamplerTables& tb, u64* cdf_lds) {
    for (int i = threadIdx.x; i < 130; i += blockDim.x) cdf_lds[i] = i < 65 ? tb.cdf_full[i] : tb.cdf_last[i - 65];
    __syncthreads();
}

// Inverse binomial CDF as integers: cdf[k] = floor(2^32 * P(Bin(nb, q) <= k)), q = T / 2^32, in IEEE doubles with
// this exact operation order (oracle/gf2_oracle.c and oracle/cpu_ref.py repeat it).  K = #{k < nb : u >= cdf[k]}.
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
        double c = __builtin_floor(cum * 4294967296.0);
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
    th->t_x = p_t > 0.0 ? gf2_quantise(p_xy / p_t) : 0;
    th->t_y = p_xy > 0.0 ? gf2_quantise(p_y / p_xy) : 0;
    th->nb_last = n > 0 ? (int)(n - ((n - 1) / 64) * 64) : 0;
    binomial_cdf_table(t_any, 64, th->cdf_full);
    binomial_cdf_table(t_any, th->nb_last, th->cdf_last);
    return GF2_OK;
}

