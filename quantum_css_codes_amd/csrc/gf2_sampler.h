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

// Sampler (DESIGN.md "Sampler"), per SEGMENT s of 512 qubits of sample i (nb valid qubits: 512, or what is left of n in the
// last segment; a code of at most 512 qubits is one segment):
//   ks = mix64(seed + G (i + 1)),  d = mix64(ks + M (s + 1))                        one draw per segment
//   K  = #{k < nb : (d >> 32) >= cdf[k]} erroneous qubits (inverse binomial CDF, an integer table made on the host)
//   for k = 0 .. K - 1:  v = mix64(d + G (k + 1));  the high half of v picks a position inside the segment by Floyd's algorithm
//   (j = nb - K + k, t = ((v >> 32) (j + 1)) >> 32, taken already -> j, else t), the low half c its kind: c < t_1 -> X,
//   t_1 <= c < t_2 -> Y, else Z.  Qubit = 512 s + position.
// A segment costs one mix64 and each erroneous qubit one more (at most 64 qubits: exactly the definition this replaced, which
// drew per 64-qubit word -- 64 draws per sample of n = 4096 before its first error instead of 8).
#define GF2_SEG_BITS 512
#define GF2_SEG_WORDS 8
#define GF2_SEG_CDF (GF2_SEG_BITS + 1)

struct SamplerTables {         // small codes (n <= 64): one segment, tables inside the kernel argument
    u64 t_1, t_2;              // thresholds in [0, 2^32]: X only below t_1, Y below t_2
    u64 cdf_full[65];          // unused since the segments (kept: the small-code kernels index cdf_last at 65)
    u64 cdf_last[65];          // K = #{k < n : u >= cdf_last[k]}
    int nb_last;
};

struct SegTables {             // any n: tables in device memory (the context's, gf2_seg_tables)
    u64 t_1, t_2;
    const u64* cdf;            // [2][GF2_SEG_CDF]: whole segments, the last segment
    int nb_last, nseg;
};

__host__ __device__ static inline u64 sample_key(u64 seed, u64 sample) { return mix64(seed + GF2_GOLDEN * (sample + 1)); }
__host__ __device__ static inline u64 word_draw(u64 ks, u64 w) { return mix64(ks + GF2_STREAM_MULT * (w + 1)); }
__host__ __device__ static inline u64 segment_draw(u64 ks, u64 s) { return mix64(ks + GF2_STREAM_MULT * (s + 1)); }

// Errors of a one-word segment (n <= 64) with `count` erroneous qubits out of nb (Floyd's sampling, one mix64 per qubit).
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

// cdf: the table for this segment; nb: its valid qubits.
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

// The k-th erroneous qubit of a segment with draw d and K errors out of nb: Floyd's candidate t (the caller resolves a collision
// to j = nb - K + k) and the kind (bit 0: X component, bit 1: Z component).  Independent of the other qubits of the segment.
__device__ __forceinline__ void error_draw(u64 d, int k, int K, int nb, u64 t_1, u64 t_2, unsigned int* t_out, unsigned int* kind) {
    const u64 v = mix64(d + GF2_GOLDEN * (u64)(k + 1));
    const unsigned int j = (unsigned int)(nb - K + k);
    *t_out = (unsigned int)(((v >> 32) * (u64)(j + 1u)) >> 32);
    const u64 c = v & 0xFFFFFFFFull;
    *kind = (c < t_2 ? 1u : 0u) | (c >= t_1 ? 2u : 0u);
}

// One segment, start to end, by one lane: xz = 32 dwords of this lane's own (x: 0..15, z: 16..31), zeroed here.
__device__ static inline void sample_segment(u64 ks, int s, int nb, const u64* cdf, u64 t_1, u64 t_2, unsigned int* xz) {
    const int used = ((nb + 63) >> 6) * 2;                             // dwords per component of the words this segment has (the callers read no more)
    for (int i = 0; i < used; ++i) xz[i] = 0, xz[16 + i] = 0;
    const u64 d = segment_draw(ks, (u64)s);
    const int K = error_count(d, nb, cdf);
    for (int k = 0; k < K; ++k) {
        unsigned int t, kind;
        error_draw(d, k, K, nb, t_1, t_2, &t, &kind);
        const unsigned int j = (unsigned int)(nb - K + k);
        const bool taken = ((xz[t >> 5] | xz[16 + (t >> 5)]) >> (t & 31u)) & 1u;
        const unsigned int pos = taken ? j : t;
        if (kind & 1u) xz[pos >> 5] |= 1u << (pos & 31u);
        if (kind & 2u) xz[16 + (pos >> 5)] |= 1u << (pos & 31u);
    }
}

// Copies the small-code tables of the kernel argument into LDS (per-lane table indices need addressable memory).
__device__ __forceinline__ void stage_cdf(const SamplerTables& tb, u64* cdf_lds) {
    for (int i = threadIdx.x; i < 130; i += blockDim.x) cdf_lds[i] = i < 65 ? tb.cdf_full[i] : tb.cdf_last[i - 65];
    __syncthreads();
}
// ... and the segment tables from device memory: cdf_lds[2 * GF2_SEG_CDF]
__device__ __forceinline__ void stage_seg_cdf(const SegTables& tb, u64* cdf_lds) {
    for (int i = threadIdx.x; i < 2 * GF2_SEG_CDF; i += blockDim.x) cdf_lds[i] = tb.cdf[i];
    __syncthreads();
}

// Inverse binomial CDF as integers: cdf[k] = round(2^32 * P(Bin(nb, q) <= k)) (to nearest, clamped to 2^32), q = T / 2^32, in
// IEEE doubles with this exact operation order (oracle/gf2_oracle.c and oracle/cpu_ref.py repeat it).
// K = #{k < nb : u >= cdf[k]}.  Rounding to nearest matters in the tail: a sum that ends one ulp short of 1.0 must still give
// 2^32 (never reached by u <= 2^32 - 1), not 2^32 - 1, or u = 2^32 - 1 would make all nb qubits of the segment err.
// `cdf` has cap entries; those from nb on are 2^32.
static inline void binomial_cdf_direct(uint64_t t_any, int nb, u64* cdf, int cap) {
    for (int k = 0; k < cap; ++k) cdf[k] = 4294967296ull;
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
// Segments of more than 64 qubits at q > 1/2: (1 - q)^nb underflows long before q reaches 1, so the table comes from the
// complementary count Y = nb - K ~ Bin(nb, 1 - q):  P(K <= k) = 1 - P(Y <= nb - k - 1).
static inline void binomial_cdf_table(uint64_t t_any, int nb, u64* cdf, int cap) {
    if (nb <= 64 || t_any <= 2147483648ull || t_any >= 4294967296ull) {
        binomial_cdf_direct(t_any, nb, cdf, cap);
        return;
    }
    u64 other[GF2_SEG_CDF];
    binomial_cdf_direct(4294967296ull - t_any, nb, other, GF2_SEG_CDF);
    for (int k = 0; k < cap; ++k) cdf[k] = 4294967296ull;
    for (int k = 0; k < nb; ++k) cdf[k] = 4294967296ull - other[nb - k - 1];
}

static inline int check_probabilities(double p_x, double p_y, double p_z) {
    if (!(p_x >= 0.0) || !(p_y >= 0.0) || !(p_z >= 0.0) || p_x + p_y + p_z > 1.0 + 1e-12)
        GF2_FAIL(GF2_E_ARG, "probabilities must be non-negative and sum to at most 1");
    return GF2_OK;
}

// small codes (n <= 64)
static inline int make_thresholds(double p_x, double p_y, double p_z, int64_t n, SamplerTables* th) {
    GF2_TRY(check_probabilities(p_x, p_y, p_z));
    const double p_t = p_x + p_y + p_z, p_xy = p_x + p_y;
    const uint64_t t_any = gf2_quantise(p_t);
    th->t_1 = p_t > 0.0 ? gf2_quantise(p_x / p_t) : 0;
    th->t_2 = p_t > 0.0 ? gf2_quantise(p_xy / p_t) : 0;
    th->nb_last = n > 0 ? (int)(n < 64 ? n : 64) : 0;
    binomial_cdf_table(t_any, 64, th->cdf_full, 65);
    binomial_cdf_table(t_any, th->nb_last, th->cdf_last, 65);
    return GF2_OK;
}

// Any n: thresholds, and the two segment tables in the context's device buffer (uploaded when the rates or n change).
int gf2_seg_tables(gf2_ctx* ctx, double p_x, double p_y, double p_z, int64_t n, SegTables* out);
