// Counter-based Pauli-error sampler and the Monte-Carlo pipeline (sample -> syndromes -> histograms).
//
// Build-defined: the reference has no sampler; its only Monte-Carlo is the QVM run of
// test/test_fidelity.py.  The generator is specified in DESIGN.md ("Sampler") and restated independently
// in oracle/cpu_ref.py (plain evaluation) and oracle/gf2_oracle.c; this file is the lazy evaluation of
// the same definition.  Sample i is a pure function of (seed, i), so any sharding of the index range over
// GPUs gives the same histograms.
#include <stdlib.h>
#include <string.h>

#include <time.h>

#include "gf2_internal.h"

#include "gf2_sampler.h"

// Generic sampler, any n and either layout: one lane per (sample, segment).  The lane runs its segment start to end into 32 dwords
// of LDS of its own and writes the segment's (up to) eight words of e_x and e_z.  Sample-major: word w of sample i at i * lde + w.
// Tiled: at (i>>6)*64*ldt + (w>>1)*128 + (i&63)*2 + (w&1).  Words of the row pitch beyond the last segment are zeroed.
#define GEN_THREADS 256
#define GEN_STRIDE 33                          // dwords per lane: odd, so the lanes' dwords fall on different banks
template <bool TILED>
__global__ __launch_bounds__(GEN_THREADS) void sampler_kernel(u64 seed, int64_t first_sample, int64_t count, int64_t n,
                                                              int64_t words, int64_t lde, SegTables th,
                                                              uint64_t* __restrict__ ex, uint64_t* __restrict__ ez) {
    __shared__ u64 cdf_lds[2 * GF2_SEG_CDF];
    __shared__ unsigned int xz_all[GEN_THREADS * GEN_STRIDE];
    stage_seg_cdf(th, cdf_lds);
    unsigned int* const xz = xz_all + threadIdx.x * GEN_STRIDE;
    const int64_t segs_row = TILED ? th.nseg : (lde + GF2_SEG_WORDS - 1) / GF2_SEG_WORDS;   // sample-major: the pitch is zero-filled
    const int64_t rows = TILED ? ((count + 63) / 64) * 64 : count;
    const int64_t total = rows * segs_row;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        const int64_t i = idx / segs_row;
        const int s = (int)(idx - i * segs_row);
        if (i < count && s < th.nseg) {
            const bool last = s == th.nseg - 1;
            sample_segment(sample_key(seed, (u64)(first_sample + i)), s, last ? th.nb_last : GF2_SEG_BITS,
                           cdf_lds + (last ? GF2_SEG_CDF : 0), th.t_1, th.t_2, xz);
        }
        // (a segment has zeroed and set the words below ceil(nb / 64) only; the words beyond are zero without being read)
        const bool drawn = i < count && s < th.nseg;
        const int used = drawn ? (((s == th.nseg - 1 ? th.nb_last : GF2_SEG_BITS) + 63) >> 6) * 2 : 0;
#pragma unroll
        for (int q = 0; q < GF2_SEG_WORDS; ++q) {
            const int64_t w = (int64_t)s * GF2_SEG_WORDS + q;
            u64 x = 0, z = 0;
            if (2 * q < used) x = ((u64)xz[2 * q + 1] << 32) | xz[2 * q], z = ((u64)xz[16 + 2 * q + 1] << 32) | xz[16 + 2 * q];
            if (TILED) {
                if (w < lde) {                                    // the tiled pitch: the words rounded up to an even number
                    const int64_t at = (i >> 6) * 64 * lde + (w >> 1) * 128 + (i & 63) * 2 + (w & 1);
                    ex[at] = x;
                    ez[at] = z;
                }
            } else if (w < lde) {
                ex[i * lde + w] = x;
                ez[i * lde + w] = z;
            }
        }
    }
}

// ---- a block of 64 segments per wavefront ---------------------------------------------------------------------------------
// Shared by the row sampler below and by the record sampler of the LDS-slab pipeline: lane = (sample, segment), NSP = segments
// per sample rounded up to a power of two, 64 / NSP samples per block.
//   1. every lane takes its segment's draw d and error count K (the table's first 16 entries sit in scalar registers: a count
//      is 16 compares, not a chain of dependent table reads);
//   2. the (segment, k) pairs of the block are laid out flat (prefix sum of K), so that everything per erroneous qubit -- its
//      draw (a mix64 and a multiply) and later its output -- runs on full wavefronts, 64 qubits per pass, whatever the
//      spread of K over the segments;
//   3. Floyd's rule (a candidate position that is taken already gives way to j = nb - K + k) is the one sequential thing: a lane
//      sends its K candidates through returning LDS atomics on a 512-bit map of its own, eight in flight, and notes which
//      came back taken.  A taken candidate is a duplicate of an earlier candidate; what the atomics cannot see is a later
//      candidate equal to the j an earlier duplicate moved to -- the (rare) lanes with duplicates check for that.
// Afterwards flat item f < E is qubit (src[f] & 63 = lane, src[f] >> 6 = k) with cand[f] = position | kind << 9.
// A block with more than SMP_FLAT erroneous qubits, or a segment with more than 64 (rates far beyond the sparse regime), is
// resolved lane by lane (seg_block_sequential).
#define SMP_WAVES 4
#define SMP_FLAT 768
#define SMP_TAKEN 17                            // dwords per lane of the position map: 16 + 1 (odd: the lanes' dwords spread over the banks)
struct alignas(16) SegBlockLds {
    u64 draw[64];                               // d of lane's segment
    unsigned short kn[64];                      // its K
    unsigned int taken[64 * SMP_TAKEN];
    unsigned short src[SMP_FLAT];
    unsigned short cand[SMP_FLAT];
};

// Steps 1 and 2 up to the flat layout: returns K, first (the lane's first flat item) and E (uniform).
__device__ __forceinline__ void seg_block_counts(const SegTables& th, u64 seed, u64 sample, int s, bool live, bool last, int lane,
                                                 u64* d_out, int* K_out, int* first_out, int* E_out) {
    u64 d = 0;
    int K = 0;
    if (live) {
        d = segment_draw(sample_key(seed, sample), (u64)s);
        const u64 u = d >> 32;
        const u64* const tab = th.cdf + (last ? GF2_SEG_CDF : 0);
        int nb = last ? th.nb_last : GF2_SEG_BITS;
#pragma unroll
        for (int k = 0; k < 16; ++k) K += u >= (last ? th.cdf[GF2_SEG_CDF + k] : th.cdf[k]) ? 1 : 0;    // uniform addresses: scalar loads
        if (K == 16)
            while (K < nb && u >= tab[K]) K += 1;
    }
    // inclusive prefix sum over the lanes with DPP moves (a shuffle goes through the LDS crossbar: six dependent round trips)
    int incl = K;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);     // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);     // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);     // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);     // row_shr:8: scans of the four rows of 16
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, false);    // row_bcast:15 into rows 1 and 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, false);    // row_bcast:31 into rows 2 and 3
    (void)lane;
    *d_out = d;
    *K_out = K;
    *first_out = incl - K;
    *E_out = __builtin_amdgcn_readlane(incl, 63);
}

// Steps 2 and 3 for a block with E <= SMP_FLAT and K <= 64 everywhere (uniform: the caller checks).
__device__ __forceinline__ void seg_block_positions(SegBlockLds& L, const SegTables& th, int lane, int nsp, u64 d, int K, int nb,
                                                    int first, int E) {
    L.draw[lane] = d;
    L.kn[lane] = (unsigned short)K;
#pragma unroll
    for (int q = 0; q < SMP_TAKEN; ++q) L.taken[q * 64 + lane] = 0;
    for (int k = 0; k < K; ++k) L.src[first + k] = (unsigned short)(lane | (k << 6));
    __builtin_amdgcn_wave_barrier();
    for (int f0 = lane; f0 < E; f0 += 3 * 64) {                        // three passes' LDS reads in flight together
        unsigned int m[3], t[3], kind[3];
        u64 dd[3];
        unsigned int kk[3];
#pragma unroll
        for (int u = 0; u < 3; ++u) m[u] = f0 + 64 * u < E ? (unsigned int)L.src[f0 + 64 * u] : 0u;
#pragma unroll
        for (int u = 0; u < 3; ++u) dd[u] = L.draw[m[u] & 63u], kk[u] = L.kn[m[u] & 63u];
#pragma unroll
        for (int u = 0; u < 3; ++u) {
            const bool from_last = (int)(m[u] & (unsigned int)(nsp - 1)) == th.nseg - 1;
            error_draw(dd[u], (int)(m[u] >> 6), (int)kk[u], from_last ? th.nb_last : GF2_SEG_BITS, th.t_1, th.t_2, &t[u], &kind[u]);
        }
#pragma unroll
        for (int u = 0; u < 3; ++u)
            if (f0 + 64 * u < E) L.cand[f0 + 64 * u] = (unsigned short)(t[u] | (kind[u] << 9));
    }
    __builtin_amdgcn_wave_barrier();
    unsigned int* const mine = L.taken + lane * SMP_TAKEN;
    u64 coll = 0, high = 0;                                            // high: candidates in [nb - K, nb), where the j live
    const unsigned int j0 = (unsigned int)(nb - K);
    for (int k0 = 0; k0 < K; k0 += 8) {
        unsigned int t[8], old[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = k0 + u < K ? (unsigned int)L.cand[first + k0 + u] & 511u : 0u;
#pragma unroll
        for (int u = 0; u < 8; ++u) old[u] = atomicOr(&mine[t[u] >> 5], k0 + u < K ? 1u << (t[u] & 31u) : 0u);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (k0 + u < K && ((old[u] >> (t[u] & 31u)) & 1u)) coll |= 1ull << (k0 + u);
            if (k0 + u < K && t[u] >= j0) high |= 1ull << (k0 + u);
        }
    }
    if (coll) {
        // duplicates move to j_i = nb - K + i; a later candidate equal to such a j_i is taken too (and moves in its turn).
        // Only candidates of `high` can equal a j, and there is hardly ever one.
        u64 todo = coll;
        while (todo) {
            const int i = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            u64 later = high & ~coll & (i < 63 ? ~0ull << (i + 1) : 0ull);
            while (later) {
                const int k = __ffsll((long long)later) - 1;
                later &= later - 1;
                if (((unsigned int)L.cand[first + k] & 511u) == j0 + (unsigned int)i) coll |= 1ull << k, todo |= 1ull << k;
            }
        }
        u64 moved = coll;
        while (moved) {
            const int k = __ffsll((long long)moved) - 1;
            moved &= moved - 1;
            L.cand[first + k] = (unsigned short)((L.cand[first + k] & 0x600u) | (j0 + (unsigned int)k));
        }
    }
    __builtin_amdgcn_wave_barrier();
}

// The same outcome, one lane per segment start to end, for blocks the flat layout does not hold: calls emit(position, kind).
template <typename Emit>
__device__ __forceinline__ void seg_block_sequential(SegBlockLds& L, const SegTables& th, int lane, u64 d, int K, int nb, Emit emit) {
    unsigned int* const mine = L.taken + lane * SMP_TAKEN;
    for (int q = 0; q < SMP_TAKEN; ++q) mine[q] = 0;
    for (int k = 0; k < K; ++k) {
        unsigned int t, kind;
        error_draw(d, k, K, nb, th.t_1, th.t_2, &t, &kind);
        const bool taken = (mine[t >> 5] >> (t & 31u)) & 1u;
        const unsigned int pos = taken ? (unsigned int)(nb - K + k) : t;
        mine[pos >> 5] |= 1u << (pos & 31u);
        emit(pos, kind);
    }
}

// Sample-major errors of 8 .. 64 words per sample (512 <= n <= 4096, or less with a wide pitch): the block's errors are set in an
// LDS image of its rows by flat passes over the erroneous qubits (LDS atomics without return), and the image leaves as whole rows.
struct alignas(16) SamplerWaveLds {
    SegBlockLds b;
    u64 img[2][64 * GF2_SEG_WORDS];             // e_x, e_z of the block: lane (sample, segment) owns words 8 lane .. 8 lane + 7
};

__global__ __launch_bounds__(64 * SMP_WAVES) void sampler_rows_kernel(u64 seed, int64_t first_sample, int64_t count, int words,
                                                                    int64_t lde, int nsp_log2, SegTables th,
                                                                    uint64_t* __restrict__ ex, uint64_t* __restrict__ ez) {
    __shared__ SamplerWaveLds lds_all[SMP_WAVES];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SamplerWaveLds& L = lds_all[wave];
    const int nsp = 1 << nsp_log2, per_block = 64 >> nsp_log2;        // segments per sample (padded), samples per block
    const int j = lane >> nsp_log2, s = lane & (nsp - 1);
    const bool seg_live = s < th.nseg;
    const bool last = s == th.nseg - 1;
    const int nb = last ? th.nb_last : GF2_SEG_BITS;
    unsigned int* const img_x = reinterpret_cast<unsigned int*>(&L.img[0][0]);
    unsigned int* const img_z = reinterpret_cast<unsigned int*>(&L.img[1][0]);
    const int64_t nblocks = (count + per_block - 1) / per_block;
    const int64_t total_waves = (int64_t)gridDim.x * SMP_WAVES;
#pragma unroll 1
    for (int64_t blk = (int64_t)blockIdx.x * SMP_WAVES + wave; blk < nblocks; blk += total_waves) {
        const int64_t i0 = blk * per_block;
        const bool live = seg_live && i0 + j < count;
#pragma unroll
        for (int q = 0; q < GF2_SEG_WORDS; ++q) L.img[0][q * 64 + lane] = 0, L.img[1][q * 64 + lane] = 0;
        u64 d;
        int K, first, E;
        seg_block_counts(th, seed, (u64)(first_sample + i0 + j), s, live, last, lane, &d, &K, &first, &E);
        const bool flat = E <= SMP_FLAT && __ballot(K > 64) == 0;      // uniform
        if (flat) {
            seg_block_positions(L.b, th, lane, nsp, d, K, nb, first, E);
            for (int f = lane; f < E; f += 64) {
                const unsigned int from = L.b.src[f] & 63u, c = L.b.cand[f], pos = c & 511u;
                if (c & 0x200u) atomicOr(&img_x[from * 16 + (pos >> 5)], 1u << (pos & 31u));
                if (c & 0x400u) atomicOr(&img_z[from * 16 + (pos >> 5)], 1u << (pos & 31u));
            }
        } else {
            __builtin_amdgcn_wave_barrier();
            if (live)
                seg_block_sequential(L.b, th, lane, d, K, nb, [&](unsigned int pos, unsigned int kind) {
                    if (kind & 1u) img_x[lane * 16 + (pos >> 5)] |= 1u << (pos & 31u);
                    if (kind & 2u) img_z[lane * 16 + (pos >> 5)] |= 1u << (pos & 31u);
                });
        }
        __builtin_amdgcn_wave_barrier();
        // rows out: sample jj of the block is words jj * nsp * 8 .. of the image, `words` of them valid, lde stored
        for (int jj = 0; jj < per_block; ++jj) {
            if (i0 + jj < count && lane < lde) {
                const bool in_row = lane < nsp * GF2_SEG_WORDS;
                ex[(i0 + jj) * lde + lane] = in_row ? L.img[0][jj * nsp * GF2_SEG_WORDS + lane] : 0ull;
                ez[(i0 + jj) * lde + lane] = in_row ? L.img[1][jj * nsp * GF2_SEG_WORDS + lane] : 0ull;
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
}

// ---- table decode + logical-error tally (small codes) --------------------------------------------------------------
struct DecodeRows {
    u64 h1[20], h2[20];
};

// One lane per sample, everything in registers: sample -> both syndromes -> table lookups -> residuals -> parities.
// counts: [0] logical X flips, [1] logical Z flips, [2] either, [3] X syndrome not in table, [4] Z syndrome not in table.
__global__ __launch_bounds__(256) void decode_kernel(DecodeRows rows, int r1, int r2, int n, const u64* __restrict__ t1,
                                                     const u64* __restrict__ t2, u64 xop, u64 zop, u64 seed,
                                                     int64_t first_sample, int64_t count, SamplerTables th,
                                                     u64* __restrict__ counts) {
    __shared__ unsigned int acc[5];
    __shared__ u64 cdf_lds[130];
    if (threadIdx.x < 5) acc[threadIdx.x] = 0;
    stage_cdf(th, cdf_lds);
    unsigned int local[5] = {0, 0, 0, 0, 0};
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        u64 ex, ez;
        sample_word(seed, (u64)(first_sample + i), 0, n, cdf_lds + 65, th.t_1, th.t_2, &ex, &ez);
        u64 kx = 0, kz = 0;                                   // vec_to_int keys: row 0 is the most significant bit
        for (int k = 0; k < r2; ++k) kx = (kx << 1) | (u64)(__popcll(rows.h2[k] & ex) & 1);
        for (int k = 0; k < r1; ++k) kz = (kz << 1) | (u64)(__popcll(rows.h1[k] & ez) & 1);
        const u64 cx = t2[kx], cz = t1[kz];
        const bool miss_x = cx == ~0ull, miss_z = cz == ~0ull;
        const u64 res_x = miss_x ? ex : ex ^ cx, res_z = miss_z ? ez : ez ^ cz;
        const bool flip_x = __popcll(zop & res_x) & 1, flip_z = __popcll(xop & res_z) & 1;
        local[0] += flip_x;
        local[1] += flip_z;
        local[2] += flip_x | flip_z;
        local[3] += miss_x;
        local[4] += miss_z;
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (local[k]) atomicAdd(&acc[k], local[k]);
    __syncthreads();
    if (threadIdx.x < 5 && acc[threadIdx.x]) atomicAdd(&counts[threadIdx.x], (u64)acc[threadIdx.x]);
}

// ---- fused Monte-Carlo for small codes (n <= 64, r_1, r_2 <= 20): sample -> syndromes -> histograms, no HBM traffic ----
// mode GF2_HIST_FULL: bins by the big-endian key; GF2_HIST_WEIGHT: bins by weight.  Bins privatised in LDS when both
// histograms fit 8192 bins together.
__global__ __launch_bounds__(256) void mc_small_kernel(DecodeRows rows, int r1, int r2, int n, int mode, u64 seed,
                                                       int64_t first_sample, int64_t count, SamplerTables th,
                                                       u64* __restrict__ hist_z, int nbz, u64* __restrict__ hist_x, int nbx) {
    __shared__ unsigned int bins[8192];
    const bool priv = nbz + nbx <= 8192;
    if (priv) {
        for (int i = threadIdx.x; i < nbz + nbx; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    __shared__ u64 cdf_lds[130];
    stage_cdf(th, cdf_lds);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += stride) {
        u64 ex, ez;
        sample_word(seed, (u64)(first_sample + i), 0, n, cdf_lds + 65, th.t_1, th.t_2, &ex, &ez);
        u64 kx = 0, kz = 0;
        if (mode == GF2_HIST_FULL) {
            for (int k = 0; k < r2; ++k) kx = (kx << 1) | (u64)(__popcll(rows.h2[k] & ex) & 1);
            for (int k = 0; k < r1; ++k) kz = (kz << 1) | (u64)(__popcll(rows.h1[k] & ez) & 1);
        } else {
            for (int k = 0; k < r2; ++k) kx += (u64)(__popcll(rows.h2[k] & ex) & 1);
            for (int k = 0; k < r1; ++k) kz += (u64)(__popcll(rows.h1[k] & ez) & 1);
        }
        if (priv) {
            atomicAdd(&bins[kz], 1u);
            atomicAdd(&bins[nbz + kx], 1u);
        } else {
            atomicAdd(&hist_z[kz], 1ull);
            atomicAdd(&hist_x[kx], 1ull);
        }
    }
    if (priv) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbz; i += blockDim.x)
            if (bins[i]) atomicAdd(&hist_z[i], (u64)bins[i]);
        for (int i = threadIdx.x; i < nbx; i += blockDim.x)
            if (bins[nbz + i]) atomicAdd(&hist_x[i], (u64)bins[nbz + i]);
    }
}

// The sampler's tables for (rates, n) in the context's device buffer; uploaded only when they change (a Monte-Carlo run asks
// once per chunk).  The upload waits for the context's stream AND its side streams: a kernel in flight on any of them (the
// packed-row route's pipelines run there) may still read the old tables.
int gf2_seg_tables(gf2_ctx* ctx, double p_x, double p_y, double p_z, int64_t n, SegTables* out) {
    GF2_TRY(check_probabilities(p_x, p_y, p_z));
    const double p_t = p_x + p_y + p_z, p_xy = p_x + p_y;
    const uint64_t t_any = gf2_quantise(p_t);
    out->t_1 = p_t > 0.0 ? gf2_quantise(p_x / p_t) : 0;
    out->t_2 = p_t > 0.0 ? gf2_quantise(p_xy / p_t) : 0;
    out->nseg = (int)gf2_cdiv(n, GF2_SEG_BITS);
    out->nb_last = n > 0 ? (int)(n - (int64_t)(out->nseg - 1) * GF2_SEG_BITS) : 0;
    if (!ctx->seg_cdf_dev) {
        GF2_HIP(hipMalloc((void**)&ctx->seg_cdf_dev, 2 * GF2_SEG_CDF * sizeof(uint64_t)));
        ctx->seg_key_nb = -1;
    }
    if (ctx->seg_key_nb != out->nb_last || ctx->seg_key_t != t_any) {
        std::vector<u64> host(2 * GF2_SEG_CDF);
        binomial_cdf_table(t_any, GF2_SEG_BITS, host.data(), GF2_SEG_CDF);
        binomial_cdf_table(t_any, out->nb_last, host.data() + GF2_SEG_CDF, GF2_SEG_CDF);
        GF2_TRY(gf2_stream_wait(ctx->stream));
        for (int k = 0; k < 2; ++k) GF2_TRY(gf2_stream_wait(ctx->side[k]));
        GF2_HIP(hipMemcpy(ctx->seg_cdf_dev, host.data(), host.size() * sizeof(u64), hipMemcpyHostToDevice));
        ctx->seg_key_nb = out->nb_last;
        ctx->seg_key_t = t_any;
        // The record sampler's lanes walk a segment's erroneous qubits in step: the smallest even count (up to 8) that at most one
        // sample in ten exceeds is where they stop, the rest is a leftover (gf2_slabs.hip).  host[k] = 2^32 P(K <= k).
        ctx->seg_tail_cap = 0;
        for (int c = 2; c <= 8 && !ctx->seg_tail_cap; c += 2)
            if (4294967296.0 - (double)host[c] <= 0.10 * 4294967296.0) ctx->seg_tail_cap = c;
    }
    out->cdf = (const u64*)ctx->seg_cdf_dev;
    return GF2_OK;
}

extern "C" {

int gf2_sample_errors_dev(gf2_ctx* ctx, int64_t n, uint64_t seed, int64_t first_sample, int64_t count, double p_x,
                          double p_y, double p_z, uint64_t* ex_dev, uint64_t* ez_dev, int64_t lde, int layout) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: null context");
    if (layout != GF2_LAYOUT_SAMPLE_MAJOR && layout != GF2_LAYOUT_TILED)
        GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: layout must be sample-major or tiled");
    if (layout == GF2_LAYOUT_TILED) lde = gf2_tiled_ld(n);
    if (n < 0 || count < 0 || first_sample < 0 || lde < gf2_words(n) || lde < 1)
        GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: bad shape");
    GF2_TRY(check_probabilities(p_x, p_y, p_z));
    if (count == 0) return GF2_OK;
    if (!ex_dev || !ez_dev) GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: null buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    if (n == 0) {
        if (layout == GF2_LAYOUT_SAMPLE_MAJOR) {
            GF2_TRY(gf2_dev_zero(ctx, ex_dev, (size_t)count * lde * 8));
            GF2_TRY(gf2_dev_zero(ctx, ez_dev, (size_t)count * lde * 8));
        }
        return GF2_OK;
    }
    SegTables th;
    GF2_TRY(gf2_seg_tables(ctx, p_x, p_y, p_z, n, &th));
    const int64_t segs_row = layout == GF2_LAYOUT_TILED ? th.nseg : gf2_cdiv(lde, GF2_SEG_WORDS);
    int64_t blocks = gf2_cdiv((layout == GF2_LAYOUT_TILED ? gf2_cdiv(count, 64) * 64 : count) * segs_row, GEN_THREADS);
    if (blocks > 16384) blocks = 16384;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_SAMPLER));
    if (layout == GF2_LAYOUT_TILED)
        hipLaunchKernelGGL(sampler_kernel<true>, dim3((unsigned)blocks), dim3(GEN_THREADS), 0, ctx->stream, (u64)seed, first_sample,
                           count, n, gf2_words(n), lde, th, ex_dev, ez_dev);
    else if (lde <= 64 && gf2_words(n) >= 8 && !gf2_flag(ctx, GF2_F_SAMPLER_GENERIC)) {
        int nsp_log2 = 0;
        while ((1 << nsp_log2) < th.nseg) nsp_log2 += 1;
        const int per_block = 64 >> nsp_log2;
        int64_t rblocks = gf2_cdiv(gf2_cdiv(count, per_block), SMP_WAVES);
        if (rblocks > (int64_t)ctx->num_cus * 8) rblocks = (int64_t)ctx->num_cus * 8;
        hipLaunchKernelGGL(sampler_rows_kernel, dim3((unsigned)rblocks), dim3(64 * SMP_WAVES), 0, ctx->stream, (u64)seed, first_sample,
                           count, (int)gf2_words(n), lde, nsp_log2, th, ex_dev, ez_dev);
    } else
        hipLaunchKernelGGL(sampler_kernel<false>, dim3((unsigned)blocks), dim3(GEN_THREADS), 0, ctx->stream, (u64)seed, first_sample,
                           count, n, gf2_words(n), lde, th, ex_dev, ez_dev);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

int gf2_mc_run(gf2_ctx* ctx, const gf2_check* c1, const gf2_check* c2, uint64_t seed, int64_t first_sample,
               int64_t count, double p_x, double p_y, double p_z, int mode, uint64_t* hist_z, int64_t nbins_z,
               uint64_t* hist_x, int64_t nbins_x) {
    if (!ctx || !c1 || !c2 || !hist_z || !hist_x) GF2_FAIL(GF2_E_ARG, "gf2_mc_run: null argument");
    if (c1->n != c2->n) GF2_FAIL(GF2_E_ARG, "gf2_mc_run: the two checks have different n");
    if (count < 0 || first_sample < 0) GF2_FAIL(GF2_E_ARG, "gf2_mc_run: negative range");
    const int64_t n = c1->n;
    const int64_t want_z = mode == GF2_HIST_FULL ? (c1->r <= 24 ? (1ll << c1->r) : -1) : c1->r + 1;
    const int64_t want_x = mode == GF2_HIST_FULL ? (c2->r <= 24 ? (1ll << c2->r) : -1) : c2->r + 1;
    if (mode != GF2_HIST_FULL && mode != GF2_HIST_WEIGHT) GF2_FAIL(GF2_E_ARG, "gf2_mc_run: unknown mode %d", mode);
    if (want_z < 0 || want_x < 0) GF2_FAIL(GF2_E_ARG, "gf2_mc_run: full histograms need r <= 24");
    if (nbins_z != want_z || nbins_x != want_x)
        GF2_FAIL(GF2_E_ARG, "gf2_mc_run: expected %lld and %lld bins", (long long)want_z, (long long)want_x);
    SamplerTables th;
    GF2_TRY(make_thresholds(p_x, p_y, p_z, n, &th));
    GF2_TRY(gf2_ctx_activate(ctx));

    // Small codes: one fused kernel, nothing but the histograms touches memory.
    if (c1->small && c2->small && c1->r <= 20 && c2->r <= 20 && n >= 1 && !gf2_flag(ctx, GF2_F_MC_PIPELINE)) {
        const size_t hzb = (size_t)nbins_z * 8, hxb = (size_t)nbins_x * 8;
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        GF2_TRY(gf2_ws_reserve(ctx, 0, al(hzb) + al(hxb)));
        uint64_t* dz = (uint64_t*)ctx->ws[0];
        uint64_t* dx = (uint64_t*)((char*)ctx->ws[0] + al(hzb));
        GF2_TRY(gf2_dev_zero(ctx, dz, hzb));
        GF2_TRY(gf2_dev_zero(ctx, dx, hxb));
        if (count > 0) {
            DecodeRows rows;
            memset(&rows, 0, sizeof(rows));
            for (int64_t k = 0; k < c1->r; ++k) rows.h1[k] = c1->rows_small[k];
            for (int64_t k = 0; k < c2->r; ++k) rows.h2[k] = c2->rows_small[k];
            int64_t blocks = gf2_cdiv(count, 256 * 16);
            if (blocks > 4096) blocks = 4096;
            if (blocks < 1) blocks = 1;
            GF2_TRY(gf2_prof_begin(ctx, GF2_K_SAMPLER));
            hipLaunchKernelGGL(mc_small_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, rows, (int)c1->r, (int)c2->r,
                               (int)n, mode, (u64)seed, first_sample, count, th, (u64*)dz, (int)nbins_z, (u64*)dx, (int)nbins_x);
            GF2_TRY(gf2_prof_end(ctx));
            GF2_HIP(hipGetLastError());
        }
        GF2_TRY(gf2_d2h(ctx, hist_z, dz, hzb));
        GF2_TRY(gf2_d2h(ctx, hist_x, dx, hxb));
        return GF2_OK;
    }
    // Sparse-error pipeline: when few bits are set per error the column kernel wins: sampler writes
    // sample-major errors, the sparse kernel accumulates the weight histograms directly, no syndromes stored.
    const double dens = (p_x + p_y > p_z + p_y ? p_x + p_y : p_z + p_y) * (double)n;
    // mid-size checks (n <= 512, r <= 256): sampler, then the lane-per-sample kernel per component on one stream (the serial
    // path below) -- 3.0e10 / 1.7e10 / 4.0e9 samples/s at n = 127 / 255 / 511 against 8.0e9 / 6.5e9 / 2.8e9 through the slab
    // pipelines and 2.8e9 / 2.3e9 / 2.0e9 with the sampler fused into the column-gather kernel
    const bool lanes = gf2_lane_ok(c1) && gf2_lane_ok(c2);
    if (mode == GF2_HIST_WEIGHT && dens <= 36.0 && gf2_mc_records_ok(c1, c2) && !gf2_flag(ctx, GF2_F_MC_DENSE) &&
        !gf2_flag(ctx, GF2_F_MC_UNFUSED) && !gf2_flag(ctx, GF2_F_MC_FUSED) && !gf2_flag(ctx, GF2_F_MC_ROWS) && count >= 65536 && !lanes) {
        // No packed rows at all: the record sampler (gf2_slabs.hip) writes what the gather kernels read -- the records of the
        // non-identity columns and the words under the identity block -- and the gather, combine and misfit kernels of the two
        // components follow, chunk by chunk on the context's stream.  (Measured with the sampler of chunk k + 1 on this stream and
        // the two components of chunk k on the side streams, two buffer sets: 1.68e9 samples/s against 1.81e9 in line -- a gather
        // workgroup needs a CU's LDS to itself and waits for the sampler's workgroups to drain wherever it lands, and all three
        // kernels are bound by instruction issue, so there is nothing for them to share.)
        // (chunks of 2^22 samples: 2.05e9 samples/s; 2^21: 1.97e9; 2^20: 1.83e9; 2^18: 1.57e9 -- profiles/r02_mc_chunk.log)
        int64_t chunk_s = 1ll << (ctx->opt[GF2_OPT_MC_CHUNK_LOG2] >= 0 ? ctx->opt[GF2_OPT_MC_CHUNK_LOG2] : 22);
        if (chunk_s > count) chunk_s = gf2_cdiv(count, 64) * 64;
        SegTables th;
        GF2_TRY(gf2_seg_tables(ctx, p_x, p_y, p_z, n, &th));
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t set_bytes = al(gf2_mc_records_bytes(n, chunk_s)), hzb = (size_t)nbins_z * 8, hxb = (size_t)nbins_x * 8;
        GF2_TRY(gf2_ws_reserve(ctx, 0, set_bytes + al(hzb) + al(hxb)));
        char* q = (char*)ctx->ws[0];
        uint64_t* dz = (uint64_t*)(q + set_bytes);
        uint64_t* dx = (uint64_t*)(q + set_bytes + al(hzb));
        GF2_TRY(gf2_dev_zero(ctx, dz, hzb));
        GF2_TRY(gf2_dev_zero(ctx, dx, hxb));
        GF2_TRY(gf2_mc_records_run(ctx, c1, c2, seed, first_sample, count, chunk_s, th, q, dz, dx, ctx->stream));
        GF2_TRY(gf2_d2h(ctx, hist_z, dz, hzb));
        GF2_TRY(gf2_d2h(ctx, hist_x, dx, hxb));
        return GF2_OK;
    }
    if (mode == GF2_HIST_WEIGHT && dens <= 160.0 && gf2_slabs_ok(c1) && gf2_slabs_ok(c2) && !gf2_flag(ctx, GF2_F_MC_DENSE) &&
        !gf2_flag(ctx, GF2_F_MC_UNFUSED) && !gf2_flag(ctx, GF2_F_MC_FUSED) && count >= 65536 && !lanes) {
        // Three streams: the sampler draws chunk k + 1 on the context's stream while the LDS-slab pipelines of the two
        // components work on chunk k on the two side streams (double-buffered errors; events carry the hand-overs).
        const int64_t lde_s = gf2_words(n);
        const int64_t chunk_s = 1ll << (ctx->opt[GF2_OPT_MC_CHUNK_LOG2] >= 0 ? ctx->opt[GF2_OPT_MC_CHUNK_LOG2] : 21);
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t eb = (size_t)chunk_s * lde_s * 8, hzb = (size_t)nbins_z * 8, hxb = (size_t)nbins_x * 8;
        GF2_TRY(gf2_ws_reserve(ctx, 0, 4 * al(eb) + al(hzb) + al(hxb)));
        GF2_TRY(gf2_slabs_reserve(ctx, c1, chunk_s, 2));
        GF2_TRY(gf2_slabs_reserve(ctx, c2, chunk_s, 3));
        char* q = (char*)ctx->ws[0];
        uint64_t* sx[2];
        uint64_t* sz[2];
        for (int b = 0; b < 2; ++b) {
            sx[b] = (uint64_t*)q; q += al(eb);
            sz[b] = (uint64_t*)q; q += al(eb);
        }
        uint64_t* dz = (uint64_t*)q; q += al(hzb);
        uint64_t* dx = (uint64_t*)q;
        hipEvent_t* sampled = ctx->side_ev;            // [2]
        hipEvent_t* done_z = ctx->side_ev + 2;         // [2]
        hipEvent_t* done_x = ctx->side_ev + 4;         // [2]
        GF2_TRY(gf2_dev_zero(ctx, dz, hzb));
        GF2_TRY(gf2_dev_zero(ctx, dx, hxb));
        int64_t k = 0;
        timespec t_begin;
        clock_gettime(CLOCK_MONOTONIC, &t_begin);
        // One chunk: sampler on the context's stream, the two pipelines on the side streams.  A failure in the middle must not
        // leave the side streams working on this call's buffers behind the caller's back: the loop joins them before returning.
        auto enqueue_chunk = [&](int64_t done, int64_t now, int b) -> int {
            if (k >= 2) {                                          // the pipelines of chunk k - 2 are done with these buffers
                GF2_HIP(hipStreamWaitEvent(ctx->stream, done_z[b], 0));
                GF2_HIP(hipStreamWaitEvent(ctx->stream, done_x[b], 0));
            }
            GF2_TRY(gf2_sample_errors_dev(ctx, n, seed, first_sample + done, now, p_x, p_y, p_z, sx[b], sz[b], lde_s,
                                          GF2_LAYOUT_SAMPLE_MAJOR));
            GF2_HIP(hipEventRecord(sampled[b], ctx->stream));
            GF2_HIP(hipStreamWaitEvent(ctx->side[0], sampled[b], 0));
            GF2_HIP(hipStreamWaitEvent(ctx->side[1], sampled[b], 0));
            GF2_TRY(gf2_syndrome_slabs(ctx, c1, sz[b], now, lde_s, dz, ctx->side[0], 2));
            GF2_TRY(gf2_syndrome_slabs(ctx, c2, sx[b], now, lde_s, dx, ctx->side[1], 3));
            GF2_HIP(hipEventRecord(done_z[b], ctx->side[0]));
            GF2_HIP(hipEventRecord(done_x[b], ctx->side[1]));
            return GF2_OK;
        };
        for (int64_t done = 0; done < count; done += chunk_s, ++k) {
            const int64_t now = count - done < chunk_s ? count - done : chunk_s;
            const int rc = enqueue_chunk(done, now, (int)(k & 1));
            if (rc != GF2_OK) {
                (void)hipStreamSynchronize(ctx->side[0]);
                (void)hipStreamSynchronize(ctx->side[1]);
                (void)hipStreamSynchronize(ctx->stream);
                return rc;
            }
        }
        timespec t_enqueued;
        clock_gettime(CLOCK_MONOTONIC, &t_enqueued);
        for (int b = 0; b < 2 && b < k; ++b) {                     // join: the histograms are read on the context's stream
            GF2_HIP(hipStreamWaitEvent(ctx->stream, done_z[b], 0));
            GF2_HIP(hipStreamWaitEvent(ctx->stream, done_x[b], 0));
        }
        GF2_TRY(gf2_d2h(ctx, hist_z, dz, hzb));
        timespec t_first;
        clock_gettime(CLOCK_MONOTONIC, &t_first);
        GF2_TRY(gf2_d2h(ctx, hist_x, dx, hxb));
        if (gf2_flag(ctx, GF2_F_DIAG_MC_TIMES)) {                              // diagnostic: where the host's time went
            timespec t_second;
            clock_gettime(CLOCK_MONOTONIC, &t_second);
            auto ms = [](const timespec& a, const timespec& b) { return (b.tv_sec - a.tv_sec) * 1e3 + (b.tv_nsec - a.tv_nsec) * 1e-6; };
            fprintf(stderr, "mc_run: enqueue %.2f ms, first d2h (waits for the GPU) %.2f ms, second d2h %.2f ms\n", ms(t_begin, t_enqueued),
                    ms(t_enqueued, t_first), ms(t_first, t_second));
        }
        return GF2_OK;
    }
    if (mode == GF2_HIST_WEIGHT && dens <= 160.0 && gf2_mc_sparse_fused_ok(c1, c2) && !gf2_flag(ctx, GF2_F_MC_DENSE) &&
        !gf2_flag(ctx, GF2_F_MC_UNFUSED) && (!lanes || gf2_flag(ctx, GF2_F_MC_FUSED))) {
        // one kernel: every lane draws its own error words, nothing but the histograms touches memory
        const size_t hzb = (size_t)nbins_z * 8, hxb = (size_t)nbins_x * 8;
        auto al2 = [](size_t v) { return (v + 255) & ~(size_t)255; };
        GF2_TRY(gf2_ws_reserve(ctx, 0, al2(hzb) + al2(hxb)));
        uint64_t* dz = (uint64_t*)ctx->ws[0];
        uint64_t* dx = (uint64_t*)((char*)ctx->ws[0] + al2(hzb));
        GF2_TRY(gf2_dev_zero(ctx, dz, hzb));
        GF2_TRY(gf2_dev_zero(ctx, dx, hxb));
        GF2_TRY(gf2_mc_sparse_fused(ctx, c1, c2, seed, first_sample, count, p_x, p_y, p_z, dz, dx));
        GF2_TRY(gf2_d2h(ctx, hist_z, dz, hzb));
        GF2_TRY(gf2_d2h(ctx, hist_x, dx, hxb));
        return GF2_OK;
    }
    if (mode == GF2_HIST_WEIGHT && c1->ht_dev && c2->ht_dev && dens <= 160.0 && !gf2_flag(ctx, GF2_F_MC_DENSE)) {
        const int64_t lde_s = gf2_words(n);
        int64_t chunk_s = (int64_t)(2048ll << 20) / (2 * lde_s * 8);      // 2 GiB of packed errors per round trip
        if (chunk_s > count) chunk_s = count > 0 ? count : 1;
        auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const size_t eb = (size_t)chunk_s * lde_s * 8, hzb = (size_t)nbins_z * 8, hxb = (size_t)nbins_x * 8;
        GF2_TRY(gf2_ws_reserve(ctx, 0, 2 * al(eb) + al(hzb) + al(hxb)));
        char* q = (char*)ctx->ws[0];
        uint64_t* sx = (uint64_t*)q; q += al(eb);
        uint64_t* sz = (uint64_t*)q; q += al(eb);
        uint64_t* dz = (uint64_t*)q; q += al(hzb);
        uint64_t* dx = (uint64_t*)q;
        GF2_TRY(gf2_dev_zero(ctx, dz, hzb));
        GF2_TRY(gf2_dev_zero(ctx, dx, hxb));
        for (int64_t done = 0; done < count; done += chunk_s) {
            const int64_t now = count - done < chunk_s ? count - done : chunk_s;
            GF2_TRY(gf2_sample_errors_dev(ctx, n, seed, first_sample + done, now, p_x, p_y, p_z, sx, sz, lde_s,
                                          GF2_LAYOUT_SAMPLE_MAJOR));
            GF2_TRY(gf2_syndrome_sparse_dev(ctx, c1, sz, now, lde_s, nullptr, 0, dz, nbins_z));
            GF2_TRY(gf2_syndrome_sparse_dev(ctx, c2, sx, now, lde_s, nullptr, 0, dx, nbins_x));
        }
        GF2_TRY(gf2_d2h(ctx, hist_z, dz, hzb));
        GF2_TRY(gf2_d2h(ctx, hist_x, dx, hxb));
        return GF2_OK;
    }
    const bool tiled = !c1->small || !c2->small;     // large checks read the tiled layout directly
    if (tiled && (c1->small || c2->small))
        GF2_FAIL(GF2_E_ARG, "gf2_mc_run: one check is small (<= 64 x 64) and the other is not; unsupported");
    const int layout = tiled ? GF2_LAYOUT_TILED : GF2_LAYOUT_SAMPLE_MAJOR;
    const int64_t lde = tiled ? gf2_tiled_ld(n) : (gf2_words(n) > 0 ? gf2_words(n) : 1);
    const int64_t sl1 = c1->slabs > 0 ? c1->slabs : 1, sl2 = c2->slabs > 0 ? c2->slabs : 1;
    // chunk sized to about 256 MiB of error words
    int64_t chunk = (int64_t)(256ll << 20) / (2 * lde * 8);
    if (chunk > (1ll << 22)) chunk = 1ll << 22;
    if (chunk < 4096) chunk = 4096;
    if (chunk > count) chunk = count > 0 ? count : 1;
    chunk = gf2_cdiv(chunk, 64) * 64;
    const size_t e_bytes = (size_t)chunk * lde * 8;
    const size_t s1_bytes = (size_t)chunk * sl1 * 8, s2_bytes = (size_t)chunk * sl2 * 8;
    // sample-major: chunk rows of `slabs` words; tiled: `slabs` rows of `chunk` words (slab-major)
    const int64_t ls1 = tiled ? chunk : sl1, ls2 = tiled ? chunk : sl2;
    const size_t hz_bytes = (size_t)nbins_z * 8, hx_bytes = (size_t)nbins_x * 8;
    auto align = [](size_t v) { return (v + 255) & ~(size_t)255; };
    const size_t total = 2 * align(e_bytes) + align(s1_bytes) + align(s2_bytes) + align(hz_bytes) + align(hx_bytes);
    GF2_TRY(gf2_ws_reserve(ctx, 0, total));
    char* p = (char*)ctx->ws[0];
    uint64_t* ex = (uint64_t*)p; p += align(e_bytes);
    uint64_t* ez = (uint64_t*)p; p += align(e_bytes);
    uint64_t* s1 = (uint64_t*)p; p += align(s1_bytes);
    uint64_t* s2 = (uint64_t*)p; p += align(s2_bytes);
    uint64_t* hz = (uint64_t*)p; p += align(hz_bytes);
    uint64_t* hx = (uint64_t*)p;
    GF2_TRY(gf2_dev_zero(ctx, hz, hz_bytes));
    GF2_TRY(gf2_dev_zero(ctx, hx, hx_bytes));
    for (int64_t done = 0; done < count; done += chunk) {
        const int64_t now = count - done < chunk ? count - done : chunk;
        GF2_TRY(gf2_sample_errors_dev(ctx, n, seed, first_sample + done, now, p_x, p_y, p_z, ex, ez, lde, layout));
        if (c1->r > 0) GF2_TRY(gf2_syndrome_dev(ctx, c1, ez, now, lde, layout, s1, ls1));
        if (c2->r > 0) GF2_TRY(gf2_syndrome_dev(ctx, c2, ex, now, lde, layout, s2, ls2));
        if (c1->r == 0) GF2_TRY(gf2_dev_zero(ctx, s1, s1_bytes));
        if (c2->r == 0) GF2_TRY(gf2_dev_zero(ctx, s2, s2_bytes));
        GF2_TRY(gf2_histogram_dev(ctx, s1, now, ls1, layout, c1->r, mode, hz, nbins_z));
        GF2_TRY(gf2_histogram_dev(ctx, s2, now, ls2, layout, c2->r, mode, hx, nbins_x));
    }
    GF2_TRY(gf2_d2h(ctx, hist_z, hz, hz_bytes));
    GF2_TRY(gf2_d2h(ctx, hist_x, hx, hx_bytes));
    return GF2_OK;
}

int gf2_mc_decode(gf2_ctx* ctx, const gf2_check* c1, const gf2_check* c2, const uint64_t* table_c1,
                  const uint64_t* table_c2, uint64_t x_operator, uint64_t z_operator, uint64_t seed, int64_t first_sample,
                  int64_t count, double p_x, double p_y, double p_z, uint64_t* counts_out) {
    if (!ctx || !c1 || !c2 || !table_c1 || !table_c2 || !counts_out) GF2_FAIL(GF2_E_ARG, "gf2_mc_decode: null argument");
    if (c1->n != c2->n) GF2_FAIL(GF2_E_ARG, "gf2_mc_decode: the two checks have different n");
    if (c1->n > 63 || c1->n < 1 || c1->r > 20 || c2->r > 20)
        GF2_FAIL(GF2_E_ARG, "gf2_mc_decode: needs n <= 63 and r_1, r_2 <= 20 (table decode of small codes)");
    if (count < 0 || first_sample < 0) GF2_FAIL(GF2_E_ARG, "gf2_mc_decode: negative range");
    SamplerTables th;
    GF2_TRY(make_thresholds(p_x, p_y, p_z, c1->n, &th));
    GF2_TRY(gf2_ctx_activate(ctx));
    for (int k = 0; k < 5; ++k) counts_out[k] = 0;
    if (count == 0) return GF2_OK;
    const size_t b1 = (size_t)8 << c1->r, b2 = (size_t)8 << c2->r;
    auto al = [](size_t v) { return (v + 255) & ~(size_t)255; };
    GF2_TRY(gf2_ws_reserve(ctx, 0, al(b1) + al(b2) + 256));
    char* q = (char*)ctx->ws[0];
    u64* t1 = (u64*)q; q += al(b1);
    u64* t2 = (u64*)q; q += al(b2);
    u64* counts = (u64*)q;
    GF2_TRY(gf2_h2d(ctx, t1, table_c1, b1));
    GF2_TRY(gf2_h2d(ctx, t2, table_c2, b2));
    GF2_TRY(gf2_dev_zero(ctx, counts, 40));
    DecodeRows rows;
    memset(&rows, 0, sizeof(rows));
    for (int64_t k = 0; k < c1->r; ++k) rows.h1[k] = c1->rows_small[k];
    for (int64_t k = 0; k < c2->r; ++k) rows.h2[k] = c2->rows_small[k];
    int64_t blocks = gf2_cdiv(count, 256 * 16);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_SAMPLER));
    hipLaunchKernelGGL(decode_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, rows, (int)c1->r, (int)c2->r,
                       (int)c1->n, t1, t2, (u64)x_operator, (u64)z_operator, (u64)seed, first_sample, count, th, counts);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    GF2_TRY(gf2_d2h(ctx, counts_out, counts, 40));
    return GF2_OK;
}

}  // extern "C"
