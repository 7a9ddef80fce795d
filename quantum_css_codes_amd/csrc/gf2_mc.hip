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

// One lane per output word.  Sample-major: idx = i * lde + w.  Tiled: idx is the tiled word offset
// (i>>6)*64*ldt + (w>>1)*128 + (i&63)*2 + (w&1).  Either way consecutive lanes write consecutive words.
template <bool TILED>
__global__ __launch_bounds__(256) void sampler_kernel(u64 seed, int64_t first_sample, int64_t count, int64_t n,
                                                      int64_t words, int64_t lde, int64_t total, SamplerTables th,
                                                      uint64_t* __restrict__ ex, uint64_t* __restrict__ ez) {
    __shared__ u64 cdf_lds[130];
    stage_cdf(th, cdf_lds);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += stride) {
        int64_t i, w;
        if (TILED) {
            const int64_t tile = idx / (64 * lde), rem = idx - tile * 64 * lde;
            i = tile * 64 + ((rem & 127) >> 1);
            w = (rem >> 7) * 2 + (rem & 1);
        } else {
            i = idx / lde;
            w = idx - i * lde;
        }
        u64 x = 0, z = 0;
        if (i < count && w < words) {
            const bool last = w == words - 1;
            sample_word(seed, (u64)(first_sample + i), (u64)w, last ? th.nb_last : 64, cdf_lds + (last ? 65 : 0), th.t_1,
                        th.t_2, &x, &z);
        }
        ex[idx] = x;
        ez[idx] = z;
    }
}

// Sample-major errors of up to 64 words per sample: one wavefront per block of 8 samples, lane = word.  Every word takes its
// one draw and writes zeros into an LDS image of the block; the words with errors (a third of them at p = 0.01) queue up in a
// ring and are worked off 64 at a time, so that the draws per erroneous qubit run on full wavefronts instead of on the
// few lanes of each sample that need them; the image then leaves as whole rows.
#define SMP_BLOCK 8
#define SMP_WAVES 4
#define SMP_RING 128
struct alignas(16) SamplerWaveLds {
    u64 ring_d[SMP_RING];                       // the word's draw
    unsigned short ring_m[SMP_RING];            // sample in block << 6 | word, error count << 9
    u64 img[2][SMP_BLOCK][64];                  // e_x, e_z of the block
};

__global__ __launch_bounds__(64 * SMP_WAVES) void sampler_rows_kernel(u64 seed, int64_t first_sample, int64_t count, int words,
                                                                    int64_t lde, SamplerTables th, uint64_t* __restrict__ ex,
                                                                    uint64_t* __restrict__ ez) {
    __shared__ u64 cdf_lds[130];
    __shared__ SamplerWaveLds lds_all[SMP_WAVES];
    stage_cdf(th, cdf_lds);
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    SamplerWaveLds& L = lds_all[wave];
    const bool live = lane < words;
    const bool last = lane == words - 1;
    const int nb = last ? th.nb_last : 64;
    const u64* const cdf = cdf_lds + (last ? 65 : 0);
    const u64 cdf0 = cdf[0];
    const int64_t nblocks = (count + SMP_BLOCK - 1) / SMP_BLOCK;
    const int64_t total_waves = (int64_t)gridDim.x * SMP_WAVES;
    unsigned int head = 0, tail = 0;
    auto process = [&](unsigned int first, unsigned int n_items) {
        if ((unsigned int)lane < n_items) {
            const unsigned int at = (first + lane) & (SMP_RING - 1);
            const u64 d = L.ring_d[at];
            const unsigned int m = L.ring_m[at];
            const unsigned int sw = m & 511u, w = m & 63u;
            u64 x, z;
            place_errors(d, (int)(m >> 9), (int)w == words - 1 ? th.nb_last : 64, th.t_1, th.t_2, &x, &z);
            (&L.img[0][0][0])[sw] = x;
            (&L.img[1][0][0])[sw] = z;
        }
    };
#pragma unroll 1
    for (int64_t blk = (int64_t)blockIdx.x * SMP_WAVES + wave; blk < nblocks; blk += total_waves) {
        const int64_t i0 = blk * SMP_BLOCK;
#pragma unroll 1
        for (int s = 0; s < SMP_BLOCK; ++s) {
            // the sample's key is the same in every lane: made scalar by hand, the compiler keeps it in vector registers
            const u64 si = (u64)(first_sample + i0 + s);
            const u64 si_s = ((u64)(unsigned int)__builtin_amdgcn_readfirstlane((int)(si >> 32)) << 32) |
                             (unsigned int)__builtin_amdgcn_readfirstlane((int)si);
            const u64 ks = sample_key(seed, si_s);
            L.img[0][s][lane] = 0;
            L.img[1][s][lane] = 0;
            int k_err = 0;
            u64 d = 0;
            if (live && i0 + s < count) {
                d = word_draw(ks, (u64)lane);
                if ((d >> 32) >= cdf0) k_err = error_count(d, nb, cdf);
            }
            const u64 act = __ballot(k_err > 0);
            if (k_err > 0) {
                const unsigned int pos = (tail + __builtin_amdgcn_mbcnt_hi((unsigned int)(act >> 32),
                                                                             __builtin_amdgcn_mbcnt_lo((unsigned int)act, 0u))) &
                                         (SMP_RING - 1);
                L.ring_d[pos] = d;
                L.ring_m[pos] = (unsigned short)((unsigned int)(s << 6) | (unsigned int)lane | ((unsigned int)k_err << 9));
            }
            tail += (unsigned int)__popcll(act);
            if (tail - head >= 64) {                                // uniform
                __builtin_amdgcn_wave_barrier();
                process(head, 64);
                head += 64;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (tail != head) process(head, tail - head);
        head = tail;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int s = 0; s < SMP_BLOCK; ++s) {
            if (i0 + s < count && lane < lde) {
                ex[(i0 + s) * lde + lane] = L.img[0][s][lane];
                ez[(i0 + s) * lde + lane] = L.img[1][s][lane];
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

extern "C" {

int gf2_sample_errors_dev(gf2_ctx* ctx, int64_t n, uint64_t seed, int64_t first_sample, int64_t count, double p_x,
                          double p_y, double p_z, uint64_t* ex_dev, uint64_t* ez_dev, int64_t lde, int layout) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: null context");
    if (layout != GF2_LAYOUT_SAMPLE_MAJOR && layout != GF2_LAYOUT_TILED)
        GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: layout must be sample-major or tiled");
    if (layout == GF2_LAYOUT_TILED) lde = gf2_tiled_ld(n);
    if (n < 0 || count < 0 || first_sample < 0 || lde < gf2_words(n) || lde < 1)
        GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: bad shape");
    SamplerTables th;
    GF2_TRY(make_thresholds(p_x, p_y, p_z, n, &th));
    if (count == 0) return GF2_OK;
    if (!ex_dev || !ez_dev) GF2_FAIL(GF2_E_ARG, "gf2_sample_errors_dev: null buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    const int64_t total = layout == GF2_LAYOUT_TILED ? gf2_tiled_words(n, count) : count * lde;
    int64_t blocks = gf2_cdiv(total, 256);
    if (blocks > 16384) blocks = 16384;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_SAMPLER));
    if (layout == GF2_LAYOUT_TILED)
        hipLaunchKernelGGL(sampler_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (u64)seed, first_sample,
                           count, n, gf2_words(n), lde, total, th, ex_dev, ez_dev);
    else if (lde <= 64 && gf2_words(n) >= 8 && !gf2_flag(ctx, GF2_F_SAMPLER_GENERIC)) {
        int64_t rblocks = gf2_cdiv(gf2_cdiv(count, SMP_BLOCK), SMP_WAVES);
        if (rblocks > (int64_t)ctx->num_cus * 8) rblocks = (int64_t)ctx->num_cus * 8;
        hipLaunchKernelGGL(sampler_rows_kernel, dim3((unsigned)rblocks), dim3(64 * SMP_WAVES), 0, ctx->stream, (u64)seed, first_sample,
                           count, (int)gf2_words(n), lde, th, ex_dev, ez_dev);
    } else
        hipLaunchKernelGGL(sampler_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, (u64)seed, first_sample,
                           count, n, gf2_words(n), lde, total, th, ex_dev, ez_dev);
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
    if (mode == GF2_HIST_WEIGHT && dens <= 160.0 && gf2_slabs_ok(c1) && gf2_slabs_ok(c2) && !gf2_flag(ctx, GF2_F_MC_DENSE) &&
        !gf2_flag(ctx, GF2_F_MC_UNFUSED) && !gf2_flag(ctx, GF2_F_MC_FUSED) && count >= 65536 && !lanes) {
        // Three streams: the sampler draws chunk k + 1 on the context's stream while the LDS-slab pipelines of the two
        // components work on chunk k on the two side streams (double-buffered errors; events carry the hand-overs).
        const int64_t lde_s = gf2_words(n);
        const int64_t chunk_s = 1ll << 21;
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
