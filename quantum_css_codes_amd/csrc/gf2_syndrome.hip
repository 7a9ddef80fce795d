// Syndrome extraction S = E . H^T over GF(2) on packed words, and syndrome histograms (gfx950).
//
// Replaces np.mod(np.matmul(parity_check, e), 2) of css_code.py:728 (and the commutation check of
// css_code.py:47) for a whole batch of errors.  Three kernels:
//
//   * syndrome_tables_kernel   Method-of-Four-Russians tables in LDS.  A workgroup owns one slab of 64
//                              parity-check rows: for every group of 4 columns the 16 XOR-combinations
//                              of the slab's column words (16 x 8 B per group, 128 KiB at n = 4096).
//                              A lane owns a sample; each nibble of its error word selects one table
//                              entry (ds_read_b64).  The 16 entries of a group span 32 distinct banks
//                              and equal addresses broadcast, so the reads are conflict-free.
//   * syndrome_small_kernel    n <= 64, r <= 64, sample-major: one lane per sample, rows in SGPRs,
//                              parity by AND + popcount.  Pure streaming.
//   * syndrome_sliced_kernel   n <= 64, r <= 64, bit-sliced: one lane per 64 samples, S[i] = XOR of the
//                              E[q] selected by row i.  Pure streaming at n + r bits per sample.
#include <stdlib.h>
#include <string.h>

#include "gf2_internal.h"

// ---- table construction ---------------------------------------------------------------------------------

// grid (slabs, ceil(groups / 64)), block 64.  Lane i of the wave holds row 64*slab+i.
__global__ void build_tables_kernel(const uint64_t* __restrict__ h, int64_t r, int64_t ld, int64_t groups,
                                    u64* __restrict__ tables) {
    const int lane = threadIdx.x;
    const int64_t slab = blockIdx.x;
    const int64_t row = slab * 64 + lane;
    const int64_t g0 = (int64_t)blockIdx.y * 64;
    for (int gi = 0; gi < 64; ++gi) {
        const int64_t g = g0 + gi;
        if (g >= groups) break;
        const int64_t word = g >> 4;
        u64 w = 0;
        if (row < r && word < ld) w = h[row * ld + word];
        const int shift = (int)(g & 15) * 4;
        u64 col[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) col[c] = __ballot((w >> (shift + c)) & 1ull);
        if (lane < 16) {
            u64 acc = 0;
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if ((lane >> c) & 1) acc ^= col[c];
            tables[(slab * groups + g) * 16 + lane] = acc;
        }
    }
}

// ---- Four-Russians syndrome kernel ------------------------------------------------------------------------

#define SYN_THREADS 1024
#define SYN_SPT 4                                  // samples per thread
#define SYN_BLOCK_SAMPLES (SYN_THREADS * SYN_SPT)
#define SYN_MAX_GROUPS 1024                         // 1024 groups x 128 B = 128 KiB of LDS per column pass

// Blocks are dealt round-robin over the 8 XCDs, so blocks b and b+8 share an L2.  Give the blocks of one
// XCD the same sample chunk and different slabs: the chunk's errors are then fetched from HBM once per
// XCD and served from its L2 to the other slabs.
__global__ __launch_bounds__(SYN_THREADS) void syndrome_tables_kernel(
    const u64* __restrict__ tables, int64_t groups, int64_t slabs, const uint64_t* __restrict__ e,
    int64_t batch, int64_t lde, int64_t e_words, uint64_t* __restrict__ s, int64_t lds_out, int64_t chunks) {
    extern __shared__ __attribute__((aligned(16))) u64 tab[];
    const int64_t b = blockIdx.x;
    const int64_t xcd = b & 7, q = b >> 3;
    const int64_t slab = q % slabs;
    const int64_t chunk = (q / slabs) * 8 + xcd;
    if (chunk >= chunks) return;

    const int64_t first = chunk * SYN_BLOCK_SAMPLES + threadIdx.x;
    u64 acc[SYN_SPT];
#pragma unroll
    for (int k = 0; k < SYN_SPT; ++k) acc[k] = 0;

    for (int64_t g0 = 0; g0 < groups; g0 += SYN_MAX_GROUPS) {
        const int64_t gn = (groups - g0 < SYN_MAX_GROUPS) ? groups - g0 : SYN_MAX_GROUPS;
        if (g0) __syncthreads();
        {   // stage this slab's tables for column groups g0 .. g0+gn-1 (16-byte copies, coalesced)
            const uint4* src = reinterpret_cast<const uint4*>(tables + (slab * groups + g0) * 16);
            uint4* dst = reinterpret_cast<uint4*>(tab);
            for (int64_t i = threadIdx.x; i < gn * 8; i += SYN_THREADS) dst[i] = src[i];
        }
        __syncthreads();
        const int64_t w0 = g0 >> 4;                       // first error word of this pass
        const int64_t wn = (gn + 15) >> 4;
#pragma unroll
        for (int k = 0; k < SYN_SPT; ++k) {
            const int64_t sample = first + (int64_t)k * SYN_THREADS;
            if (sample >= batch) continue;
            const uint64_t* erow = e + sample * lde;
            u64 a = acc[k];
            for (int64_t w = 0; w < wn; ++w) {
                const u64 v = (w0 + w < e_words) ? erow[w0 + w] : 0ull;
                const u64* t = tab + w * 256;
#pragma unroll
                for (int j = 0; j < 16; ++j) a ^= t[j * 16 + ((v >> (4 * j)) & 15ull)];
            }
            acc[k] = a;
        }
    }
#pragma unroll
    for (int k = 0; k < SYN_SPT; ++k) {
        const int64_t sample = first + (int64_t)k * SYN_THREADS;
        if (sample < batch) s[sample * lds_out + slab] = acc[k];
    }
}

// ---- small codes: n <= 64, r <= 64 -----------------------------------------------------------------------

struct SmallRows {
    u64 row[64];
};

__global__ __launch_bounds__(256) void syndrome_small_kernel(SmallRows rows, int r, const uint64_t* __restrict__ e,
                                                             int64_t batch, int64_t lde,
                                                             uint64_t* __restrict__ s, int64_t lds_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < batch; i += stride) {
        const u64 v = e[i * lde];
        u64 out = 0;
        for (int k = 0; k < r; ++k) out |= (u64)(__popcll(rows.row[k] & v) & 1) << k;
        s[i * lds_out] = out;
    }
}

template <int NMAX>
__global__ __launch_bounds__(256) void syndrome_sliced_kernel(SmallRows rows, int r, int n,
                                                              const uint64_t* __restrict__ e, int64_t words,
                                                              int64_t lde, uint64_t* __restrict__ s,
                                                              int64_t lds_out) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t b = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; b < words; b += stride) {
        u64 v[NMAX];
#pragma unroll
        for (int q = 0; q < NMAX; ++q) v[q] = (q < n) ? e[(int64_t)q * lde + b] : 0ull;
        for (int k = 0; k < r; ++k) {
            const u64 row = rows.row[k];
            u64 acc = 0;
#pragma unroll
            for (int q = 0; q < NMAX; ++q) acc ^= v[q] & (0ull - ((row >> q) & 1ull));
            s[(int64_t)k * lds_out + b] = acc;
        }
    }
}

// ---- histograms ------------------------------------------------------------------------------------------

#define HIST_LDS_BINS 8192

// mode 0: key = big-endian integer of the r syndrome bits (row 0 most significant, bin_matrix.py:36-43);
// mode 1: key = Hamming weight.  Bins privatised in LDS when they fit, one global atomic per bin and block.
__global__ __launch_bounds__(256) void histogram_kernel(const uint64_t* __restrict__ s, int64_t batch, int64_t lds_in,
                                                        int r, int mode, u64* __restrict__ hist, int64_t nbins) {
    __shared__ unsigned int bins[HIST_LDS_BINS];
    const bool priv = nbins <= HIST_LDS_BINS;
    if (priv) {
        for (int i = threadIdx.x; i < nbins; i += blockDim.x) bins[i] = 0;
        __syncthreads();
    }
    const int words = (r + 63) >> 6;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < batch; i += stride) {
        const uint64_t* row = s + i * lds_in;
        u64 key;
        if (mode == GF2_HIST_FULL) {
            key = r ? (__brevll(row[0]) >> (64 - r)) : 0ull;
        } else {
            key = 0;
            for (int w = 0; w < words; ++w) key += __popcll(row[w]);
        }
        if (priv)
            atomicAdd(&bins[key], 1u);
        else
            atomicAdd(&hist[key], 1ull);
    }
    if (priv) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbins; i += blockDim.x)
            if (bins[i]) atomicAdd(&hist[i], (u64)bins[i]);
    }
}

// ---- host side -------------------------------------------------------------------------------------------

extern "C" {

int gf2_check_create(gf2_ctx* ctx, const uint64_t* h, int64_t r, int64_t n, int64_t ld, gf2_check** check_out) {
    if (!ctx || !check_out) GF2_FAIL(GF2_E_ARG, "gf2_check_create: null argument");
    *check_out = nullptr;
    if (r < 0 || n < 0 || ld < gf2_words(n) || (!h && r > 0 && n > 0))
        GF2_FAIL(GF2_E_ARG, "gf2_check_create: bad shape r=%lld n=%lld ld=%lld", (long long)r, (long long)n, (long long)ld);
    GF2_TRY(gf2_ctx_activate(ctx));
    gf2_check* ck = (gf2_check*)calloc(1, sizeof(gf2_check));
    if (!ck) GF2_FAIL(GF2_E_NOMEM, "gf2_check_create: out of host memory");
    ck->r = r;
    ck->n = n;
    ck->ld = ld > 0 ? ld : 1;
    ck->slabs = gf2_cdiv(r, 64);
    ck->groups = gf2_cdiv(gf2_cdiv(n, 4), 16) * 16;         // whole error words
    if (ck->groups == 0) ck->groups = 16;
    if (n <= 64 && r <= 64)
        for (int64_t i = 0; i < r; ++i) ck->rows_small[i] = h[i * ld];
    int rc = GF2_OK;
    const size_t hbytes = (size_t)(r > 0 ? r : 1) * ck->ld * 8;
    const size_t tbytes = (size_t)(ck->slabs > 0 ? ck->slabs : 1) * ck->groups * 16 * 8;
    if ((rc = gf2_dev_alloc(ctx, hbytes, (void**)&ck->h_dev)) != GF2_OK) goto fail;
    if ((rc = gf2_dev_alloc(ctx, tbytes, (void**)&ck->tables_dev)) != GF2_OK) goto fail;
    if (r > 0) {
        if (ld == ck->ld) {
            if ((rc = gf2_h2d(ctx, ck->h_dev, h, (size_t)r * ld * 8)) != GF2_OK) goto fail;
        } else {
            if ((rc = gf2_dev_zero(ctx, ck->h_dev, hbytes)) != GF2_OK) goto fail;
        }
        dim3 grid((unsigned)ck->slabs, (unsigned)gf2_cdiv(ck->groups, 64));
        hipLaunchKernelGGL(build_tables_kernel, grid, dim3(64), 0, ctx->stream, ck->h_dev, r, ck->ld, ck->groups,
                           (u64*)ck->tables_dev);
        hipError_t err = hipGetLastError();
        if (err == hipSuccess) err = hipStreamSynchronize(ctx->stream);
        if (err != hipSuccess) {
            gf2_set_error("build_tables_kernel failed: %s", hipGetErrorString(err));
            rc = GF2_E_HIP;
            goto fail;
        }
    }
    *check_out = ck;
    return GF2_OK;
fail:
    if (ck->h_dev) (void)hipFree(ck->h_dev);
    if (ck->tables_dev) (void)hipFree(ck->tables_dev);
    free(ck);
    return rc;
}

int gf2_check_destroy(gf2_ctx* ctx, gf2_check* check) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_check_destroy: null context");
    if (!check) return GF2_OK;
    GF2_TRY(gf2_dev_free(ctx, check->h_dev));
    GF2_TRY(gf2_dev_free(ctx, check->tables_dev));
    free(check);
    return GF2_OK;
}

int gf2_syndrome_dev(gf2_ctx* ctx, const gf2_check* ck, const uint64_t* e_dev, int64_t batch, int64_t lde,
                     int layout, uint64_t* s_dev, int64_t lds) {
    if (!ctx || !ck) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: null argument");
    if (batch < 0) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: negative batch");
    if (batch == 0 || ck->r == 0) return GF2_OK;
    if (!e_dev || !s_dev) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: null buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    const bool small = ck->n <= 64 && ck->r <= 64;

    if (layout == GF2_LAYOUT_BIT_SLICED) {
        if (!small) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: bit-sliced layout needs n <= 64 and r <= 64");
        const int64_t words = gf2_words(batch);
        if (lde < words || lds < words) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: bit-sliced strides too small");
        SmallRows rows;
        memcpy(rows.row, ck->rows_small, sizeof(rows.row));
        int64_t blocks = gf2_cdiv(words, 256);
        if (blocks > 8192) blocks = 8192;
        dim3 grid((unsigned)blocks), block(256);
        GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
        const int n = (int)ck->n, r = (int)ck->r;
        if (n <= 8)
            hipLaunchKernelGGL(syndrome_sliced_kernel<8>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        else if (n <= 16)
            hipLaunchKernelGGL(syndrome_sliced_kernel<16>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        else if (n <= 32)
            hipLaunchKernelGGL(syndrome_sliced_kernel<32>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        else
            hipLaunchKernelGGL(syndrome_sliced_kernel<64>, grid, block, 0, ctx->stream, rows, r, n, e_dev, words, lde, s_dev, lds);
        GF2_TRY(gf2_prof_end(ctx));
        GF2_HIP(hipGetLastError());
        return GF2_OK;
    }
    if (layout != GF2_LAYOUT_SAMPLE_MAJOR) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: unknown layout %d", layout);
    if (lde < gf2_words(ck->n) || lde < 1) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: lde too small");
    if (lds < ck->slabs) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: lds too small");

    if (small) {
        SmallRows rows;
        memcpy(rows.row, ck->rows_small, sizeof(rows.row));
        int64_t blocks = gf2_cdiv(batch, 256);
        if (blocks > 8192) blocks = 8192;
        GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
        hipLaunchKernelGGL(syndrome_small_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, rows, (int)ck->r,
                           e_dev, batch, lde, s_dev, lds);
        GF2_TRY(gf2_prof_end(ctx));
        GF2_HIP(hipGetLastError());
        return GF2_OK;
    }

    const int64_t chunks = gf2_cdiv(batch, SYN_BLOCK_SAMPLES);
    const int64_t chunks8 = gf2_cdiv(chunks, 8) * 8;
    const int64_t blocks = chunks8 * ck->slabs;
    if (blocks > 0x7fffffffLL) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_dev: batch too large for one launch");
    const int64_t pass_groups = ck->groups < SYN_MAX_GROUPS ? ck->groups : SYN_MAX_GROUPS;
    const size_t shmem = (size_t)pass_groups * 16 * 8;
    static bool attr_set = false;
    if (!attr_set) {
        GF2_HIP(hipFuncSetAttribute((const void*)syndrome_tables_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                                    SYN_MAX_GROUPS * 16 * 8));
        attr_set = true;
    }
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_SYNDROME));
    hipLaunchKernelGGL(syndrome_tables_kernel, dim3((unsigned)blocks), dim3(SYN_THREADS), shmem, ctx->stream,
                       (const u64*)ck->tables_dev, ck->groups, ck->slabs, e_dev, batch, lde, gf2_words(ck->n), s_dev,
                       lds, chunks);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

int gf2_syndrome_batch(gf2_ctx* ctx, const uint64_t* h, int64_t r, int64_t n, int64_t ldh, const uint64_t* e,
                       int64_t batch, int64_t lde, int layout, uint64_t* s_out, int64_t lds) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: null context");
    if (batch < 0 || r < 0 || n < 0) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: negative size");
    if (batch == 0 || r == 0) return GF2_OK;
    if (!e || !s_out) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: null buffer");
    int64_t e_rows, s_rows;
    if (layout == GF2_LAYOUT_BIT_SLICED) {
        e_rows = n;
        s_rows = r;
    } else {
        e_rows = batch;
        s_rows = batch;
        if (lds < gf2_cdiv(r, 64)) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_batch: lds too small");
    }
    gf2_check* ck = nullptr;
    uint64_t *e_dev = nullptr, *s_dev = nullptr;
    int rc = gf2_check_create(ctx, h, r, n, ldh, &ck);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)e_rows * lde * 8, (void**)&e_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)s_rows * lds * 8, (void**)&s_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, e_dev, e, (size_t)e_rows * lde * 8);
    if (rc == GF2_OK) rc = gf2_dev_zero(ctx, s_dev, (size_t)s_rows * lds * 8);
    if (rc == GF2_OK) rc = gf2_syndrome_dev(ctx, ck, e_dev, batch, lde, layout, s_dev, lds);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, s_out, s_dev, (size_t)s_rows * lds * 8);
    gf2_dev_free(ctx, e_dev);
    gf2_dev_free(ctx, s_dev);
    gf2_check_destroy(ctx, ck);
    return rc;
}

int gf2_matmul_abt(gf2_ctx* ctx, const uint64_t* a, int64_t ra, int64_t lda, const uint64_t* b, int64_t rb,
                   int64_t ldb, int64_t n, uint64_t* c, int64_t ldc) {
    // C[i][j] = <A_i, B_j>: the rows of A are the "errors", B is the "parity check".
    if (ldc < gf2_cdiv(rb, 64)) GF2_FAIL(GF2_E_ARG, "gf2_matmul_abt: ldc too small");
    return gf2_syndrome_batch(ctx, b, rb, n, ldb, a, ra, lda, GF2_LAYOUT_SAMPLE_MAJOR, c, ldc);
}

int gf2_histogram_dev(gf2_ctx* ctx, const uint64_t* s_dev, int64_t batch, int64_t lds, int64_t r, int mode,
                      uint64_t* hist_dev, int64_t nbins) {
    if (!ctx || !hist_dev) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: null argument");
    if (batch < 0 || r < 0) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: negative size");
    if (mode == GF2_HIST_FULL) {
        if (r > 24) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: full histogram needs r <= 24 (r=%lld)", (long long)r);
        if (nbins != (1ll << r)) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: full histogram needs 2^r bins");
    } else if (mode == GF2_HIST_WEIGHT) {
        if (nbins != r + 1) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: weight histogram needs r+1 bins");
    } else {
        GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: unknown mode %d", mode);
    }
    if (batch == 0) return GF2_OK;
    if (!s_dev || lds < gf2_cdiv(r, 64) || lds < 1) GF2_FAIL(GF2_E_ARG, "gf2_histogram_dev: bad syndrome buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    int64_t blocks = gf2_cdiv(batch, 256 * 8);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_HIST));
    hipLaunchKernelGGL(histogram_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, s_dev, batch, lds, (int)r,
                       mode, (u64*)hist_dev, nbins);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

}  // extern "C"
