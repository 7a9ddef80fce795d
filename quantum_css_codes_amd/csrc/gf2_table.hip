// css_code.syndrome_table (css_code.py:715-735) on the device for codes of at most 24 checks (gfx950): one-word errors (n <= 64),
// two-word errors (n <= 128) and errors as position lists (n <= 8192).
//
// The reference walks the weight classes w = 0, 1, 2, ..., maps every error of weight w to vec_to_int(H e mod 2)
// (bin_matrix.py:36-43: row 0 is the most significant bit) and stops at the first class that contains a syndrome
// already seen, in an earlier class or earlier in the same one; that class is dropped as a whole and the classes before
// it are the table.  Inside the accepted classes every syndrome occurs once, so the result does not depend on the
// order in which a class is enumerated, and the device may enumerate it in any order:
//
//   table[key] (2^r words, all ones = empty) holds the packed error (qubit j = bit j) that produced syndrome `key`.
//   One launch per weight class.  A lane takes a run of consecutive errors in colexicographic order: it unranks its
//   first error with a binomial table in LDS (combinatorial number system) and steps to the next one with Gosper's
//   bit trick; the parity-check rows sit in SGPRs (kernel argument); the syndrome key is r AND + popcount parities; the
//   table slot is claimed with a 64-bit atomicCAS.  A failed claim raises the collision flag and the class is swept out
//   of the table again (its entries are the ones of popcount w).
#include <vector>

#include "gf2_internal.h"

#define TBL_MAX_R 24
#define TBL_RUN 64                             // consecutive errors per lane
#define TBL_EMPTY (~0ull)

struct TableRows {
    u64 row[TBL_MAX_R];
};

// binom[c * 65 + k] = C(c, k) for c, k <= 64 (saturated at 2^63)
__global__ __launch_bounds__(256) void table_class_kernel(TableRows h, int r, int n, int w, u64 total,
                                                          const u64* __restrict__ binom, u64* __restrict__ table,
                                                          int* __restrict__ collide) {
    __shared__ u64 cw[65];                                          // C(c, k) for the k this lane is unranking
    const u64 lane_first = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * TBL_RUN;
    // unrank lane_first in the combinatorial number system: for k = w .. 1 take the largest c with C(c, k) <= rank
    u64 e = 0, rank = lane_first;
    for (int k = w; k >= 1; --k) {
        __syncthreads();
        if (threadIdx.x < 65) cw[threadIdx.x] = binom[threadIdx.x * 65 + k];
        __syncthreads();
        if (lane_first < total) {
            int c = k - 1;                                          // C(k - 1, k) = 0 <= rank always
            while (c + 1 < n && cw[c + 1] <= rank) ++c;
            e |= 1ull << c;
            rank -= cw[c];
        }
    }
    if (lane_first >= total) return;
    const u64 limit = n < 64 ? (1ull << n) : 0ull;                  // 0: no limit below 2^64
    u64 left = total - lane_first < TBL_RUN ? total - lane_first : TBL_RUN;
    for (; left; --left) {
        if (*reinterpret_cast<volatile int*>(collide)) return;
        u64 key = 0;
#pragma unroll
        for (int i = 0; i < TBL_MAX_R; ++i)
            if (i < r) key = (key << 1) | (u64)(__popcll(h.row[i] & e) & 1);
        if (atomicCAS(&table[key], TBL_EMPTY, e) != TBL_EMPTY) {
            atomicExch(collide, 1);
            return;
        }
        if (left > 1) {                                             // Gosper: next word with the same popcount
            const u64 lowest = e & (0ull - e);
            const u64 ripple = e + lowest;
            e = (((ripple ^ e) >> 2) >> (__ffsll((long long)lowest) - 1)) | ripple;
            if (limit && e >= limit) return;                        // cannot happen within `total`; guards the table
        }
    }
}

__global__ void table_sweep_kernel(u64* __restrict__ table, u64 entries, int w) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < entries && table[i] != TBL_EMPTY && __popcll(table[i]) == w) table[i] = TBL_EMPTY;
}

__global__ void table_fill_kernel(u64* __restrict__ table, u64 entries) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < entries) table[i] = TBL_EMPTY;
}

// ---- 64 < n <= 128 -------------------------------------------------------------------------------------------------------
// The same search with two-word errors.  A table slot cannot hold a 128-bit error and be claimed by one atomic, so it holds
// (weight << 32) | rank, the error's rank inside its weight class in the combinatorial number system: a class with more than
// 2^24 errors cannot fit a table of at most 2^24 syndromes (pigeonhole, checked on the host), so every rank that is ever
// written fits 32 bits.  The host unranks the entries it reads back.
typedef unsigned __int128 u128;

struct TableRowsWide {
    u64 lo[TBL_MAX_R], hi[TBL_MAX_R];
};

__global__ __launch_bounds__(256) void table_class_wide_kernel(TableRowsWide h, int r, int n, int w, u64 total,
                                                               const u64* __restrict__ binom, u64* __restrict__ table,
                                                               int* __restrict__ collide) {
    __shared__ u64 cw[129];                                         // C(c, k), c <= 128, for the k being unranked
    const u64 lane_first = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * TBL_RUN;
    u128 e = 0;
    u64 rank = lane_first;
    for (int k = w; k >= 1; --k) {
        __syncthreads();
        if (threadIdx.x < 129) cw[threadIdx.x] = binom[threadIdx.x * 129 + k];
        __syncthreads();
        if (lane_first < total) {
            int c = k - 1;
            while (c + 1 < n && cw[c + 1] <= rank) ++c;
            e |= (u128)1 << c;
            rank -= cw[c];
        }
    }
    if (lane_first >= total) return;
    u64 left = total - lane_first < TBL_RUN ? total - lane_first : TBL_RUN;
    u64 my_rank = lane_first;
    for (; left; --left, ++my_rank) {
        if (*reinterpret_cast<volatile int*>(collide)) return;
        const u64 elo = (u64)e, ehi = (u64)(e >> 64);
        u64 key = 0;
#pragma unroll
        for (int i = 0; i < TBL_MAX_R; ++i)
            if (i < r) key = (key << 1) | (u64)((__popcll(h.lo[i] & elo) + __popcll(h.hi[i] & ehi)) & 1);
        if (atomicCAS(&table[key], TBL_EMPTY, ((u64)w << 32) | my_rank) != TBL_EMPTY) {
            atomicExch(collide, 1);
            return;
        }
        if (left > 1) {                                             // Gosper on 128 bits: next word with the same popcount
            const u128 lowest = e & (~e + 1);
            const u128 ripple = e + lowest;
            const int tz = (u64)lowest ? __ffsll((long long)(u64)lowest) - 1 : 64 + __ffsll((long long)(u64)(lowest >> 64)) - 1;
            e = (((ripple ^ e) >> 2) >> tz) | ripple;
        }
    }
}

__global__ void table_sweep_wide_kernel(u64* __restrict__ table, u64 entries, int w) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < entries && table[i] != TBL_EMPTY && (int)(table[i] >> 32) == w) table[i] = TBL_EMPTY;
}

extern "C" int gf2_syndrome_table_wide(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t max_weight,
                                       uint64_t* table_out, int64_t* t_out, int64_t* entries_out) {
    if (!ctx || !table_out || !t_out) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_wide: null argument");
    if (r < 0 || r > TBL_MAX_R || n < 0 || n > 128) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_wide: needs n <= 128 and r <= %d", TBL_MAX_R);
    if (r > 0 && !h_rows) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_wide: null rows");
    GF2_TRY(gf2_ctx_activate(ctx));
    TableRowsWide rows;
    for (int i = 0; i < TBL_MAX_R; ++i) {
        rows.lo[i] = i < r ? h_rows[2 * i] : 0ull;
        rows.hi[i] = i < r ? h_rows[2 * i + 1] : 0ull;
    }
    std::vector<u64> binom_vec(129 * 129);                          // Pascal's triangle, saturated just above 2^63
    u64* const binom_host = binom_vec.data();
    for (int c = 0; c <= 128; ++c)
        for (int k = 0; k <= 128; ++k) {
            u64 v;
            if (k == 0)
                v = 1;
            else if (c == 0)
                v = 0;
            else {
                const u64 a = binom_host[(c - 1) * 129 + k - 1], b = binom_host[(c - 1) * 129 + k];
                v = (a > (1ull << 63) || b > (1ull << 63) || a + b > (1ull << 63)) ? (1ull << 63) + 1 : a + b;
            }
            binom_host[c * 129 + k] = v;
        }
    const size_t binom_bytes = binom_vec.size() * sizeof(u64);
    const u64 entries = 1ull << r;
    u64 *table_dev = nullptr, *binom_dev = nullptr;
    int* collide_dev = nullptr;
    GF2_TRY(gf2_dev_alloc(ctx, entries * 8, (void**)&table_dev));
    int rc = gf2_dev_alloc(ctx, binom_bytes, (void**)&binom_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 4, (void**)&collide_dev);
    int64_t t = n, kept = 0;
    if (rc == GF2_OK) {
        hipLaunchKernelGGL(table_fill_kernel, dim3((unsigned)gf2_cdiv((int64_t)entries, 256)), dim3(256), 0, ctx->stream,
                           table_dev, entries);
        if (hipMemcpyAsync(binom_dev, binom_host, binom_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemsetAsync(collide_dev, 0, 4, ctx->stream) != hipSuccess)
            rc = GF2_E_HIP;
    }
    for (int64_t w = 0; rc == GF2_OK && w <= n; ++w) {
        if (max_weight >= 0 && w > max_weight) {
            t = max_weight;
            break;
        }
        const u64 total = binom_host[n * 129 + w];
        bool collided = total > entries - (u64)kept;                // pigeonhole: more errors than free syndromes
        if (!collided) {
            const u64 lanes = (total + TBL_RUN - 1) / TBL_RUN;
            hipLaunchKernelGGL(table_class_wide_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, rows,
                               (int)r, (int)n, (int)w, total, binom_dev, table_dev, collide_dev);
            int flag = 0;
            if (hipGetLastError() != hipSuccess ||
                hipMemcpyAsync(&flag, collide_dev, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                gf2_set_error("gf2_syndrome_table_wide: weight class %lld failed on the device", (long long)w);
                rc = GF2_E_HIP;
                break;
            }
            collided = flag != 0;
            if (collided)
                hipLaunchKernelGGL(table_sweep_wide_kernel, dim3((unsigned)gf2_cdiv((int64_t)entries, 256)), dim3(256), 0,
                                   ctx->stream, table_dev, entries, (int)w);
        }
        if (collided) {
            t = w - 1;
            break;
        }
        kept += (int64_t)total;
    }
    if (rc == GF2_OK) {
        if (hipMemcpyAsync(table_out, table_dev, entries * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            gf2_set_error("gf2_syndrome_table_wide: copying the table back failed");
            rc = GF2_E_HIP;
        }
    }
    (void)gf2_dev_free(ctx, table_dev);
    (void)gf2_dev_free(ctx, binom_dev);
    (void)gf2_dev_free(ctx, collide_dev);
    if (rc != GF2_OK) return rc;
    *t_out = t;
    if (entries_out) *entries_out = kept;
    return GF2_OK;
}

// ---- n > 128 --------------------------------------------------------------------------------------------------------------
// Errors as POSITIONS.  A class that is enumerated at all has at most 2^r <= 2^24 members (pigeonhole, as above), so beyond
// 128 bits its weight is small: C(129, 5) > 2^24 already, and w <= TBL_COLS_MAX_W = 8 covers every n.  The key of an error is the
// XOR of its columns' keys (the syndrome is linear: vec_to_int(H e_j) for column j, a table of n words in LDS), so neither the
// rows nor a packed error are needed.  A lane unranks its first error (binary search in column k of Pascal's triangle) and steps
// through TBL_RUN successors in colexicographic order: the lowest position that can move does, the ones below it fall back to
// 0, 1, 2, ...  Table slots as in the two-word kernel: (weight << 32) | rank, unranked on the host.
#define TBL_COLS_MAX_N 8192
#define TBL_COLS_MAX_W 8

__global__ __launch_bounds__(256) void table_class_cols_kernel(const unsigned int* __restrict__ colkey, int n, int w, u64 total,
                                                               const u64* __restrict__ binom, u64* __restrict__ table,
                                                               int* __restrict__ collide) {
    __shared__ unsigned int ck[TBL_COLS_MAX_N];
    for (int j = threadIdx.x; j < n; j += blockDim.x) ck[j] = colkey[j];
    __syncthreads();
    const u64 lane_first = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * TBL_RUN;
    if (lane_first >= total) return;
    int c[TBL_COLS_MAX_W];
    u64 rank = lane_first;
#pragma unroll
    for (int k = TBL_COLS_MAX_W; k >= 1; --k) {
        if (k > w) continue;                                        // uniform
        // binom[cc * (TBL_COLS_MAX_W + 1) + k] = C(cc, k), ascending in cc: the largest cc < n with C(cc, k) <= rank
        int lo = k - 1, hi = n - 1;                                 // C(k - 1, k) = 0 <= rank always
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (binom[(u64)mid * (TBL_COLS_MAX_W + 1) + k] <= rank)
                lo = mid;
            else
                hi = mid - 1;
        }
        c[k - 1] = lo;
        rank -= binom[(u64)lo * (TBL_COLS_MAX_W + 1) + k];
    }
    u64 left = total - lane_first < TBL_RUN ? total - lane_first : TBL_RUN;
    u64 my_rank = lane_first;
    for (; left; --left, ++my_rank) {
        if (*reinterpret_cast<volatile int*>(collide)) return;
        unsigned int key = 0;
#pragma unroll
        for (int k = 0; k < TBL_COLS_MAX_W; ++k)
            if (k < w) key ^= ck[c[k]];
        if (atomicCAS(&table[key], TBL_EMPTY, ((u64)w << 32) | my_rank) != TBL_EMPTY) {
            atomicExch(collide, 1);
            return;
        }
        if (left > 1) {
            // colexicographic successor: position i moves up when the one above it leaves room (the top one always may:
            // my_rank + 1 < total); registers are indexed by constants only, hence the flags instead of a search loop
            bool moved = false;
#pragma unroll
            for (int i = 0; i < TBL_COLS_MAX_W; ++i) {
                if (i >= w || moved) continue;
                const bool room = i + 1 >= w || c[i] + 1 < c[i + 1];
                if (room) {
                    c[i] += 1;
                    moved = true;
                } else
                    c[i] = i;                                       // falls back; a lower one never blocks a higher one again
            }
        }
    }
}

extern "C" int gf2_syndrome_table_cols(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t ld, int64_t max_weight,
                                       uint64_t* table_out, int64_t* t_out, int64_t* entries_out) {
    if (!ctx || !table_out || !t_out) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_cols: null argument");
    if (r < 0 || r > TBL_MAX_R || n < 0 || n > TBL_COLS_MAX_N || ld < gf2_words(n))
        GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_cols: needs n <= %d and r <= %d", TBL_COLS_MAX_N, TBL_MAX_R);
    if (r > 0 && n > 0 && !h_rows) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_cols: null rows");
    GF2_TRY(gf2_ctx_activate(ctx));
    // column keys: row 0 is the most significant bit (bin_matrix.py:36-43)
    std::vector<unsigned int> colkey((size_t)(n > 0 ? n : 1), 0u);
    for (int64_t i = 0; i < r; ++i)
        for (int64_t j = 0; j < n; ++j)
            colkey[(size_t)j] |= (unsigned int)((h_rows[i * ld + (j >> 6)] >> (j & 63)) & 1ull) << (r - 1 - i);
    const int kw = TBL_COLS_MAX_W + 1;
    std::vector<u64> binom_vec((size_t)(n + 1) * kw);               // C(c, k), k <= 8: no saturation needed below 2^63 (C(8192, 8) < 2^90: saturate)
    for (int64_t c = 0; c <= n; ++c)
        for (int k = 0; k < kw; ++k) {
            u64 v;
            if (k == 0)
                v = 1;
            else if (c == 0)
                v = 0;
            else {
                const u64 a = binom_vec[(size_t)(c - 1) * kw + k - 1], b = binom_vec[(size_t)(c - 1) * kw + k];
                v = (a > (1ull << 63) || b > (1ull << 63) || a + b > (1ull << 63)) ? (1ull << 63) + 1 : a + b;
            }
            binom_vec[(size_t)c * kw + k] = v;
        }
    const u64 entries = 1ull << r;
    u64 *table_dev = nullptr, *binom_dev = nullptr;
    unsigned int* colkey_dev = nullptr;
    int* collide_dev = nullptr;
    GF2_TRY(gf2_dev_alloc(ctx, entries * 8, (void**)&table_dev));
    int rc = gf2_dev_alloc(ctx, binom_vec.size() * 8, (void**)&binom_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, colkey.size() * 4, (void**)&colkey_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 4, (void**)&collide_dev);
    int64_t t = n, kept = 0;
    if (rc == GF2_OK) {
        hipLaunchKernelGGL(table_fill_kernel, dim3((unsigned)gf2_cdiv((int64_t)entries, 256)), dim3(256), 0, ctx->stream,
                           table_dev, entries);
        if (hipMemcpyAsync(binom_dev, binom_vec.data(), binom_vec.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemcpyAsync(colkey_dev, colkey.data(), colkey.size() * 4, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemsetAsync(collide_dev, 0, 4, ctx->stream) != hipSuccess)
            rc = GF2_E_HIP;
    }
    for (int64_t w = 0; rc == GF2_OK && w <= n; ++w) {
        if (max_weight >= 0 && w > max_weight) {
            t = max_weight;
            break;
        }
        // C(n, w) beyond the triangle's eight columns is beyond 2^24 for every n > 8 (and saturated entries compare as huge)
        const u64 total = w < kw ? binom_vec[(size_t)n * kw + w] : ~0ull;
        bool collided = total > entries - (u64)kept;                // pigeonhole: more errors than free syndromes
        if (!collided) {
            const u64 lanes = (total + TBL_RUN - 1) / TBL_RUN;
            hipLaunchKernelGGL(table_class_cols_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream,
                               (const unsigned int*)colkey_dev, (int)n, (int)w, total, (const u64*)binom_dev, table_dev, collide_dev);
            int flag = 0;
            if (hipGetLastError() != hipSuccess ||
                hipMemcpyAsync(&flag, collide_dev, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                gf2_set_error("gf2_syndrome_table_cols: weight class %lld failed on the device", (long long)w);
                rc = GF2_E_HIP;
                break;
            }
            collided = flag != 0;
            if (collided)
                hipLaunchKernelGGL(table_sweep_wide_kernel, dim3((unsigned)gf2_cdiv((int64_t)entries, 256)), dim3(256), 0,
                                   ctx->stream, table_dev, entries, (int)w);
        }
        if (collided) {
            t = w - 1;
            break;
        }
        kept += (int64_t)total;
    }
    if (rc == GF2_OK) {
        if (hipMemcpyAsync(table_out, table_dev, entries * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            gf2_set_error("gf2_syndrome_table_cols: copying the table back failed");
            rc = GF2_E_HIP;
        }
    }
    (void)gf2_dev_free(ctx, table_dev);
    (void)gf2_dev_free(ctx, binom_dev);
    (void)gf2_dev_free(ctx, colkey_dev);
    (void)gf2_dev_free(ctx, collide_dev);
    if (rc != GF2_OK) return rc;
    *t_out = t;
    if (entries_out) *entries_out = kept;
    return GF2_OK;
}

extern "C" int gf2_syndrome_table(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t max_weight,
                                  uint64_t* table_out, int64_t* t_out, int64_t* entries_out) {
    if (!ctx || !table_out || !t_out) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table: null argument");
    if (r < 0 || r > TBL_MAX_R || n < 0 || n > 64) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table: needs n <= 64 and r <= %d", TBL_MAX_R);
    if (r > 0 && !h_rows) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table: null rows");
    GF2_TRY(gf2_ctx_activate(ctx));
    TableRows rows;
    for (int i = 0; i < TBL_MAX_R; ++i) rows.row[i] = i < r ? h_rows[i] : 0ull;
    // Pascal's triangle, saturated (on the heap: callers may run on several threads, and 33 KiB is a lot of stack)
    std::vector<u64> binom_vec(65 * 65);
    u64* const binom_host = binom_vec.data();
    const size_t binom_bytes = binom_vec.size() * sizeof(u64);
    for (int c = 0; c <= 64; ++c)
        for (int k = 0; k <= 64; ++k) {
            u64 v;
            if (k == 0)
                v = 1;
            else if (c == 0)
                v = 0;
            else {
                const u64 a = binom_host[(c - 1) * 65 + k - 1], b = binom_host[(c - 1) * 65 + k];
                v = (a > (1ull << 63) || b > (1ull << 63) || a + b > (1ull << 63)) ? (1ull << 63) + 1 : a + b;
            }
            binom_host[c * 65 + k] = v;
        }
    const u64 entries = 1ull << r;
    u64 *table_dev = nullptr, *binom_dev = nullptr;
    int* collide_dev = nullptr;
    GF2_TRY(gf2_dev_alloc(ctx, entries * 8, (void**)&table_dev));
    int rc = gf2_dev_alloc(ctx, binom_bytes, (void**)&binom_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 4, (void**)&collide_dev);
    int64_t t = n, kept = 0;
    if (rc == GF2_OK) {
        hipLaunchKernelGGL(table_fill_kernel, dim3((unsigned)gf2_cdiv((int64_t)entries, 256)), dim3(256), 0, ctx->stream,
                           table_dev, entries);
        if (hipMemcpyAsync(binom_dev, binom_host, binom_bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
            hipMemsetAsync(collide_dev, 0, 4, ctx->stream) != hipSuccess)
            rc = GF2_E_HIP;
    }
    for (int64_t w = 0; rc == GF2_OK && w <= n; ++w) {
        if (max_weight >= 0 && w > max_weight) {
            t = max_weight;
            break;
        }
        const u64 total = binom_host[n * 65 + w];
        bool collided = total > entries - (u64)kept;                // pigeonhole: more errors than free syndromes
        if (!collided) {
            const u64 lanes = (total + TBL_RUN - 1) / TBL_RUN;
            hipLaunchKernelGGL(table_class_kernel, dim3((unsigned)((lanes + 255) / 256)), dim3(256), 0, ctx->stream, rows,
                               (int)r, (int)n, (int)w, total, binom_dev, table_dev, collide_dev);
            int flag = 0;
            if (hipGetLastError() != hipSuccess ||
                hipMemcpyAsync(&flag, collide_dev, 4, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                hipStreamSynchronize(ctx->stream) != hipSuccess) {
                gf2_set_error("gf2_syndrome_table: weight class %lld failed on the device", (long long)w);
                rc = GF2_E_HIP;
                break;
            }
            collided = flag != 0;
            if (collided)
                hipLaunchKernelGGL(table_sweep_kernel, dim3((unsigned)gf2_cdiv((int64_t)entries, 256)), dim3(256), 0,
                                   ctx->stream, table_dev, entries, (int)w);
        }
        if (collided) {
            t = w - 1;
            break;
        }
        kept += (int64_t)total;
    }
    if (rc == GF2_OK) {
        if (hipMemcpyAsync(table_out, table_dev, entries * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            gf2_set_error("gf2_syndrome_table: copying the table back failed");
            rc = GF2_E_HIP;
        }
    }
    (void)gf2_dev_free(ctx, table_dev);
    (void)gf2_dev_free(ctx, binom_dev);
    (void)gf2_dev_free(ctx, collide_dev);
    if (rc != GF2_OK) return rc;
    *t_out = t;
    if (entries_out) *entries_out = kept;
    return GF2_OK;
}

// ---- r > 24: hashed tables -------------------------------------------------------------------------------------------------
// A dense table of 2^r slots stops at r = 24.  Beyond it the same search runs against an open-addressing hash table sized from
// the classes it has to hold (sum of C(n, w)), so what bounds it is the number of errors enumerated, not r: syndrome keys of up
// to 63 bits are one word, of up to 127 bits two (css_code.py:729 forms them with vec_to_int, row 0 most significant; the
// reference's own keys wrap beyond 63 bits, SURVEY.md 7.3 item 2 -- these are exact).  Errors are position lists keyed by the
// XOR of their columns' keys, as in table_class_cols_kernel.
//
// Slot = claim word + (two-word keys) second key word + value (weight << 32 | rank in the class, as above).  The claim word is
// the key itself (one word: below 2^63) or its HIGH word (two words: below 2^63 as well), never all ones, which means "empty";
// it is taken with one atomicCAS.  Linear probing; the table is kept at most half full.  A probe that meets its own key has
// found a second error with that syndrome: the collision the reference's `if syndrome_int in table` (css_code.py:730) reports.
// Two-word keys: the owner of a slot writes the low word and then the value (all ones until then); a probe that meets its
// high word waits for the value before it compares the low words.  Owners publish before anybody of their wavefront waits
// (claim, publish and compare are three phases of a probe step, not branches of one if), so a wait is only ever for another
// wavefront, and it is bounded all the same.
#define TBL_HASH_MAX_W 12
#define TBL_HASH_MAX_N 8192
#define TBL_HASH_MAX_ENTRIES (1ull << 28)     // errors enumerated in all; the table has twice as many slots (12 or 16 bytes each... 8 + 8 [+ 8])
#define TBL_HASH_SPIN (1u << 22)

struct HashTab {
    u64* claim;                  // slots
    u64* low;                    // slots (two-word keys only)
    u64* val;                    // slots
    u64 mask;                    // slots - 1
};

__device__ __forceinline__ u64 hash_mix(u64 z) {
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

// Inserts (key -> v).  Returns 0: inserted, 1: the key is there already (collision), 2: gave up (a slot's owner did not
// publish in time, or the table is full: both reported as an error by the caller).  `active`: lanes with nothing to insert
// still walk through the phases.
template <int KW>
__device__ __forceinline__ int hash_insert(const HashTab& t, u64 khi, u64 klo, u64 v, bool active) {
    const u64 claim_word = KW == 2 ? khi : klo;
    u64 slot = hash_mix(klo ^ (KW == 2 ? hash_mix(khi + 0x9E3779B97F4A7C15ull) : 0ull)) & t.mask;
    int result = active ? -1 : 0;
    for (u64 probes = 0; __ballot(result < 0) != 0; ++probes) {
        if (probes > t.mask) {                                     // (cannot happen below half load)
            if (result < 0) result = 2;
            break;
        }
        // claim
        u64 old = 0;
        if (result < 0) old = atomicCAS(&t.claim[slot], TBL_EMPTY, claim_word);
        // publish
        if (result < 0 && old == TBL_EMPTY) {
            if (KW == 2) {
                __hip_atomic_store(&t.low[slot], klo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __atomic_thread_fence(__ATOMIC_RELEASE);
            }
            __hip_atomic_store(&t.val[slot], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            result = 0;
        }
        // compare
        if (result < 0 && old == claim_word) {
            if (KW == 1)
                result = 1;
            else {
                unsigned int spins = 0;
                while (__hip_atomic_load(&t.val[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == TBL_EMPTY && spins < TBL_HASH_SPIN) {
                    __builtin_amdgcn_s_sleep(2);
                    ++spins;
                }
                __atomic_thread_fence(__ATOMIC_ACQUIRE);
                if (spins >= TBL_HASH_SPIN)
                    result = 2;
                else if (__hip_atomic_load(&t.low[slot], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == klo)
                    result = 1;
            }
        }
        slot = (slot + 1) & t.mask;
    }
    return result;
}

// Looks `key` up (table complete, nobody writing).  Returns the slot or ~0.
template <int KW>
__device__ __forceinline__ u64 hash_find(const HashTab& t, u64 khi, u64 klo) {
    const u64 claim_word = KW == 2 ? khi : klo;
    u64 slot = hash_mix(klo ^ (KW == 2 ? hash_mix(khi + 0x9E3779B97F4A7C15ull) : 0ull)) & t.mask;
    for (u64 probes = 0; probes <= t.mask; ++probes) {
        const u64 c = t.claim[slot];
        if (c == TBL_EMPTY) return ~0ull;
        if (c == claim_word && (KW == 1 || t.low[slot] == klo)) return slot;
        slot = (slot + 1) & t.mask;
    }
    return ~0ull;
}

// One weight class.  colkey: n x KW words (word 0 = low).  flags[0]: collision, flags[1]: error (see hash_insert).
template <int KW>
__global__ __launch_bounds__(256) void table_class_hash_kernel(const u64* __restrict__ colkey, int n, int w, u64 total,
                                                               const u64* __restrict__ binom, HashTab tab, int* __restrict__ flags) {
    extern __shared__ u64 ck[];                                     // n x KW
    for (int j = threadIdx.x; j < n * KW; j += blockDim.x) ck[j] = colkey[j];
    __syncthreads();
    const u64 lane_first = ((u64)blockIdx.x * blockDim.x + threadIdx.x) * TBL_RUN;
    const bool live = lane_first < total;
    int c[TBL_HASH_MAX_W];
#pragma unroll
    for (int k = 0; k < TBL_HASH_MAX_W; ++k) c[k] = k;
    if (live) {
        u64 rank = lane_first;
#pragma unroll
        for (int k = TBL_HASH_MAX_W; k >= 1; --k) {
            if (k > w) continue;                                    // uniform
            int lo = k - 1, hi = n - 1;                             // the largest cc < n with C(cc, k) <= rank
            while (lo < hi) {
                const int mid = (lo + hi + 1) >> 1;
                if (binom[(u64)mid * (TBL_HASH_MAX_W + 1) + k] <= rank)
                    lo = mid;
                else
                    hi = mid - 1;
            }
            c[k - 1] = lo;
            rank -= binom[(u64)lo * (TBL_HASH_MAX_W + 1) + k];
        }
    }
    u64 left = live ? (total - lane_first < TBL_RUN ? total - lane_first : TBL_RUN) : 0;
    u64 my_rank = lane_first;
    // (all lanes of a wavefront go round together: hash_insert's phases are wave-wide)
    for (int it = 0; it < TBL_RUN; ++it, ++my_rank) {
        if (__ballot(left != 0) == 0) break;
        if (*reinterpret_cast<volatile int*>(flags) | *reinterpret_cast<volatile int*>(flags + 1)) return;
        u64 klo = 0, khi = 0;
#pragma unroll
        for (int k = 0; k < TBL_HASH_MAX_W; ++k)
            if (k < w) {
                klo ^= ck[c[k] * KW];
                if (KW == 2) khi ^= ck[c[k] * KW + 1];
            }
        const int res = hash_insert<KW>(tab, khi, klo, ((u64)w << 32) | my_rank, left != 0);
        if (res == 1) atomicExch(&flags[0], 1);
        if (res == 2) atomicExch(&flags[1], 1);
        if (left > 1) {                                             // colexicographic successor (table_class_cols_kernel)
            bool moved = false;
#pragma unroll
            for (int i = 0; i < TBL_HASH_MAX_W; ++i) {
                if (i >= w || moved) continue;
                const bool room = i + 1 >= w || c[i] + 1 < c[i + 1];
                if (room) {
                    c[i] += 1;
                    moved = true;
                } else
                    c[i] = i;
            }
        }
        if (left) left -= 1;
    }
}

__global__ void hash_fill_kernel(HashTab tab, int kw) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i <= tab.mask) {
        tab.claim[i] = TBL_EMPTY;
        tab.val[i] = TBL_EMPTY;
        if (kw == 2) tab.low[i] = 0;
    }
}

// The entries of weight <= t, in slot order: keys_out KW words each (word 0 = low), vals_out; *count_out = how many there are
// (the host sized the outputs from the class sizes).
template <int KW>
__global__ void hash_extract_kernel(HashTab tab, int t, u64* __restrict__ keys_out, u64* __restrict__ vals_out, u64 cap,
                                    unsigned long long* __restrict__ count_out) {
    const u64 i = (u64)blockIdx.x * blockDim.x + threadIdx.x;
    if (i > tab.mask) return;
    const u64 c = tab.claim[i];
    if (c == TBL_EMPTY) return;
    const u64 v = tab.val[i];
    if ((int64_t)(v >> 32) > (int64_t)t) return;
    const u64 at = atomicAdd(count_out, 1ull);
    if (at >= cap) return;
    if (KW == 1)
        keys_out[at] = c;
    else {
        keys_out[2 * at] = tab.low[i];
        keys_out[2 * at + 1] = c;
    }
    vals_out[at] = v;
}

// Column keys of packed rows: bit (r - 1 - i) of column j's key = H[i][j] (bin_matrix.py:36-43), kw words per column, word 0 low.
static void column_keys(const uint64_t* h_rows, int64_t r, int64_t n, int64_t ld, int kw, std::vector<u64>* out) {
    out->assign((size_t)(n > 0 ? n : 1) * kw, 0ull);
    for (int64_t i = 0; i < r; ++i) {
        const int64_t bit = r - 1 - i;
        for (int64_t j = 0; j < n; ++j)
            if ((h_rows[i * ld + (j >> 6)] >> (j & 63)) & 1ull) (*out)[(size_t)j * kw + (bit >> 6)] |= 1ull << (bit & 63);
    }
}

// (per context, like the other large-LDS opt-ins: a context is used by one thread at a time, distinct contexts may run on distinct
// threads -- a process-wide table of devices would be written by several)
static int hash_lds_optin(gf2_ctx* ctx) {
    if (ctx->lds_optin[1]) return GF2_OK;
    GF2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(table_class_hash_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TBL_HASH_MAX_N * 8));
    GF2_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(table_class_hash_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                TBL_HASH_MAX_N * 16));
    ctx->lds_optin[1] = true;
    return GF2_OK;
}

struct HashAlloc {
    gf2_ctx* ctx;
    HashTab tab;
    u64 slots;
    HashAlloc(gf2_ctx* c) : ctx(c), tab{nullptr, nullptr, nullptr, 0}, slots(0) {}
    void release() {
        if (tab.claim) (void)gf2_dev_free(ctx, tab.claim);
        if (tab.low) (void)gf2_dev_free(ctx, tab.low);
        if (tab.val) (void)gf2_dev_free(ctx, tab.val);
        tab.claim = tab.low = tab.val = nullptr;
        slots = 0;
    }
    int make(u64 want_slots, int kw) {
        release();
        GF2_TRY(gf2_dev_alloc(ctx, want_slots * 8, (void**)&tab.claim));
        GF2_TRY(gf2_dev_alloc(ctx, want_slots * 8, (void**)&tab.val));
        if (kw == 2) GF2_TRY(gf2_dev_alloc(ctx, want_slots * 8, (void**)&tab.low));
        tab.mask = want_slots - 1;
        slots = want_slots;
        hipLaunchKernelGGL(hash_fill_kernel, dim3((unsigned)((want_slots + 255) / 256)), dim3(256), 0, ctx->stream, tab, kw);
        GF2_HIP(hipGetLastError());
        return GF2_OK;
    }
    ~HashAlloc() { release(); }
};

static u64 pow2_at_least(u64 v) {
    u64 p = 1024;
    while (p < v) p <<= 1;
    return p;
}

extern "C" int gf2_syndrome_table_hashed(gf2_ctx* ctx, const uint64_t* h_rows, int64_t r, int64_t n, int64_t ld, int64_t max_weight,
                                         uint64_t* keys_out, uint64_t* vals_out, int64_t capacity, int64_t* t_out,
                                         int64_t* entries_out) {
    if (!ctx || !t_out || !entries_out) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_hashed: null argument");
    if (r < 1 || r > 127 || n < 1 || n > TBL_HASH_MAX_N || ld < gf2_words(n))
        GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_hashed: needs 1 <= r <= 127 and 1 <= n <= %d", TBL_HASH_MAX_N);
    if (!h_rows || capacity < 0 || (capacity > 0 && (!keys_out || !vals_out))) GF2_FAIL(GF2_E_ARG, "gf2_syndrome_table_hashed: null buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_TRY(hash_lds_optin(ctx));
    const int kw = r <= 63 ? 1 : 2;
    std::vector<u64> colkey;
    column_keys(h_rows, r, n, ld, kw, &colkey);
    const int kcols = TBL_HASH_MAX_W + 1;
    std::vector<u64> binom_vec((size_t)(n + 1) * kcols);            // C(c, k), k <= 12, saturated just above 2^63
    for (int64_t c = 0; c <= n; ++c)
        for (int k = 0; k < kcols; ++k) {
            u64 v;
            if (k == 0)
                v = 1;
            else if (c == 0)
                v = 0;
            else {
                const u64 a = binom_vec[(size_t)(c - 1) * kcols + k - 1], b = binom_vec[(size_t)(c - 1) * kcols + k];
                v = (a > (1ull << 63) || b > (1ull << 63) || a + b > (1ull << 63)) ? (1ull << 63) + 1 : a + b;
            }
            binom_vec[(size_t)c * kcols + k] = v;
        }
    auto class_size = [&](int64_t w) -> u64 { return w < kcols ? binom_vec[(size_t)n * kcols + w] : ~0ull; };
    u64 *binom_dev = nullptr, *colkey_dev = nullptr, *keys_dev = nullptr, *vals_dev = nullptr;
    int* flags_dev = nullptr;
    unsigned long long* count_dev = nullptr;
    HashAlloc table(ctx);
    int rc = gf2_dev_alloc(ctx, binom_vec.size() * 8, (void**)&binom_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, colkey.size() * 8, (void**)&colkey_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 16, (void**)&flags_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 8, (void**)&count_dev);
    if (rc == GF2_OK && (hipMemcpyAsync(binom_dev, binom_vec.data(), binom_vec.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                         hipMemcpyAsync(colkey_dev, colkey.data(), colkey.size() * 8, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
                         hipMemsetAsync(flags_dev, 0, 16, ctx->stream) != hipSuccess))
        rc = GF2_E_HIP;
    auto run_class = [&](int64_t w, int* flags_host) -> int {       // enumerates class w into the table
        const u64 total = class_size(w);
        const u64 lanes = (total + TBL_RUN - 1) / TBL_RUN;
        const dim3 grid((unsigned)((lanes + 255) / 256)), block(256);
        if (kw == 1)
            hipLaunchKernelGGL((table_class_hash_kernel<1>), grid, block, (size_t)n * 8, ctx->stream, (const u64*)colkey_dev, (int)n, (int)w,
                               total, (const u64*)binom_dev, table.tab, flags_dev);
        else
            hipLaunchKernelGGL((table_class_hash_kernel<2>), grid, block, (size_t)n * 16, ctx->stream, (const u64*)colkey_dev, (int)n, (int)w,
                               total, (const u64*)binom_dev, table.tab, flags_dev);
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(flags_host, flags_dev, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
            hipStreamSynchronize(ctx->stream) != hipSuccess) {
            gf2_set_error("gf2_syndrome_table_hashed: weight class %lld failed on the device", (long long)w);
            return GF2_E_HIP;
        }
        if (flags_host[1]) {
            gf2_set_error("gf2_syndrome_table_hashed: the hash table gave up in weight class %lld", (long long)w);
            return GF2_E_HIP;
        }
        return GF2_OK;
    };
    int64_t t = n;
    u64 kept = 0;
    for (int64_t w = 0; rc == GF2_OK && w <= n; ++w) {
        if (max_weight >= 0 && w > max_weight) {
            t = max_weight;
            break;
        }
        const u64 total = class_size(w);
        if (w >= kcols || total > TBL_HASH_MAX_ENTRIES || kept + total > TBL_HASH_MAX_ENTRIES) {
            gf2_set_error("gf2_syndrome_table_hashed: no collision up to weight %lld and class %lld has more than 2^28 errors "
                          "(pass max_weight to cap the search)", (long long)(w - 1), (long long)w);
            rc = GF2_E_NOMEM;
            break;
        }
        int flags_host[2] = {0, 0};
        if ((kept + total) * 2 > table.slots) {
            // a larger table, sized for this class and the next if that stays moderate, and the classes so far once more (they
            // grow geometrically: enumerating them again costs a fraction of the class to come)
            u64 want = (kept + total) * 2;
            const u64 next = class_size(w + 1);
            if (w + 1 < kcols && kept + total + next <= (1ull << 24)) want = (kept + total + next) * 2;
            rc = table.make(pow2_at_least(want), kw);
            for (int64_t v = 0; rc == GF2_OK && v < w; ++v) {
                rc = run_class(v, flags_host);
                if (rc == GF2_OK && flags_host[0]) {
                    gf2_set_error("gf2_syndrome_table_hashed: class %lld collided when enumerated again", (long long)v);
                    rc = GF2_E_HIP;
                }
            }
            if (rc != GF2_OK) break;
        }
        rc = run_class(w, flags_host);
        if (rc != GF2_OK) break;
        if (flags_host[0]) {                                        // the class goes as a whole (its entries are filtered out below)
            t = w - 1;
            break;
        }
        kept += total;
    }
    if (rc == GF2_OK) {
        *t_out = t;
        *entries_out = (int64_t)kept;
        if (capacity > 0 && (u64)capacity >= kept && kept > 0) {
            rc = gf2_dev_alloc(ctx, kept * 8 * kw, (void**)&keys_dev);
            if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, kept * 8, (void**)&vals_dev);
            unsigned long long got = 0;
            if (rc == GF2_OK) {
                if (hipMemsetAsync(count_dev, 0, 8, ctx->stream) != hipSuccess) rc = GF2_E_HIP;
                const dim3 grid((unsigned)((table.slots + 255) / 256)), block(256);
                if (kw == 1)
                    hipLaunchKernelGGL((hash_extract_kernel<1>), grid, block, 0, ctx->stream, table.tab, (int)t, keys_dev, vals_dev, kept, count_dev);
                else
                    hipLaunchKernelGGL((hash_extract_kernel<2>), grid, block, 0, ctx->stream, table.tab, (int)t, keys_dev, vals_dev, kept, count_dev);
                if (hipGetLastError() != hipSuccess ||
                    hipMemcpyAsync(&got, count_dev, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(keys_out, keys_dev, kept * 8 * kw, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipMemcpyAsync(vals_out, vals_dev, kept * 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                    hipStreamSynchronize(ctx->stream) != hipSuccess) {
                    gf2_set_error("gf2_syndrome_table_hashed: copying the entries back failed");
                    rc = GF2_E_HIP;
                } else if (got != kept) {
                    gf2_set_error("gf2_syndrome_table_hashed: %llu entries in the table, %llu expected", got, (unsigned long long)kept);
                    rc = GF2_E_HIP;
                }
            }
        }
    }
    (void)gf2_dev_free(ctx, binom_dev);
    (void)gf2_dev_free(ctx, colkey_dev);
    (void)gf2_dev_free(ctx, flags_dev);
    (void)gf2_dev_free(ctx, count_dev);
    (void)gf2_dev_free(ctx, keys_dev);
    (void)gf2_dev_free(ctx, vals_dev);
    return rc;
}

// ---- table decode + logical-error tally through hashed tables (n <= 128) ---------------------------------------------------
// The classical content of quil_classical_correct (css_code.py:649-685) and noisy_measure (css_code.py:599-646), as
// decode_kernel (gf2_mc.hip) does it for n <= 63 and dense tables of 2^r <= 2^20 words: a k = 1 CSS code has r_1 + r_2 = n - 1,
// so from n = 51 on one of its checks has more than 24 rows and its table (css_code.py:715-735) only exists hashed.
// Lane = sample: the sampler's one segment (n <= 512 qubits are one segment, gf2_sampler.h), both syndrome keys as XORs of
// column keys, two table lookups, residuals, the two logical parities; counts as gf2_mc_decode's.
#include "gf2_sampler.h"

struct DecodeHashArgs {
    HashTab tab[2];            // [0]: parity_check_c2's table (X errors), [1]: parity_check_c1's (Z errors)  (css_code.py:457-470)
    const u64* corr[2];        // corrections, two words per entry; a slot's value is the entry's index
    const u64* colkey[2];      // n x kw[c] words
    int kw[2];
    u64 op[2][2];              // [0]: z_operator (tested against the residual X error), [1]: x_operator
    int n;
    u64 seed;
    int64_t first_sample, count;
    SegTables th;
    u64* counts;
};

__global__ __launch_bounds__(256) void table_insert_kernel(HashTab tab, const u64* __restrict__ keys, int kw, int64_t entries,
                                                           int* __restrict__ flags) {
    const int64_t base = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) & ~(int64_t)63;
    if (base >= entries) return;                                    // whole wavefronts leave together
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = i < entries;
    const u64 klo = live ? keys[i * kw] : 0ull, khi = live && kw == 2 ? keys[i * kw + 1] : 0ull;
    const int res = kw == 1 ? hash_insert<1>(tab, khi, klo, (u64)i, live) : hash_insert<2>(tab, khi, klo, (u64)i, live);
    if (res == 1) atomicExch(&flags[0], 1);                         // the same key twice: not a syndrome table
    if (res == 2) atomicExch(&flags[1], 1);
}

__global__ __launch_bounds__(256) void decode_hash_kernel(DecodeHashArgs a) {
    __shared__ unsigned int acc[5];
    __shared__ u64 cdf_lds[GF2_SEG_CDF];
    __shared__ u64 ck[2][128 * 2];
    if (threadIdx.x < 5) acc[threadIdx.x] = 0;
    for (int i = threadIdx.x; i < GF2_SEG_CDF; i += blockDim.x) cdf_lds[i] = a.th.cdf[GF2_SEG_CDF + i];   // the last (= only) segment's table
    for (int c = 0; c < 2; ++c)
        for (int i = threadIdx.x; i < a.n * a.kw[c]; i += blockDim.x) ck[c][i] = a.colkey[c][i];
    __syncthreads();
    unsigned int local[5] = {0, 0, 0, 0, 0};
    const int nb = a.n;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < a.count; i += stride) {
        const u64 d = segment_draw(sample_key(a.seed, (u64)(a.first_sample + i)), 0);
        const int K = error_count(d, nb, cdf_lds);
        u64 e[2][2] = {{0, 0}, {0, 0}}, chosen[2] = {0, 0};          // e[0]: X component, e[1]: Z component
        u64 key[2][2] = {{0, 0}, {0, 0}};
        for (int k = 0; k < K; ++k) {
            unsigned int t, kind;
            error_draw(d, k, K, nb, a.th.t_1, a.th.t_2, &t, &kind);
            const unsigned int j = (unsigned int)(nb - K + k);
            const unsigned int pos = (((t >> 6 ? chosen[1] : chosen[0]) >> (t & 63u)) & 1ull) ? j : t;
            const u64 bit = 1ull << (pos & 63u);
            if (pos >> 6) chosen[1] |= bit; else chosen[0] |= bit;
#pragma unroll
            for (int c = 0; c < 2; ++c)
                if ((kind >> c) & 1u) {
                    if (pos >> 6) e[c][1] |= bit; else e[c][0] |= bit;
                    key[c][0] ^= ck[c][pos * a.kw[c]];
                    if (a.kw[c] == 2) key[c][1] ^= ck[c][pos * 2 + 1];
                }
        }
        bool flip[2], miss[2];
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const u64 slot = a.kw[c] == 1 ? hash_find<1>(a.tab[c], 0ull, key[c][0]) : hash_find<2>(a.tab[c], key[c][1], key[c][0]);
            miss[c] = slot == ~0ull;
            u64 r0 = e[c][0], r1 = e[c][1];
            if (!miss[c]) {                                         // css_code.py:655-657: no match leaves the error as it is
                const u64 idx = a.tab[c].val[slot];
                r0 ^= a.corr[c][2 * idx];
                r1 ^= a.corr[c][2 * idx + 1];
            }
            flip[c] = (__popcll(a.op[c][0] & r0) + __popcll(a.op[c][1] & r1)) & 1;
        }
        local[0] += flip[0];
        local[1] += flip[1];
        local[2] += flip[0] | flip[1];
        local[3] += miss[0];
        local[4] += miss[1];
    }
#pragma unroll
    for (int k = 0; k < 5; ++k)
        if (local[k]) atomicAdd(&acc[k], local[k]);
    __syncthreads();
    if (threadIdx.x < 5 && acc[threadIdx.x]) atomicAdd(&a.counts[threadIdx.x], (u64)acc[threadIdx.x]);
}

// h1 / h2: packed rows of the two checks (ld words each); keys: kw words per entry (kw = 1 for r <= 63, else 2; word 0 low),
// corr: two words per entry; operators: two words each.  counts_out[5] as gf2_mc_decode's.
extern "C" int gf2_mc_decode_hashed(gf2_ctx* ctx, int64_t n, int64_t ld, const uint64_t* h1, int64_t r1, const uint64_t* keys1,
                                    const uint64_t* corr1, int64_t entries1, const uint64_t* h2, int64_t r2, const uint64_t* keys2,
                                    const uint64_t* corr2, int64_t entries2, const uint64_t* x_operator, const uint64_t* z_operator,
                                    uint64_t seed, int64_t first_sample, int64_t count, double p_x, double p_y, double p_z,
                                    uint64_t* counts_out) {
    if (!ctx || !h1 || !h2 || !x_operator || !z_operator || !counts_out) GF2_FAIL(GF2_E_ARG, "gf2_mc_decode_hashed: null argument");
    if (n < 1 || n > 128 || ld < gf2_words(n) || r1 < 1 || r2 < 1 || r1 > 127 || r2 > 127)
        GF2_FAIL(GF2_E_ARG, "gf2_mc_decode_hashed: needs n <= 128 and 1 <= r_1, r_2 <= 127");
    if (entries1 < 0 || entries2 < 0 || (entries1 && (!keys1 || !corr1)) || (entries2 && (!keys2 || !corr2)))
        GF2_FAIL(GF2_E_ARG, "gf2_mc_decode_hashed: bad table");
    if (entries1 > (int64_t)TBL_HASH_MAX_ENTRIES || entries2 > (int64_t)TBL_HASH_MAX_ENTRIES)
        GF2_FAIL(GF2_E_ARG, "gf2_mc_decode_hashed: table too large");
    if (count < 0 || first_sample < 0) GF2_FAIL(GF2_E_ARG, "gf2_mc_decode_hashed: negative range");
    GF2_TRY(gf2_ctx_activate(ctx));
    for (int k = 0; k < 5; ++k) counts_out[k] = 0;
    if (count == 0) return GF2_OK;
    DecodeHashArgs a;
    GF2_TRY(gf2_seg_tables(ctx, p_x, p_y, p_z, n, &a.th));
    a.n = (int)n;
    a.seed = seed;
    a.first_sample = first_sample;
    a.count = count;
    for (int w = 0; w < 2; ++w) {
        a.op[0][w] = w < ld ? z_operator[w] : 0ull;
        a.op[1][w] = w < ld ? x_operator[w] : 0ull;
    }
    // side 0: X errors against parity_check_c2; side 1: Z errors against parity_check_c1
    const uint64_t* hs[2] = {h2, h1};
    const int64_t rs[2] = {r2, r1}, es[2] = {entries2, entries1};
    const uint64_t* ks[2] = {keys2, keys1};
    const uint64_t* cs[2] = {corr2, corr1};
    HashAlloc tabs[2] = {HashAlloc(ctx), HashAlloc(ctx)};
    u64 *dev[2][3] = {{nullptr, nullptr, nullptr}, {nullptr, nullptr, nullptr}}, *counts_dev = nullptr;
    int* flags_dev = nullptr;
    int rc = gf2_dev_alloc(ctx, 16, (void**)&flags_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 40, (void**)&counts_dev);
    if (rc == GF2_OK && (hipMemsetAsync(flags_dev, 0, 16, ctx->stream) != hipSuccess || hipMemsetAsync(counts_dev, 0, 40, ctx->stream) != hipSuccess))
        rc = GF2_E_HIP;
    for (int c = 0; c < 2 && rc == GF2_OK; ++c) {
        const int kw = rs[c] <= 63 ? 1 : 2;
        a.kw[c] = kw;
        std::vector<u64> colkey;
        column_keys(hs[c], rs[c], n, ld, kw, &colkey);
        rc = gf2_dev_alloc(ctx, colkey.size() * 8, (void**)&dev[c][0]);
        if (rc == GF2_OK) rc = gf2_h2d(ctx, dev[c][0], colkey.data(), colkey.size() * 8);
        const size_t ent = (size_t)(es[c] > 0 ? es[c] : 1);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, ent * 8 * kw, (void**)&dev[c][1]);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, ent * 16, (void**)&dev[c][2]);
        if (rc == GF2_OK && es[c]) rc = gf2_h2d(ctx, dev[c][1], ks[c], (size_t)es[c] * 8 * kw);
        if (rc == GF2_OK && es[c]) rc = gf2_h2d(ctx, dev[c][2], cs[c], (size_t)es[c] * 16);
        if (rc == GF2_OK) rc = tabs[c].make(pow2_at_least((u64)es[c] * 2 + 2), kw);
        if (rc == GF2_OK && es[c]) {
            hipLaunchKernelGGL(table_insert_kernel, dim3((unsigned)gf2_cdiv(es[c], 256)), dim3(256), 0, ctx->stream, tabs[c].tab,
                               (const u64*)dev[c][1], kw, es[c], flags_dev);
            if (hipGetLastError() != hipSuccess) rc = GF2_E_HIP;
        }
        a.tab[c] = tabs[c].tab;
        a.corr[c] = dev[c][2];
        a.colkey[c] = dev[c][0];
    }
    int flags_host[2] = {0, 0};
    if (rc == GF2_OK && (hipMemcpyAsync(flags_host, flags_dev, 8, hipMemcpyDeviceToHost, ctx->stream) != hipSuccess ||
                         hipStreamSynchronize(ctx->stream) != hipSuccess))
        rc = GF2_E_HIP;
    if (rc == GF2_OK && (flags_host[0] || flags_host[1])) {
        gf2_set_error(flags_host[0] ? "gf2_mc_decode_hashed: a syndrome key occurs twice in a table" : "gf2_mc_decode_hashed: the hash table gave up");
        rc = flags_host[0] ? GF2_E_ARG : GF2_E_HIP;
    }
    if (rc == GF2_OK) {
        a.counts = counts_dev;
        int64_t blocks = gf2_cdiv(count, 256 * 16);
        if (blocks > 4096) blocks = 4096;
        if (blocks < 1) blocks = 1;
        rc = gf2_prof_begin(ctx, GF2_K_SAMPLER);
        if (rc == GF2_OK) {
            hipLaunchKernelGGL(decode_hash_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
            rc = gf2_prof_end(ctx);
        }
        if (rc == GF2_OK && hipGetLastError() != hipSuccess) rc = GF2_E_HIP;
        if (rc == GF2_OK) rc = gf2_d2h(ctx, counts_out, counts_dev, 40);
    }
    for (int c = 0; c < 2; ++c)
        for (int k = 0; k < 3; ++k) (void)gf2_dev_free(ctx, dev[c][k]);
    (void)gf2_dev_free(ctx, flags_dev);
    (void)gf2_dev_free(ctx, counts_dev);
    return rc;
}
