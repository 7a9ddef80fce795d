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
