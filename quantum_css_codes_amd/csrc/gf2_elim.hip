// GF(2) elimination on packed words (gfx950): reduced row echelon form, the reference's order-sensitive
// parity-check normalisation, nullspace, column swap and row weights.
//
//   bin_matrix.reduced_row_echelon_form   bin_matrix.py:8-34     -> eliminate_kernel<ELIM_RREF>
//   css_code.normalize_parity_check       css_code.py:809-836    -> eliminate_kernel<ELIM_NORMALIZE>
//   css_code.swap_columns                 css_code.py:783-785    -> swap_columns_kernel
//   css_code.is_doubly_even (row sums)    css_code.py:846-850    -> row_weights_kernel
//   nullspace [build-defined, x1]                                 -> nullspace_kernel
//
// One workgroup of 1024 lanes owns one matrix (a batch of matrices is one workgroup each); the matrix
// stays in global memory and, at the sizes of interest (1 MiB packed at 2048 x 4096), in the XCD's L2.
// Pivots are processed strictly one after another.  The RREF is unique, so for it the kernel is free to
// swap rows; the normalisation is not: there the kernel performs the reference's
// operations in the reference's order -- first odd row at or below the diagonal is XOR-ed into the
// diagonal row, otherwise the first odd column of the diagonal row's current state is swapped in -- so
// the column swaps and the result are bit-identical.
#include <stdlib.h>
#include <string.h>

#include <vector>

#include "gf2_internal.h"

#define ELIM_RREF 0
#define ELIM_NORMALIZE 1
#define ELIM_THREADS 1024
#define ELIM_WAVES (ELIM_THREADS / 64)
#define ELIM_MAX_LD 2048          // pivot row staged in LDS: 16 KiB

#define ELIM_STATUS_OK 0
#define ELIM_STATUS_DEPENDENT 1

template <int MODE>
__global__ __launch_bounds__(ELIM_THREADS) void eliminate_kernel(u64* __restrict__ base, int64_t m, int64_t n,
                                                                 int64_t ld, int64_t offset, int64_t* __restrict__ pivots_base,
                                                                 int64_t pivots_stride, int64_t* __restrict__ rank_base,
                                                                 int64_t* __restrict__ swaps, int64_t* __restrict__ nswaps,
                                                                 int* __restrict__ status) {
    __shared__ u64 pivot_row[ELIM_MAX_LD];
    __shared__ int found;          // first row (phase A) / first column (swap search), or INT_MAX
    u64* a = base + (int64_t)blockIdx.x * m * ld;
    int64_t* pivots = pivots_base ? pivots_base + (int64_t)blockIdx.x * pivots_stride : nullptr;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    int64_t lead = 0;              // RREF: next pivot row.  NORMALIZE: the diagonal index i.
    int64_t swap_count = 0;
    const int64_t steps = MODE == ELIM_RREF ? n : m;
    for (int64_t step = 0; step < steps; ++step) {
        const int64_t col = MODE == ELIM_RREF ? step : step + offset;
        if (MODE == ELIM_RREF && lead >= m) break;
        if (MODE == ELIM_NORMALIZE) lead = step;
        const int64_t cw = col >> 6;
        const int cb = (int)(col & 63);

        // ---- A: first row >= lead with an odd entry in this column ------------------------------------------
        if (tid == 0) found = 0x7fffffff;
        __syncthreads();
        int first = 0x7fffffff;
        for (int64_t r0 = lead; r0 < m; r0 += ELIM_THREADS) {
            const int64_t row = r0 + tid;
            const bool odd = row < m && ((a[row * ld + cw] >> cb) & 1ull);
            const u64 mask = __ballot(odd);
            if (mask && lane == 0) atomicMin(&found, (int)(r0 + wave * 64 + __ffsll((long long)mask) - 1));
            __syncthreads();
            first = found;
            __syncthreads();                               // everyone has read `found` before it changes again
            if (first != 0x7fffffff) break;
        }
        const int64_t donor = first == 0x7fffffff ? -1 : first;

        // ---- B: bring a pivot to row `lead` ---------------------------------------------------------------------
        if (MODE == ELIM_RREF) {
            if (donor < 0) continue;                         // no pivot in this column
            if (donor != lead)
                for (int64_t w = tid; w < ld; w += ELIM_THREADS) {
                    const u64 x = a[lead * ld + w], y = a[donor * ld + w];
                    a[lead * ld + w] = y;
                    a[donor * ld + w] = x;
                }
            if (tid == 0 && pivots) pivots[lead] = col;
        } else if (donor >= 0) {
            if (donor != lead)                               // diagonal entry is even: add the donor row
                for (int64_t w = tid; w < ld; w += ELIM_THREADS) a[lead * ld + w] ^= a[donor * ld + w];
        } else {
            // no odd row: swap in the first odd column of the diagonal row (its state right now)
            if (tid == 0) found = 0x7fffffff;
            __syncthreads();
            for (int64_t w = cw + tid; w < ld; w += ELIM_THREADS) {
                u64 v = a[lead * ld + w];
                if (w == cw) v &= ~0ull << cb;
                if (v) atomicMin(&found, (int)(w * 64 + __ffsll((long long)v) - 1));
            }
            __syncthreads();
            const int64_t other = found;
            __syncthreads();
            if (other == 0x7fffffff) {
                if (tid == 0) {
                    status[blockIdx.x] = ELIM_STATUS_DEPENDENT;
                    if (nswaps) nswaps[blockIdx.x] = swap_count;
                }
                return;
            }
            if (tid == 0 && swaps) {
                swaps[2 * swap_count] = col;
                swaps[2 * swap_count + 1] = other;
            }
            swap_count += 1;
            const int64_t ow = other >> 6;
            const int ob = (int)(other & 63);
            for (int64_t row = tid; row < m; row += ELIM_THREADS) {
                u64 x = a[row * ld + cw], y = a[row * ld + ow];
                const u64 diff = ((x >> cb) ^ (y >> ob)) & 1ull;
                if (cw == ow) {
                    x ^= (diff << cb) | (diff << ob);
                    a[row * ld + cw] = x;
                } else {
                    a[row * ld + cw] = x ^ (diff << cb);
                    a[row * ld + ow] = y ^ (diff << ob);
                }
            }
        }
        __syncthreads();

        // ---- C: clear the column in every other row ---------------------------------------------------------------
        for (int64_t w = tid; w < ld; w += ELIM_THREADS) pivot_row[w] = a[lead * ld + w];
        __syncthreads();
        for (int64_t r0 = (int64_t)wave * 64; r0 < m; r0 += ELIM_THREADS) {
            const int64_t row = r0 + lane;
            const bool odd = row < m && row != lead && ((a[row * ld + cw] >> cb) & 1ull);
            u64 mask = __ballot(odd);
            while (mask) {
                const int k = __ffsll((long long)mask) - 1;
                mask &= mask - 1;
                u64* target = a + (r0 + k) * ld;
                for (int64_t w = lane; w < ld; w += 64) target[w] ^= pivot_row[w];
            }
        }
        __syncthreads();
        if (MODE == ELIM_RREF) lead += 1;
    }
    if (tid == 0) {
        if (MODE == ELIM_RREF && rank_base) rank_base[blockIdx.x] = lead;
        if (MODE == ELIM_NORMALIZE && nswaps) nswaps[blockIdx.x] = swap_count;
    }
}

// ---- blocked RREF -----------------------------------------------------------------------------------------------------
//
// The reduced row echelon form is unique, so the elimination may pick any pivot row and move rows whenever it
// likes.  rref_blocked_kernel keeps every row in place and works in panels of 64 columns:
//
//   1. panel analysis (registers; one barrier per column).  A lane owns rows tid, tid+1024, .. and holds their
//      panel words.  For each column: every wavefront proposes an unused row with the bit set (__ballot), the first
//      proposal wins, and every lane clears the bit in its rows with the winner's panel word, recording the
//      operation in the row's 64-bit coefficient c_i.  With P_p the pivot row's value when chosen,
//      P = V . OLDPIV (V unit lower triangular, from the coefficients the pivot rows had when chosen) and
//      new_i = old_i ^ (c_i . V) . OLDPIV, where OLDPIV are the chosen rows as they stand at the start of the panel.
//   2. d_i = c_i . V through byte tables of V.
//   3. trailing update A[i] ^= d_i . OLDPIV, Method of Four Russians: per chunk of W words the 8 x 256 XOR
//      combinations of the pivot rows go to LDS and every row does 8 lookups.  Chunks left of the panel are skipped
//      while no pivot-free column has been seen there (they cannot change).
//   4. after the last panel the pivot rows are gathered into rows 0..rank-1 (gather_rows_kernel) and the rest zeroed.
//
// One workgroup per matrix; a batch of matrices is one workgroup each.
#define RB_THREADS 1024

template <int RPT, int W>
__global__ __launch_bounds__(RB_THREADS) void rref_blocked_kernel(u64* __restrict__ base, int64_t m, int64_t n, int64_t ld,
                                                                 int64_t* __restrict__ pivots_base, int64_t cap,
                                                                 int64_t* __restrict__ rank_base,
                                                                 int32_t* __restrict__ pivrow_base) {
    extern __shared__ __attribute__((aligned(16))) u64 smem[];
    const int64_t m_pad = (m + 1) & ~(int64_t)1;
    u64* T = smem;                                 // 2048 * W   (also VT: 8 x 256 words, before T is built)
    u64* dl = T + 2048 * W;                        // m_pad
    u64* snap = dl + m_pad;                        // 64 * W
    u64* V = snap + 64 * W;                        // 64
    u64* csel = V + 64;                            // 64
    u64* slot_w = csel + 64;                       // 2 x 16
    int* slot_row = reinterpret_cast<int*>(slot_w + 32);      // 2 x 16 ints
    int* prow_l = slot_row + 32;                   // 64 ints

    u64* a = base + (int64_t)blockIdx.x * m * ld;
    int64_t* pivots = pivots_base ? pivots_base + (int64_t)blockIdx.x * cap : nullptr;
    int32_t* pivrow = pivrow_base + (int64_t)blockIdx.x * cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

    unsigned int usedmask = 0;                     // bit k: owned row tid + 1024 k is a pivot row already
    int64_t rank = 0;
    int64_t first_free = n;                        // first column seen without a pivot

    for (int64_t pw = 0; pw < ld && rank < m && pw * 64 < n; ++pw) {
        // ---- 1. panel analysis ---------------------------------------------------------------------------------------
        u64 w[RPT], c[RPT];
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int64_t row = tid + (int64_t)RB_THREADS * k;
            w[k] = row < m ? a[row * ld + pw] : 0ull;
            c[k] = 0;
        }
        int t = 0;
        for (int b = 0; b < 64; ++b) {
            const int64_t col = pw * 64 + b;
            if (col >= n || rank + t >= m) break;
            const int par = b & 1;
            int kc = -1;
            u64 wc = 0;
#pragma unroll
            for (int k = RPT - 1; k >= 0; --k)
                if (!((usedmask >> k) & 1u) && ((w[k] >> b) & 1ull)) {
                    kc = k;
                    wc = w[k];
                }
            const u64 bal = __ballot(kc >= 0);
            if (bal) {
                if (lane == __ffsll((long long)bal) - 1) {
                    slot_row[par * 16 + wave] = tid + RB_THREADS * kc;
                    slot_w[par * 16 + wave] = wc;
                }
            } else if (lane == 0) {
                slot_row[par * 16 + wave] = -1;
            }
            __syncthreads();
            const u64 valid = __ballot(slot_row[par * 16 + (lane & 15)] >= 0) & 0xFFFFull;
            if (!valid) {
                if (col < first_free) first_free = col;
                continue;
            }
            const int wsel = __ffsll((long long)valid) - 1;
            const int prow = slot_row[par * 16 + wsel];
            const u64 pword = slot_w[par * 16 + wsel];
#pragma unroll
            for (int k = 0; k < RPT; ++k) {
                const int row = tid + RB_THREADS * k;
                if (row == prow) {
                    csel[t] = c[k];
                    usedmask |= 1u << k;
                } else if ((w[k] >> b) & 1ull) {
                    w[k] ^= pword;
                    c[k] |= 1ull << t;
                }
            }
            if (tid == 0) {
                prow_l[t] = prow;
                pivrow[rank + t] = prow;
                if (pivots) pivots[rank + t] = col;
            }
            t += 1;
        }
        __syncthreads();
        if (t == 0) continue;

        // ---- 2. V (forward substitution in one wavefront), its byte tables, d_i = c_i . V ---------------------------------
        if (wave == 0) {
            u64 v = lane < t ? 1ull << lane : 0ull;
            const u64 cs = lane < t ? csel[lane] : 0ull;
            for (int q = 0; q < t; ++q) {
                const u64 vq = ((u64)(unsigned int)__builtin_amdgcn_readlane((int)(v >> 32), q) << 32) |
                               (unsigned int)__builtin_amdgcn_readlane((int)(unsigned int)v, q);
                if ((cs >> q) & 1ull) v ^= vq;
            }
            V[lane] = v;
        }
        __syncthreads();
        for (int idx = tid; idx < 2048; idx += RB_THREADS) {
            const int g = idx >> 8, vv = idx & 255;
            u64 x = 0;
#pragma unroll
            for (int k = 0; k < 8; ++k)
                if ((vv >> k) & 1) x ^= V[8 * g + k];
            T[idx] = x;
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < RPT; ++k) {
            const int64_t row = tid + (int64_t)RB_THREADS * k;
            if (row < m) {
                u64 d = 0;
#pragma unroll
                for (int g = 0; g < 8; ++g) d ^= T[g * 256 + (int)((c[k] >> (8 * g)) & 255ull)];
                dl[row] = d;
            }
        }
        __syncthreads();

        // ---- 3. trailing update, chunk by chunk --------------------------------------------------------------------------
        const int groups = (t + 7) >> 3;
        const int64_t untouched = first_free < pw * 64 ? first_free : pw * 64;   // columns below this cannot change
        for (int64_t cw0 = 0; cw0 < ld; cw0 += W) {
            if ((cw0 + W) * 64 <= untouched) continue;
            const int wc_n = ld - cw0 < W ? (int)(ld - cw0) : W;
            for (int idx = tid; idx < t * W; idx += RB_THREADS) {
                const int q = idx / W, wd = idx % W;
                snap[idx] = wd < wc_n ? a[(int64_t)prow_l[q] * ld + cw0 + wd] : 0ull;
            }
            __syncthreads();
            for (int idx = tid; idx < groups * 256 * W; idx += RB_THREADS) {
                const int wd = idx % W, vv = (idx / W) & 255, g = idx / (256 * W);
                u64 x = 0;
#pragma unroll
                for (int k = 0; k < 8; ++k)
                    if (((vv >> k) & 1) && 8 * g + k < t) x ^= snap[(8 * g + k) * W + wd];
                T[idx] = x;
            }
            __syncthreads();
            {   // all loads first (independent, in flight together), then the lookups, then the stores
                constexpr int ITEMS = RPT * W;
                u64 x[ITEMS], dd[ITEMS];
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int64_t idx = tid + (int64_t)RB_THREADS * it;
                    const int64_t row = idx / W;
                    const int wd = (int)(idx % W);
                    const bool live = row < m && wd < wc_n;
                    dd[it] = live ? dl[row] : 0ull;
                    x[it] = dd[it] ? a[row * ld + cw0 + wd] : 0ull;
                }
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int wd = (int)((tid + (int64_t)RB_THREADS * it) % W);
                    for (int g = 0; g < groups; ++g)
                        x[it] ^= T[(g * 256 + (int)((dd[it] >> (8 * g)) & 255ull)) * W + wd];
                }
#pragma unroll
                for (int it = 0; it < ITEMS; ++it) {
                    const int64_t idx = tid + (int64_t)RB_THREADS * it;
                    if (dd[it]) a[(idx / W) * ld + cw0 + (idx % W)] = x[it];
                }
            }
            __syncthreads();
        }
        rank += t;
    }
    if (tid == 0) rank_base[blockIdx.x] = rank;
}

// out row k (k < rank) = in row pivrow[k]; rows >= rank are zero.  grid (m, batch), block 64.
__global__ void gather_rows_kernel(const u64* __restrict__ in, u64* __restrict__ out, const int32_t* __restrict__ pivrow,
                                   const int64_t* __restrict__ rank, int64_t m, int64_t ld, int64_t cap) {
    const int64_t k = blockIdx.x, mat = blockIdx.y;
    const u64* src = in + mat * m * ld;
    u64* dst = out + mat * m * ld + k * ld;
    if (k < rank[mat]) {
        const u64* row = src + (int64_t)pivrow[mat * cap + k] * ld;
        for (int64_t wd = threadIdx.x; wd < ld; wd += 64) dst[wd] = row[wd];
    } else {
        for (int64_t wd = threadIdx.x; wd < ld; wd += 64) dst[wd] = 0ull;
    }
}

__global__ void swap_columns_kernel(u64* __restrict__ a, int64_t m, int64_t ld, int64_t i, int64_t j) {
    const int64_t row = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (row >= m) return;
    const int64_t wi = i >> 6, wj = j >> 6;
    const int bi = (int)(i & 63), bj = (int)(j & 63);
    u64 x = a[row * ld + wi], y = a[row * ld + wj];
    const u64 diff = ((x >> bi) ^ (y >> bj)) & 1ull;
    if (wi == wj) {
        a[row * ld + wi] = x ^ ((diff << bi) | (diff << bj));
    } else {
        a[row * ld + wi] = x ^ (diff << bi);
        a[row * ld + wj] = y ^ (diff << bj);
    }
}

// one wave per row
__global__ __launch_bounds__(256) void row_weights_kernel(const u64* __restrict__ a, int64_t m, int64_t ld,
                                                          uint32_t* __restrict__ weights) {
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (row >= m) return;
    unsigned int acc = 0;
    for (int64_t w = lane; w < ld; w += 64) acc += __popcll(a[row * ld + w]);
    for (int off = 32; off > 0; off >>= 1) acc += __shfl_xor(acc, off);
    if (lane == 0) weights[row] = acc;
}

// Output word (t, w): column c is pivot column P[i] -> bit R[i, F[t]]; c == F[t] -> 1; other free columns -> 0.
__global__ void nullspace_kernel(const u64* __restrict__ red, int64_t n, int64_t ld, const int32_t* __restrict__ free_cols,
                                 int64_t nfree, const int32_t* __restrict__ col_pivot_row, u64* __restrict__ out,
                                 int64_t ldn) {
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t t = blockIdx.y;
    if (w >= ldn || t >= nfree) return;
    const int64_t fc = free_cols[t];
    const int64_t fw = fc >> 6;
    const int fb = (int)(fc & 63);
    u64 acc = 0;
    for (int b = 0; b < 64; ++b) {
        const int64_t c = w * 64 + b;
        if (c >= n) break;
        const int32_t pr = col_pivot_row[c];
        u64 bit;
        if (pr >= 0)
            bit = (red[(int64_t)pr * ld + fw] >> fb) & 1ull;
        else
            bit = c == fc ? 1ull : 0ull;
        acc |= bit << b;
    }
    out[t * ldn + w] = acc;
}

// ---- host side -----------------------------------------------------------------------------------------------------

static int launch_eliminate(gf2_ctx* ctx, int mode, u64* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld,
                            int64_t offset, int64_t* pivots_dev, int64_t pivots_stride, int64_t* rank_dev,
                            int64_t* swaps_dev, int64_t* nswaps_dev, int* status_dev) {
    if (ld > ELIM_MAX_LD) GF2_FAIL(GF2_E_ARG, "elimination supports at most %d columns (ld=%lld)", ELIM_MAX_LD * 64, (long long)ld);
    if (m >= 0x7fffffffLL || n >= 0x7fffffffLL) GF2_FAIL(GF2_E_ARG, "elimination: matrix too large");
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
    if (mode == ELIM_RREF)
        hipLaunchKernelGGL(eliminate_kernel<ELIM_RREF>, dim3((unsigned)batch), dim3(ELIM_THREADS), 0, ctx->stream, a_dev, m, n,
                           ld, offset, pivots_dev, pivots_stride, rank_dev, swaps_dev, nswaps_dev, status_dev);
    else
        hipLaunchKernelGGL(eliminate_kernel<ELIM_NORMALIZE>, dim3((unsigned)batch), dim3(ELIM_THREADS), 0, ctx->stream, a_dev,
                           m, n, ld, offset, pivots_dev, pivots_stride, rank_dev, swaps_dev, nswaps_dev, status_dev);
    GF2_TRY(gf2_prof_end(ctx));
    GF2_HIP(hipGetLastError());
    return GF2_OK;
}

extern "C" {

// Blocked path: m <= 8192.  Needs workspace for the row gather (a copy of the batch) and the pivot-row lists.
static int launch_rref_blocked(gf2_ctx* ctx, u64* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld,
                               int64_t* pivots_dev, int64_t cap, int64_t* rank_dev) {
    const int rpt = (int)gf2_cdiv(m, RB_THREADS);
    const int64_t m_pad = (m + 1) & ~(int64_t)1;
    const size_t abytes = (size_t)batch * m * ld * 8;
    const size_t pbytes = ((size_t)batch * cap * 4 + 255) & ~(size_t)255;
    GF2_TRY(gf2_ws_reserve(ctx, 1, pbytes + abytes));
    int32_t* pivrow = (int32_t*)ctx->ws[1];
    u64* tmp = (u64*)((char*)ctx->ws[1] + pbytes);
    const int w = m <= 2048 ? 8 : 4;
    const size_t shmem = ((size_t)2048 * w + m_pad + 64 * w + 64 + 64 + 32) * 8 + (32 + 64) * 4;
    GF2_TRY(gf2_prof_begin(ctx, GF2_K_ELIM));
#define GF2_RB_LAUNCH(RPT, WW)                                                                                           \
    do {                                                                                                                 \
        static bool attr_done = false;                                                                                   \
        if (!attr_done) {                                                                                                \
            GF2_HIP(hipFuncSetAttribute((const void*)rref_blocked_kernel<RPT, WW>,                                        \
                                        hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));                         \
            attr_done = true;                                                                                            \
        }                                                                                                                \
        hipLaunchKernelGGL((rref_blocked_kernel<RPT, WW>), dim3((unsigned)batch), dim3(RB_THREADS), shmem, ctx->stream,   \
                           a_dev, m, n, ld, pivots_dev, cap, rank_dev, pivrow);                                           \
    } while (0)
    if (rpt <= 1)
        GF2_RB_LAUNCH(1, 8);
    else if (rpt <= 2)
        GF2_RB_LAUNCH(2, 8);
    else if (rpt <= 4)
        GF2_RB_LAUNCH(4, 4);
    else
        GF2_RB_LAUNCH(8, 4);
#undef GF2_RB_LAUNCH
    GF2_HIP(hipGetLastError());
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)m, (unsigned)batch), dim3(64), 0, ctx->stream, (const u64*)a_dev,
                       tmp, pivrow, rank_dev, m, ld, cap);
    GF2_HIP(hipGetLastError());
    GF2_HIP(hipMemcpyAsync(a_dev, tmp, abytes, hipMemcpyDeviceToDevice, ctx->stream));
    GF2_TRY(gf2_prof_end(ctx));
    return GF2_OK;
}

int gf2_rref_batch_dev(gf2_ctx* ctx, uint64_t* a_dev, int64_t batch, int64_t m, int64_t n, int64_t ld,
                       int64_t* pivots_dev, int64_t* rank_dev) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: null context");
    if (batch < 0 || m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: bad shape");
    if (!rank_dev) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: null rank buffer");
    GF2_TRY(gf2_ctx_activate(ctx));
    if (batch == 0) return GF2_OK;
    if (m == 0 || n == 0) return gf2_dev_zero(ctx, rank_dev, (size_t)batch * 8);
    if (!a_dev) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch_dev: null matrix");
    const int64_t cap = m < n ? m : n;
    if (m <= 8 * RB_THREADS && batch <= 65535 && m <= 0x7fffffff && getenv("GF2_RREF_SEQUENTIAL") == nullptr)
        return launch_rref_blocked(ctx, (u64*)a_dev, batch, m, n, ld, pivots_dev, cap, rank_dev);
    return launch_eliminate(ctx, ELIM_RREF, (u64*)a_dev, batch, m, n, ld, 0, pivots_dev, cap, rank_dev, nullptr, nullptr,
                            nullptr);
}

int gf2_rref_batch(gf2_ctx* ctx, uint64_t* a, int64_t batch, int64_t m, int64_t n, int64_t ld, int64_t* pivots_out,
                   int64_t* rank_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: null context");
    if (batch < 0 || m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: bad shape");
    if (!rank_out) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: null rank output");
    if (batch == 0) return GF2_OK;
    if (m == 0 || n == 0) {
        for (int64_t b = 0; b < batch; ++b) rank_out[b] = 0;
        return GF2_OK;
    }
    if (!a) GF2_FAIL(GF2_E_ARG, "gf2_rref_batch: null matrix");
    const int64_t cap = m < n ? m : n;
    const size_t abytes = (size_t)batch * m * ld * 8;
    uint64_t* a_dev = nullptr;
    int64_t *piv_dev = nullptr, *rank_dev = nullptr;
    int rc = gf2_dev_alloc(ctx, abytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)batch * cap * 8, (void**)&piv_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)batch * 8, (void**)&rank_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, a_dev, a, abytes);
    if (rc == GF2_OK) rc = gf2_dev_zero(ctx, piv_dev, (size_t)batch * cap * 8);
    if (rc == GF2_OK) rc = gf2_rref_batch_dev(ctx, a_dev, batch, m, n, ld, piv_dev, rank_dev);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, a, a_dev, abytes);
    if (rc == GF2_OK && pivots_out) rc = gf2_d2h(ctx, pivots_out, piv_dev, (size_t)batch * cap * 8);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, rank_out, rank_dev, (size_t)batch * 8);
    gf2_dev_free(ctx, a_dev);
    gf2_dev_free(ctx, piv_dev);
    gf2_dev_free(ctx, rank_dev);
    return rc;
}

int gf2_rref(gf2_ctx* ctx, uint64_t* a, int64_t m, int64_t n, int64_t ld, int64_t* pivots_out, int64_t* rank_out) {
    return gf2_rref_batch(ctx, a, 1, m, n, ld, pivots_out, rank_out);
}

int gf2_normalize_dev(gf2_ctx* ctx, uint64_t* h_dev, int64_t r, int64_t n, int64_t ld, int64_t offset,
                      int64_t* swaps_dev, int64_t* nswaps_dev, int* status_dev) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: null context");
    if (r < 0 || n < 0 || offset < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: bad shape");
    if (n < offset + r) GF2_FAIL(GF2_E_COLUMNS, "not enough columns");
    if (!nswaps_dev || !status_dev) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: null output");
    GF2_TRY(gf2_ctx_activate(ctx));
    GF2_TRY(gf2_dev_zero(ctx, nswaps_dev, 8));
    GF2_TRY(gf2_dev_zero(ctx, status_dev, 4));
    if (r == 0) return GF2_OK;
    if (!h_dev) GF2_FAIL(GF2_E_ARG, "gf2_normalize_dev: null matrix");
    return launch_eliminate(ctx, ELIM_NORMALIZE, (u64*)h_dev, 1, r, n, ld, offset, nullptr, 0, nullptr, swaps_dev,
                            nswaps_dev, status_dev);
}

int gf2_normalize(gf2_ctx* ctx, uint64_t* h, int64_t r, int64_t n, int64_t ld, int64_t offset, int64_t* swaps_out,
                  int64_t* nswaps_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_normalize: null context");
    if (r < 0 || n < 0 || offset < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_normalize: bad shape");
    if (!nswaps_out) GF2_FAIL(GF2_E_ARG, "gf2_normalize: null output");
    *nswaps_out = 0;
    if (n < offset + r) GF2_FAIL(GF2_E_COLUMNS, "not enough columns");
    if (r == 0) return GF2_OK;
    if (!h || !swaps_out) GF2_FAIL(GF2_E_ARG, "gf2_normalize: null buffer");
    const size_t hbytes = (size_t)r * ld * 8;
    uint64_t* h_dev = nullptr;
    int64_t *swaps_dev = nullptr, *nswaps_dev = nullptr;
    int* status_dev = nullptr;
    int status = 0;
    int rc = gf2_dev_alloc(ctx, hbytes, (void**)&h_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)2 * r * 8, (void**)&swaps_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 8, (void**)&nswaps_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 4, (void**)&status_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, h_dev, h, hbytes);
    if (rc == GF2_OK) rc = gf2_normalize_dev(ctx, h_dev, r, n, ld, offset, swaps_dev, nswaps_dev, status_dev);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, &status, status_dev, 4);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, nswaps_out, nswaps_dev, 8);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, h, h_dev, hbytes);
    if (rc == GF2_OK && *nswaps_out > 0) rc = gf2_d2h(ctx, swaps_out, swaps_dev, (size_t)2 * (*nswaps_out) * 8);
    gf2_dev_free(ctx, h_dev);
    gf2_dev_free(ctx, swaps_dev);
    gf2_dev_free(ctx, nswaps_dev);
    gf2_dev_free(ctx, status_dev);
    if (rc == GF2_OK && status == ELIM_STATUS_DEPENDENT) GF2_FAIL(GF2_E_DEPENDENT, "rows are not independent");
    return rc;
}

int gf2_swap_columns(gf2_ctx* ctx, uint64_t* a, int64_t m, int64_t n, int64_t ld, int64_t i, int64_t j) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: null context");
    if (m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: bad shape");
    if (i < 0 || j < 0 || i >= n || j >= n) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: column index out of range");
    if (m == 0 || i == j) return GF2_OK;
    if (!a) GF2_FAIL(GF2_E_ARG, "gf2_swap_columns: null matrix");
    const size_t bytes = (size_t)m * ld * 8;
    uint64_t* a_dev = nullptr;
    int rc = gf2_dev_alloc(ctx, bytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, a_dev, a, bytes);
    if (rc == GF2_OK) {
        hipLaunchKernelGGL(swap_columns_kernel, dim3((unsigned)gf2_cdiv(m, 256)), dim3(256), 0, ctx->stream, (u64*)a_dev, m,
                           ld, i, j);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) {
            gf2_set_error("swap_columns_kernel failed: %s", hipGetErrorString(err));
            rc = GF2_E_HIP;
        }
    }
    if (rc == GF2_OK) rc = gf2_d2h(ctx, a, a_dev, bytes);
    gf2_dev_free(ctx, a_dev);
    return rc;
}

int gf2_row_weights(gf2_ctx* ctx, const uint64_t* a, int64_t m, int64_t n, int64_t ld, uint32_t* weights_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: null context");
    if (m < 0 || n < 0 || ld < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: bad shape");
    if (m == 0) return GF2_OK;
    if (!weights_out) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: null output");
    if (n == 0 || ld == 0) {
        for (int64_t i = 0; i < m; ++i) weights_out[i] = 0;
        return GF2_OK;
    }
    if (!a) GF2_FAIL(GF2_E_ARG, "gf2_row_weights: null matrix");
    const size_t bytes = (size_t)m * ld * 8;
    uint64_t* a_dev = nullptr;
    uint32_t* w_dev = nullptr;
    int rc = gf2_dev_alloc(ctx, bytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)m * 4, (void**)&w_dev);
    if (rc == GF2_OK) rc = gf2_h2d(ctx, a_dev, a, bytes);
    if (rc == GF2_OK) {
        hipLaunchKernelGGL(row_weights_kernel, dim3((unsigned)gf2_cdiv(m, 4)), dim3(256), 0, ctx->stream, (const u64*)a_dev,
                           m, ld, w_dev);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess) {
            gf2_set_error("row_weights_kernel failed: %s", hipGetErrorString(err));
            rc = GF2_E_HIP;
        }
    }
    if (rc == GF2_OK) rc = gf2_d2h(ctx, weights_out, w_dev, (size_t)m * 4);
    gf2_dev_free(ctx, a_dev);
    gf2_dev_free(ctx, w_dev);
    return rc;
}

int gf2_nullspace(gf2_ctx* ctx, const uint64_t* a, int64_t m, int64_t n, int64_t ld, uint64_t* n_out, int64_t ldn,
                  int64_t* rows_out) {
    if (!ctx) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: null context");
    if (m < 0 || n < 0 || ld < gf2_words(n) || ldn < gf2_words(n)) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: bad shape");
    if (!rows_out) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: null output");
    *rows_out = 0;
    if (n == 0) return GF2_OK;
    if (!n_out || (!a && m > 0)) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: null buffer");
    if (n >= 0x7fffffffLL) GF2_FAIL(GF2_E_ARG, "gf2_nullspace: too many columns");
    const int64_t cap = m < n ? m : n;
    const size_t abytes = (size_t)(m > 0 ? m : 1) * ld * 8;
    uint64_t *a_dev = nullptr, *out_dev = nullptr;
    int64_t *piv_dev = nullptr, *rank_dev = nullptr;
    int32_t *free_dev = nullptr, *map_dev = nullptr;
    int64_t rank = 0;
    std::vector<int64_t> pivots((size_t)(cap > 0 ? cap : 1));
    std::vector<int32_t> col_pivot_row((size_t)n, -1), free_cols;
    int rc = gf2_dev_alloc(ctx, abytes, (void**)&a_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)(cap > 0 ? cap : 1) * 8, (void**)&piv_dev);
    if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, 8, (void**)&rank_dev);
    if (rc == GF2_OK && m > 0) rc = gf2_h2d(ctx, a_dev, a, (size_t)m * ld * 8);
    if (rc == GF2_OK) rc = gf2_rref_batch_dev(ctx, a_dev, 1, m, n, ld, piv_dev, rank_dev);
    if (rc == GF2_OK) rc = gf2_d2h(ctx, &rank, rank_dev, 8);
    if (rc == GF2_OK && rank > 0) rc = gf2_d2h(ctx, pivots.data(), piv_dev, (size_t)rank * 8);
    if (rc == GF2_OK) {
        for (int64_t i = 0; i < rank; ++i) col_pivot_row[(size_t)pivots[(size_t)i]] = (int32_t)i;
        for (int64_t c = 0; c < n; ++c)
            if (col_pivot_row[(size_t)c] < 0) free_cols.push_back((int32_t)c);
    }
    const int64_t nfree = (int64_t)free_cols.size();
    if (rc == GF2_OK && nfree > 0) {
        rc = gf2_dev_alloc(ctx, (size_t)nfree * ldn * 8, (void**)&out_dev);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)nfree * 4, (void**)&free_dev);
        if (rc == GF2_OK) rc = gf2_dev_alloc(ctx, (size_t)n * 4, (void**)&map_dev);
        if (rc == GF2_OK) rc = gf2_h2d(ctx, free_dev, free_cols.data(), (size_t)nfree * 4);
        if (rc == GF2_OK) rc = gf2_h2d(ctx, map_dev, col_pivot_row.data(), (size_t)n * 4);
        if (rc == GF2_OK) {
            dim3 grid((unsigned)gf2_cdiv(ldn, 64), (unsigned)nfree);
            hipLaunchKernelGGL(nullspace_kernel, grid, dim3(64), 0, ctx->stream, (const u64*)a_dev, n, ld, free_dev, nfree,
                               map_dev, (u64*)out_dev, ldn);
            hipError_t err = hipGetLastError();
            if (err != hipSuccess) {
                gf2_set_error("nullspace_kernel failed: %s", hipGetErrorString(err));
                rc = GF2_E_HIP;
            }
        }
        if (rc == GF2_OK) rc = gf2_d2h(ctx, n_out, out_dev, (size_t)nfree * ldn * 8);
    }
    if (rc == GF2_OK) *rows_out = nfree;
    gf2_dev_free(ctx, a_dev);
    gf2_dev_free(ctx, piv_dev);
    gf2_dev_free(ctx, rank_dev);
    gf2_dev_free(ctx, out_dev);
    gf2_dev_free(ctx, free_dev);
    gf2_dev_free(ctx, map_dev);
    return rc;
}

}  // extern "C"
